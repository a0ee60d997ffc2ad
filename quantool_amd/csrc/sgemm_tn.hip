// See sgemm_tn.h.  256 threads = 4 waves (2x2); each wave owns a (BM/2)x(BN/2) block of
// 32x32 MFMA accumulators.  Operand panels [BK][BM] / [BK][BN] are register-staged into a
// double-buffered LDS image (rows contiguous, so fragment reads are conflict-free b32 reads of
// 32 consecutive floats per lane half) and zero-filled at every edge, so any M, N, k works.
#include <stdlib.h>

#include <type_traits>

#include "sgemm_tn.h"

namespace {

constexpr int BK = 16;

#ifndef QT_SGEMM_PD          // k-steps whose global loads are in flight in the 64x64 tile's loop (lab builds: -DQT_SGEMM_PD=n)
#define QT_SGEMM_PD 2
#endif

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), in order
template <int N, class F>
__device__ __forceinline__ void sg_static_for(F&& f) {
    if constexpr (N > 0) {
        sg_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// amdgpu_waves_per_eu(2): the MODE_SUB variants hold a C tile next to the accumulators; capping the
// register budget at two waves per SIMD makes hipcc park the excess in AGPRs instead of taking all
// 512 registers (one wave per SIMD leaves the C read / write phases of a workgroup uncovered).
// NWAVE = 8 (MODE_SUB, 128x128): the same tile on 8 waves (4 x 2, 32x64 outputs each), so a wave holds
// 32 + 32 accumulator / C registers instead of 64 + 64 and FOUR waves fit a SIMD (two workgroups per CU):
// the f32 MFMA pipe measured 61 % busy at two waves per SIMD on the sweep's k = 512 updates (PMC,
// profiles/r02_gemm_pmc.txt).  Per output element the k order is unchanged (bit-identical results).
template <int BM, int BN, int MODE, bool CHAIN = false, int NWAVE = 4>
__global__ __launch_bounds__(64 * NWAVE) __attribute__((amdgpu_waves_per_eu(NWAVE == 8 ? 4 : 2))) void sgemm_tn_kernel(const SgemmArgs p) {
    constexpr int NT = 64 * NWAVE;
    constexpr int WGM = NWAVE == 8 ? 4 : 2, WGN = 2;          // wave grid
    constexpr int WM = BM / (32 * WGM), WN = BN / (32 * WGN);  // 32x32 sub-tiles per wave in m / n
    constexpr int A4 = BK * BM / 4 / NT;       // float4 loads per thread for A
    constexpr int B4 = BK * BN / 4 / NT;
    static_assert(A4 >= 1 && B4 >= 1, "tile too small for the workgroup");
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave >> 1, wave_n = wave & 1;   // WGN == 2
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (p.upper_only && m0 >= n0 + BN) return;   // every row of this tile lies below every column: not wanted
    // The argument block stays untouched (const): a kernel that modifies its by-value struct AND indexes an array in it at
    // run time gets the whole struct copied to scratch -- every operand pointer and pitch inside the k-loop was a scratch load in
    // the first round-4 builds (264 B of private segment per lane; the chains' 64x64 products ran 10-25 % longer).
    const float* Ap = p.A;
    const float* Bp = p.B;
    const float* Cinp = p.Cin;
    float* Coutp = p.Cout;
    int zsplit = blockIdx.z;
    if (p.batch > 1) {        // problem b of a batch of identical shapes: every operand at its own stride
        const int b = zsplit / p.n_splits;
        zsplit -= b * p.n_splits;
        Ap += (size_t)b * p.bsA;
        Bp += (size_t)b * p.bsB;
        if (Cinp) Cinp += (size_t)b * p.bsCin;
        Coutp += (size_t)b * p.bsCout;
    }
    if (p.n_groups > 0) {     // stacked rows: the B operand of this tile's row group (boundaries are multiples of BM)
        int g = 0;
        while (g + 1 < p.n_groups && m0 >= p.group_m_end[g]) ++g;
        Bp += (size_t)g * p.group_bsB;
    }
    const bool tile_inside = p.fast_interior && m0 + BM <= p.M && n0 + BN <= p.N;   // wave-uniform

    int k_begin = 0;
    if (p.k_mode == SG_K_FROM_N0) k_begin = n0 / BK * BK;
    int k_end = p.kdim;
    if (p.k_chunk > 0) {  // split-K: this z-slice owns [z*k_chunk, (z+1)*k_chunk)
        const int lo = zsplit * p.k_chunk, hi = lo + p.k_chunk;
        k_begin = k_begin > lo ? k_begin : lo;
        k_end = k_end < hi ? k_end : hi;
        Coutp += (size_t)zsplit * (size_t)p.M * (size_t)p.N;
    }

    const bool a_vec = (p.lda % 4 == 0) && (((uintptr_t)Ap & 15) == 0);
    const bool b_vec = (p.ldb % 4 == 0) && (((uintptr_t)Bp & 15) == 0);

    // Global loads in flight: PD k-steps for the 64x64 tile (one float4 per operand and thread per step, so a deeper ring
    // is cheap there; a step is only 8 MFMAs per wave).  Measured once the argument struct was out of scratch (above):
    // PD = 2 / 4 / 8 give the same chain times (K = 4096: 3.09-3.11 / 3.13 / 3.17 ms, K = 14336: 21.1 / 21.3 / 21.3) -- the
    // k-loop's loads are not what a 64x64 product waits for: a wave's 32x32 output is ONE dependent chain of k / 2
    // MFMAs of 64 cycles each (k = 384: 12.3 k cycles = 5.6 us of a 9-12 us launch), the rest is the C read, the first
    // panel and the store.  PD = 2 with the branch-free fast loop is kept; the larger tiles (32+ MFMAs per step) keep
    // PD = 1.  Same k order: bit-identical.
    constexpr int PD = (BM == 64 && BN == 64 && !CHAIN) ? QT_SGEMM_PD : 1;
    static_assert(PD == 1 || PD % 2 == 0, "the LDS buffer of a step is s & 1: the unrolled ring must be even");
    f32x4 ra[PD][A4], rb[PD][B4];

    // whole tile inside the matrices and 16-byte loads legal: the k-steps that are also inside the
    // k range take straight vector loads (no per-element edge tests in the steady state); the C
    // prefetch and the epilogue of such a tile skip their bounds tests too
    const bool interior = p.fast_interior && a_vec && b_vec && m0 + BM <= p.M && n0 + BN <= p.N;
    auto gload_fast = [&](int k0, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int r = 0; r < A4; ++r) {
            const int idx = tid + NT * r;
            ra[slot][r] = *(const f32x4*)(Ap + (size_t)(k0 + idx / (BM / 4)) * p.lda + m0 + (idx % (BM / 4)) * 4);
        }
#pragma unroll
        for (int r = 0; r < B4; ++r) {
            const int idx = tid + NT * r;
            rb[slot][r] = *(const f32x4*)(Bp + (size_t)(k0 + idx / (BN / 4)) * p.ldb + n0 + (idx % (BN / 4)) * 4);
        }
    };
    auto gload = [&](int k0, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
        if (interior && k0 + BK <= k_end) {
            gload_fast(k0, slot_c);
            return;
        }
#pragma unroll
        for (int r = 0; r < A4; ++r) {
            const int idx = tid + NT * r;
            const int kr = idx / (BM / 4), c = (idx % (BM / 4)) * 4;
            const int k = k0 + kr, m = m0 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < k_end) {
                const float* src = Ap + (size_t)k * p.lda + m;
                if (a_vec && m + 3 < p.M) {
                    v = *(const f32x4*)src;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (m + e < p.M) v[e] = src[e];
                }
            }
            ra[slot][r] = v;
        }
#pragma unroll
        for (int r = 0; r < B4; ++r) {
            const int idx = tid + NT * r;
            const int kr = idx / (BN / 4), c = (idx % (BN / 4)) * 4;
            const int k = k0 + kr, n = n0 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < k_end) {
                const float* src = Bp + (size_t)k * p.ldb + n;
                if (b_vec && n + 3 < p.N) {
                    v = *(const f32x4*)src;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < p.N) v[e] = src[e];
                }
            }
            rb[slot][r] = v;
        }
    };
    auto lstore = [&](int buf, auto slot_c) {
        constexpr int slot = decltype(slot_c)::value;
#pragma unroll
        for (int r = 0; r < A4; ++r) {
            const int idx = tid + NT * r;
            *(f32x4*)&As[buf][idx / (BM / 4)][(idx % (BM / 4)) * 4] = ra[slot][r];
        }
#pragma unroll
        for (int r = 0; r < B4; ++r) {
            const int idx = tid + NT * r;
            *(f32x4*)&Bs[buf][idx / (BN / 4)][(idx % (BN / 4)) * 4] = rb[slot][r];
        }
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int h = lane >> 5, l31 = lane & 31;
    // MODE_SUB reads C: issue those loads before the k-loop so their latency hides behind the
    // MFMAs instead of heading the epilogue (short-k launches -- the sweep's trailing update has
    // only 8 k-steps -- are otherwise epilogue-bound)
    f32x16 cpre[WM][WN];
    if (MODE == SG_MODE_SUB) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wave_m * (BM / WGM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int col = n0 + wave_n * (BN / WGN) + j * 32 + l31;
                    cpre[i][j][r] = (tile_inside || (row < p.M && col < p.N)) ? Cinp[(size_t)row * p.ldcin + col] : 0.0f;
                }
    }
    const int nsteps = (k_end - k_begin + BK - 1) / BK;
    // one k-step: refill the register slot this step's panel came from with the panel PD steps ahead, run the step's
    // MFMAs from LDS, move the next step's panel (requested PD - 1 steps ago) into the other LDS buffer
    // FAST (whole tile and whole k-steps inside, nsteps % PD == 0): no branch in the step -- the refill is clamped to the
    // last panel and the last step's LDS store goes to the buffer nobody reads again -- so the compiler's s_waitcnt
    // counts stay counts (a conditional load or the edge path in the loop makes every wait a vmcnt(0))
    auto kstep = [&](int s, auto u_c, auto fast_c) {
        constexpr int u = decltype(u_c)::value;                // s % PD, static: registers cannot be indexed at run time
        constexpr bool FAST = decltype(fast_c)::value;
        const int buf = s & 1;
        if constexpr (FAST) {
            const int sn = s + PD < nsteps ? s + PD : nsteps - 1;
            gload_fast(k_begin + sn * BK, std::integral_constant<int, u>{});
        } else {
            if (s + PD < nsteps) gload(k_begin + (s + PD) * BK, std::integral_constant<int, u>{});
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
                float a[WM], b[WN];
#pragma unroll
                for (int i = 0; i < WM; ++i) a[i] = As[buf][2 * kk + h][wave_m * (BM / WGM) + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < WN; ++j) b[j] = Bs[buf][2 * kk + h][wave_n * (BN / WGN) + j * 32 + l31];
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        if (FAST || s + 1 < nsteps) lstore(buf ^ 1, std::integral_constant<int, (u + 1) % PD>{});
        if (CHAIN && s + 1 < nsteps && ((s + 1) * BK) % p.chain_len == 0) {
            // end of a chain: fold it into the C registers and start the next one from zero
            // (one 32x32 accumulator at a time, so only 16 extra registers are live)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        cpre[i][j][r] = cpre[i][j][r] - acc[i][j][r];
                        acc[i][j][r] = 0.0f;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        __syncthreads();
    };
    using I0 = std::integral_constant<int, 0>;
    if (PD > 1 && interior && nsteps > 0 && nsteps % PD == 0 && (k_end - k_begin) % BK == 0) {
        sg_static_for<PD>([&](auto d) { gload_fast(k_begin + decltype(d)::value * BK, d); });
        lstore(0, I0{});
        __syncthreads();
        for (int s0 = 0; s0 < nsteps; s0 += PD)           // buf = s & 1 stays right because PD is even
            sg_static_for<PD>([&](auto u) { kstep(s0 + decltype(u)::value, u, std::true_type{}); });
    } else if (nsteps > 0) {
        sg_static_for<PD>([&](auto d) {
            if (decltype(d)::value < nsteps) gload(k_begin + decltype(d)::value * BK, d);
        });
        lstore(0, I0{});
        __syncthreads();
        for (int s0 = 0; s0 < nsteps; s0 += PD)
            sg_static_for<PD>([&](auto u) {
                if (s0 + decltype(u)::value < nsteps) kstep(s0 + decltype(u)::value, u, std::false_type{});
            });
    }

#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wave_m * (BM / WGM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int col = n0 + wave_n * (BN / WGN) + j * 32 + l31;
                if (tile_inside || (row < p.M && col < p.N)) {
                    float v = acc[i][j][r];
                    if (MODE == SG_MODE_SUB) v = cpre[i][j][r] - v;
                    if (MODE == SG_MODE_NEG) v = -v;
                    Coutp[(size_t)row * p.ldcout + col] = v;
                }
            }
}

// ---------------------------------------------------------------------------------------------------
// sgemm_ring_kernel: MODE_SUB on an LDS-DMA ring, persistent over the launch's tiles.
//
// The register-staged kernel above runs the sweep's k = 512 far update at 66-73 % MFMA-busy (profiles/
// r02_gemm_pmc.txt) and even an 8192-deep product at only ~80 %: every k-step ends in vmcnt(0) + ds_write +
// __syncthreads (the next panel was requested one k-step earlier, so its latency is exposed whenever it
// exceeds one k-step), fragment reads are waited for right in front of their MFMAs, and every workgroup
// pays its C read, first-panel latency and store tail while its co-resident partner is in the same phase.
// Here:
//   * panels go global -> LDS by LDS-DMA (16 B per lane, one instruction per thread and panel) into a ring
//     of RS = 4 stages of BK = 16 k-rows; stages are consumed in pairs: the next pair is requested while the
//     current one is multiplied (32 MFMAs per wave of latency cover), with ONE raw barrier per pair;
//   * a workgroup walks its tiles (blockIdx.x, + gridDim.x, ...) as ONE stream of stages: the ring never
//     drains between tiles; a tile's C is loaded during its own second k-step (into the registers the
//     previous tile's stores have just released) and is not needed before its first chain ends.
// Wave -> output mapping, MFMA shape and k order are those of sgemm_tn_kernel<128, 128, MODE_SUB, *, 8>:
// per output element the same ascending-k fmaf chain(s) from 0 and the same subtraction(s), so the results
// are bit-identical (tests/test_gpu_sgemm.py compares the two kernels bit for bit).
// Requirements (the caller falls back to the kernel above otherwise): MODE_SUB, k_mode FULL, M % 128 == 0 and
// N % 128 == 0 (whole tiles only: no edge predicates anywhere), kdim % 32 == 0, kdim >= 64, chain_len % 32 == 0,
// lda / ldb % 4 == 0, A / B 16-byte aligned, 16 * ld * 4 < 2^32, 32 * ldc * 4 < 2^31.
constexpr int RBK = 16;                       // k rows per stage
constexpr int RBM = 128, RBN = 128;
constexpr int RS = 4;                         // ring stages
constexpr int RSTAGE_BYTES = RBK * (RBM + RBN) * 4;   // 16 KiB: A panel then B panel

struct RingArgs {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    const float* Cin; int64_t ldcin;
    float* Cout; int64_t ldcout;
    int M, N, kdim, chain_len;     // chain_len > 0
    int tiles_n, n_tiles;          // n_tiles = tiles_m * tiles_n (upper_only: listed tiles only), per problem
    int upper_only;
    // several problems in one launch (SgemmArgs): `batch` problems of this shape at operand strides, walked as
    // batch * n_tiles tiles; or row groups whose B operand sits at B + g * group_bsB
    int batch;
    int64_t bsA, bsB, bsCin, bsCout;
    int n_groups;
    int64_t group_bsB;
    int group_m_end[SG_MAX_GROUPS];
};

__device__ __forceinline__ void glds16_one(unsigned voff, const void* sbase, unsigned lds_dst) {
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

template <int NW>
__device__ __forceinline__ void ring_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW) : "memory");
}

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void sgemm_ring_kernel(RingArgs p) {
    __shared__ __attribute__((aligned(16))) char ring[RS * RSTAGE_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 1, wave_n = wave & 1;
    const int h = lane >> 5, l31 = lane & 31;
    const int steps = p.kdim / RBK;                 // per tile
    const int fold_every = p.chain_len / RBK;
    const int all_tiles = p.n_tiles * p.batch;
    const int my_tiles = (all_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;
    const int total = my_tiles * steps;             // stages this workgroup streams

    // tile it (0 .. my_tiles) of this workgroup -> (m0, n0) and the element offsets of its problem's / row group's operands
    auto tile_origin = [&](int it, int& m0, int& n0, size_t& offA, size_t& offB, size_t& offCin, size_t& offCout) {
        int t = (int)blockIdx.x + it * (int)gridDim.x;
        offA = offB = offCin = offCout = 0;
        if (p.batch > 1) {
            const int b = t / p.n_tiles;
            t -= b * p.n_tiles;
            offA = (size_t)b * p.bsA;
            offB = (size_t)b * p.bsB;
            offCin = (size_t)b * p.bsCin;
            offCout = (size_t)b * p.bsCout;
        }
        int tm, tn;
        if (p.upper_only) {
            // tiles with tn >= tm, listed row by row: row tm holds tiles_n - tm of them (tiles_m == tiles_n)
            int rem = t;
            tm = 0;
            while (rem >= p.tiles_n - tm) { rem -= p.tiles_n - tm; ++tm; }
            tn = tm + rem;
        } else {
            tm = t / p.tiles_n;
            tn = t - tm * p.tiles_n;
        }
        m0 = tm * RBM;
        n0 = tn * RBN;
        if (p.n_groups > 0) {
            int g = 0;
            while (g + 1 < p.n_groups && m0 >= p.group_m_end[g]) ++g;
            offB = (size_t)g * p.group_bsB;
        }
    };

    // DMA geometry: thread t moves 16 bytes of k row (t >> 5), columns 4 * (t & 31) .. + 3 of each panel; a wave's
    // instruction writes 1 KiB = two whole 512-byte panel rows, so the LDS image is plain [k][128] per panel
    const int drow = tid >> 5, dcol = (tid & 31) * 4;
    const unsigned voffA = (unsigned)(((size_t)drow * (size_t)p.lda + dcol) * 4), voffB = (unsigned)(((size_t)drow * (size_t)p.ldb + dcol) * 4);
    const unsigned ring_lds = (unsigned)(size_t)(QT_LDS char*)ring;
    const unsigned dst_wave = __builtin_amdgcn_readfirstlane(ring_lds + (unsigned)wave * 1024u);
    // The stream of stages is issued in order, so its position is kept incrementally (no division per stage: SALU
    // work right behind a barrier is executed by every wave at once, with the MFMA pipe idle)
    int is_it = 0, is_s = 0, is_g = 0;          // next stage to request: tile is_it, k-step is_s, stream index is_g
    const float *is_a, *is_b;
    {
        int m0i, n0i;
        size_t oa, ob, oci, oco;
        tile_origin(0, m0i, n0i, oa, ob, oci, oco);
        is_a = p.A + oa + m0i;
        is_b = p.B + ob + n0i;
    }
    const size_t a_step = (size_t)RBK * p.lda, b_step = (size_t)RBK * p.ldb;
    auto issue_next = [&]() {
        const unsigned d = dst_wave + (unsigned)(is_g & (RS - 1)) * RSTAGE_BYTES;
        glds16_one(voffA, is_a, d);
        glds16_one(voffB, is_b, d + RBK * RBM * 4);
        ++is_g;
        if (++is_s == steps) {
            is_s = 0;
            ++is_it;
            if (is_it < my_tiles) {
                int m0i, n0i;
                size_t oa, ob, oci, oco;
                tile_origin(is_it, m0i, n0i, oa, ob, oci, oco);
                is_a = p.A + oa + m0i;
                is_b = p.B + ob + n0i;
            }
        } else {
            is_a += a_step;
            is_b += b_step;
        }
    };
    static_assert((RS & (RS - 1)) == 0, "slot = stream index & (RS - 1)");

    f32x16 acc[2], cpre[2];
    // C addressing through buffer instructions: descriptor base = the wave's corner of the tile (scalar), soffset =
    // row / sub-tile offset (scalar), voffset = ONE per-lane byte offset that never changes -- no 64-bit per-lane
    // addresses (32 of them cost 64 registers and spill the C tile)
    const int cvoff_in = (int)(((size_t)(4 * h) * (size_t)p.ldcin + (size_t)l31) * 4);
    const int cvoff_out = (int)(((size_t)(4 * h) * (size_t)p.ldcout + (size_t)l31) * 4);
    const int cstep_in = (int)((size_t)p.ldcin * 4), cstep_out = (int)((size_t)p.ldcout * 4);
    // register r of a 32x32 accumulator is row (r & 3) + 8 (r >> 2) (+ 4 h): the per-lane offset walks the rows
    // (+1 row, and +5 rows after every fourth), the sub-tile j is an immediate
    size_t c_off_in = 0, c_off_out = 0;      // the current tile's problem (batch): element offsets of its C
    auto load_c = [&](int m0, int n0) {
        const int rw = m0 + wave_m * 32, cw = n0 + wave_n * 64;      // wave-uniform
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Cin + c_off_in + (size_t)rw * p.ldcin + cw), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int soff = ((r & 3) + 8 * (r >> 2)) * cstep_in;      // scalar; the sub-tile j is an immediate
#pragma unroll
            for (int j = 0; j < 2; ++j)
                cpre[j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, cvoff_in + j * 128, soff, 0));
        }
    };
    auto store_c = [&](int m0, int n0) {      // Cout = cpre - acc
        const int rw = m0 + wave_m * 32, cw = n0 + wave_n * 64;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Cout + c_off_out + (size_t)rw * p.ldcout + cw), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int soff = ((r & 3) + 8 * (r >> 2)) * cstep_out;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cpre[j][r] - acc[j][r]), rs, cvoff_out + j * 128, soff, 0);
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    };

    // prologue: first tile's C (older than every DMA), stages 0 and 1
    int m0, n0;
    {
        size_t oa, ob;
        tile_origin(0, m0, n0, oa, ob, c_off_in, c_off_out);
    }
    load_c(m0, n0);
    issue_next();
    issue_next();

    // ONE barrier per PAIR of stages (32 MFMAs per wave between barriers: a barrier is a moment at which every wave
    // of the workgroup has stopped feeding the MFMA pipe).  At the top of pair-step P the stages 2P and 2P + 1 were
    // requested one pair-step ago; stages 2P + 2, 2P + 3 are requested right behind the barrier into the slots of
    // pair P - 1, whose reads every wave finished before arriving here.
    // vmcnt bookkeeping: vmcnt counts EVERY vector-memory operation of the wave in issue order (DMA, C loads, C
    // stores), so "pair P has landed" = "everything is done except what was issued after pair P's request", and the
    // only such operations are the C batch of the END of pair-step P - 1: a tile's 32 stores at the end of its last
    // pair-step, or the 32 loads of a tile (not the first: prologue) at the end of its FIRST pair-step, into the
    // registers the previous tile's stores released one pair-step earlier.
    const int nchains = steps / fold_every;
    const int pairs_per_chain = fold_every / 2;
    int batch_prev = 0;
    int g = 0;
    for (int it = 0; it < my_tiles; ++it) {
        for (int ch = 0; ch < nchains; ++ch) {
            zero_acc();      // every chain starts from zero (its own accumulator live range: no copies at the loop edges)
            for (int ps = 0; ps < pairs_per_chain; ++ps, g += 2) {
                if (batch_prev) ring_wait_vmcnt<32>();
                else ring_wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                if (g + 2 < total) {
                    issue_next();
                    issue_next();
                }
                batch_prev = 0;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const char* stage = ring + ((g + half) & (RS - 1)) * RSTAGE_BYTES;
                    const float* As = (const float*)stage + wave_m * 32 + l31;
                    const float* Bs = (const float*)(stage + RBK * RBM * 4) + wave_n * 64 + l31;
#pragma unroll
                    for (int kk = 0; kk < RBK / 2; ++kk) {
                        const float a = As[(2 * kk + h) * RBM];
                        const float b0 = Bs[(2 * kk + h) * RBN], b1 = Bs[(2 * kk + h) * RBN + 32];
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
                    }
                }
                if (ps == 0 && ch == 0 && it > 0) {
                    // this tile's C into the C registers; first needed when this tile's first chain ends
                    load_c(m0, n0);
                    __builtin_amdgcn_sched_barrier(0);
                    batch_prev = 32;
                }
            }
            if (ch + 1 < nchains) {
                // end of a chain inside the tile: fold it into the C registers
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) cpre[j][r] = cpre[j][r] - acc[j][r];
            }
        }
        // last chain of the tile: C - chain, stored (the batch of the end of the tile's last pair-step)
        store_c(m0, n0);
        __builtin_amdgcn_sched_barrier(0);
        batch_prev = 32;
        if (it + 1 < my_tiles) {
            size_t oa, ob;
            tile_origin(it + 1, m0, n0, oa, ob, c_off_in, c_off_out);     // after this tile's stores were issued
        }
    }
}

// Cout = mode(Cin, sum_z slab[z]) in ascending z (deterministic)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int splits, int M,
                                                            int N, const float* __restrict__ Cin, int64_t ldcin,
                                                            float* __restrict__ Cout, int64_t ldcout, int mode,
                                                            int64_t bs_slabs, int64_t bs_cin, int64_t bs_cout) {
    const int col = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int row = blockIdx.y;
    if (col >= N) return;
    if (blockIdx.z) {       // problem b of a batch
        slabs += (size_t)blockIdx.z * bs_slabs;
        if (Cin) Cin += (size_t)blockIdx.z * bs_cin;
        Cout += (size_t)blockIdx.z * bs_cout;
    }
    const size_t mn = (size_t)M * N;
    if (col + 3 < N && (N & 3) == 0) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        // loads in batches of 8 (independent, all in flight together), adds in ascending z: the sum
        // order -- hence the result -- is that of the plain loop, without one memory round trip per slab
        const float* src = slabs + (size_t)row * N + col;
        int z = 0;
        for (; z + 8 <= splits; z += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(src + (size_t)(z + u) * mn);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < splits; ++z) s += *(const f32x4*)(src + (size_t)z * mn);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = s[e];
            if (mode == SG_MODE_SUB) v = Cin[(size_t)row * ldcin + col + e] - v;
            if (mode == SG_MODE_NEG) v = -v;
            Cout[(size_t)row * ldcout + col + e] = v;
        }
    } else {
        for (int e = 0; e < 4 && col + e < N; ++e) {
            float v = 0.f;
            for (int z = 0; z < splits; ++z) v += slabs[z * mn + (size_t)row * N + col + e];
            if (mode == SG_MODE_SUB) v = Cin[(size_t)row * ldcin + col + e] - v;
            if (mode == SG_MODE_NEG) v = -v;
            Cout[(size_t)row * ldcout + col + e] = v;
        }
    }
}

template <int BM, int BN>
int launch(const SgemmArgs& a, hipStream_t stream, int splits = 1) {
    dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, splits * (a.batch > 1 ? a.batch : 1));
    // 128x128 MODE_SUB tiles on 8 waves (four waves per SIMD) unless QT_SGEMM_SUB_WAVES=4
    static const bool sub8 = [] {
        const char* e = getenv("QT_SGEMM_SUB_WAVES");
        return !(e && atoi(e) == 4);
    }();
    switch (a.mode) {
        case SG_MODE_SUB:
            if constexpr (BM == 128 && BN == 128) {
                if (sub8) {
                    if (a.chain_len > 0)
                        hipLaunchKernelGGL((sgemm_tn_kernel<BM, BN, SG_MODE_SUB, true, 8>), grid, dim3(512), 0, stream, a);
                    else
                        hipLaunchKernelGGL((sgemm_tn_kernel<BM, BN, SG_MODE_SUB, false, 8>), grid, dim3(512), 0, stream, a);
                    break;
                }
            }
            if (a.chain_len > 0)
                hipLaunchKernelGGL((sgemm_tn_kernel<BM, BN, SG_MODE_SUB, true>), grid, dim3(256), 0, stream, a);
            else
                hipLaunchKernelGGL((sgemm_tn_kernel<BM, BN, SG_MODE_SUB>), grid, dim3(256), 0, stream, a);
            break;
        case SG_MODE_SET:
            hipLaunchKernelGGL((sgemm_tn_kernel<BM, BN, SG_MODE_SET>), grid, dim3(256), 0, stream, a);
            break;
        default:
            hipLaunchKernelGGL((sgemm_tn_kernel<BM, BN, SG_MODE_NEG>), grid, dim3(256), 0, stream, a);
            break;
    }
    QT_LAUNCH_CHECK();
    return QT_OK;
}

}  // namespace

int qt_splitk_reduce(const float* slabs, int splits, int M, int N, const float* Cin, int64_t ldcin, float* Cout,
                     int64_t ldcout, int mode, hipStream_t stream, int batch, int64_t bs_slabs, int64_t bs_cin,
                     int64_t bs_cout) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((N / 4 + 255) / 256 + 1, M, batch > 1 ? batch : 1), dim3(256), 0, stream,
                       slabs, splits, M, N, Cin, ldcin, Cout, ldcout, mode, bs_slabs, bs_cin, bs_cout);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

int qt_sgemm_tn(const SgemmArgs& a_, hipStream_t stream) {
    if (a_.M <= 0 || a_.N <= 0) return QT_OK;
    QT_CHECK_ARG(a_.batch >= 1 && a_.batch <= SG_MAX_BATCH && a_.n_groups >= 0 && a_.n_groups <= SG_MAX_GROUPS &&
                     !(a_.batch > 1 && a_.n_groups > 0),
                 "qt_sgemm_tn: batch %d / row groups %d unsupported", a_.batch, a_.n_groups);
    for (int g = 0; g < a_.n_groups; ++g)
        QT_CHECK_ARG((a_.group_m_end[g] % 128 == 0 || g == a_.n_groups - 1) && (g == 0 || a_.group_m_end[g] >= a_.group_m_end[g - 1]),
                     "qt_sgemm_tn: row group %d ends at %d (boundaries must be ascending multiples of 128)", g, a_.group_m_end[g]);
    static const int fast_interior = [] {
        const char* e = getenv("QT_SGEMM_INTERIOR");
        return (e && atoi(e) == 0) ? 0 : 1;
    }();
    SgemmArgs a = a_;
    a.fast_interior = fast_interior;
    const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
    // Latency-bound shape (fewer 128x128 tiles than CUs, long k): keep the MFMA-dense 128x128
    // tile and split k over workgroups to fill the chip; slabs are reduced in ascending order.
    if (a.split_ws && t128 < 384 && a.kdim >= 512) {
        // Target workgroups per product.  Rounds 1-3: 2048 (~8 per CU), tuned on the f32 Cholesky chain at K = 14336
        // (44.2 / 39.5 / 37.3 / 37.9 ms at 512 / 1024 / 2048 / 3072), where these products carried the K^3.  Since the
        // bf16x3 block-row products took that over, only SHORT products come here, and since round 4 they usually come
        // as a batch that fills the chip by itself: one round of the CUs is the better target (K = 4096: single chain
        // 3.92 -> 3.84 ms, three batched 5.94 -> 5.62, ten batched 12.7 -> 11.1; K = 8192 / 14336 unchanged) -- a
        // quarter of the slab traffic.  (The split is part of the bits: one value for single and batched calls.)
        static const int target_wgs = [] {
            const char* e = getenv("QT_SGEMM_SPLIT_TARGET");
            const int v = e ? atoi(e) : 256;
            return v < 256 ? 256 : v;
        }();
        static const int min_chunk = [] {
            const char* e = getenv("QT_SGEMM_SPLIT_MIN_CHUNK");
            const int v = e ? atoi(e) : 128;
            return v < 64 ? 64 : v;
        }();
        int splits = (int)(target_wgs / t128);
        if (splits > 32) splits = 32;
        if (splits < 1) splits = 1;
        int chunk = (a.kdim + splits - 1) / splits;
        chunk = (chunk + 63) / 64 * 64;  // whole BK steps
        if (chunk < min_chunk) chunk = min_chunk;
        splits = (a.kdim + chunk - 1) / chunk;
        const size_t need = (size_t)splits * a.M * a.N * sizeof(float);
        // (the decision depends on ONE problem's shape only: a batch takes the path -- hence the bits -- of its members)
        if (splits >= 2 && need <= a.split_ws_bytes && (a.batch <= 1 || (size_t)a.bs_split * sizeof(float) >= need)) {
            SgemmArgs part = a;
            part.Cin = nullptr;
            part.ldcin = 0;
            part.Cout = a.split_ws;
            part.ldcout = a.N;
            part.bsCin = 0;
            part.bsCout = a.bs_split;
            part.mode = SG_MODE_SET;
            part.k_chunk = chunk;
            part.n_splits = splits;
            const int rc = launch<128, 128>(part, stream, splits);
            if (rc) return rc;
            return qt_splitk_reduce(a.split_ws, splits, a.M, a.N, a.Cin, a.ldcin, a.Cout, a.ldcout, a.mode, stream, a.batch,
                                    a.bs_split, a.bsCin, a.bsCout);
        }
    }
    // MODE_SUB products with enough 128x128 tiles: the LDS-DMA ring kernel, persistent over its tiles
    // (both knobs are read per call: the tests and A/B tools switch them inside one process)
    const char* ring_env = getenv("QT_SGEMM_RING");
    const int use_ring = (ring_env && atoi(ring_env) == 0) ? 0 : 1;
    const char* ring_min_env = getenv("QT_SGEMM_RING_MIN_TILES");
    // from how many 128x128 tiles on: 384 in round 3 (1.5 rounds of the CUs); round 4, with the chains batched and three
    // layers in flight: 32 -- the persistent ring workgroups also win on the near updates and the short folds (bench
    // 75.9-76.4 -> 74.8 ms/step over 384 / 192 / 128 / 64 / 32: 75.9 / 75.5 / 75.4 / 75.0 / 74.8; K = 14336 sweep
    // 12.7 -> 11.9 ms alone).  Bit-identical to the register-staged kernel either way.
    const int ring_min_tiles = ring_min_env ? atoi(ring_min_env) : 32;
    if (use_ring && a.mode == SG_MODE_SUB && a.k_mode == SG_K_FULL && a.k_chunk == 0 && a.kdim >= 4 * RBK &&
        a.kdim % (2 * RBK) == 0 && (a.chain_len == 0 || (a.chain_len % (2 * RBK) == 0 && a.kdim % a.chain_len == 0)) &&
        a.kdim % RBK == 0 && (a.chain_len == 0 || a.chain_len % RBK == 0) && a.M % RBM == 0 && a.N % RBN == 0 &&
        (size_t)40 * (size_t)(a.ldcin > a.ldcout ? a.ldcin : a.ldcout) * 4 < ((size_t)1 << 31) && a.lda % 4 == 0 &&
        a.ldb % 4 == 0 && (((uintptr_t)a.A | (uintptr_t)a.B) & 15) == 0 &&
        (size_t)RBK * (size_t)(a.lda > a.ldb ? a.lda : a.ldb) * 4 < ((size_t)1 << 32) && (!a.upper_only || a.M == a.N)) {
        RingArgs r;
        r.A = a.A; r.lda = a.lda; r.B = a.B; r.ldb = a.ldb;
        r.Cin = a.Cin; r.ldcin = a.ldcin; r.Cout = a.Cout; r.ldcout = a.ldcout;
        r.M = a.M; r.N = a.N; r.kdim = a.kdim;
        r.chain_len = a.chain_len > 0 ? a.chain_len : a.kdim;
        const int tm = (a.M + RBM - 1) / RBM, tn = (a.N + RBN - 1) / RBN;
        r.tiles_n = tn;
        r.upper_only = a.upper_only;
        r.n_tiles = a.upper_only ? tn * (tn + 1) / 2 : tm * tn;
        r.batch = a.batch > 1 ? a.batch : 1;
        r.bsA = a.bsA; r.bsB = a.bsB; r.bsCin = a.bsCin; r.bsCout = a.bsCout;
        r.n_groups = a.n_groups;
        r.group_bsB = a.group_bsB;
        for (int g = 0; g < SG_MAX_GROUPS; ++g) r.group_m_end[g] = a.group_m_end[g];
        if (r.n_tiles >= ring_min_tiles) {
            const int all = r.n_tiles * r.batch;
            const int grid = all >= 2 * 256 ? 2 * 256 : all;      // two workgroups per CU
            hipLaunchKernelGGL(sgemm_ring_kernel, dim3(grid), dim3(512), 0, stream, r);
            QT_LAUNCH_CHECK();
            return QT_OK;
        }
    }
    // enough 128x128 tiles for ~1.5 rounds on 256 CUs, else 64x64 tiles for 4x the workgroups
    if (t128 >= 384) return launch<128, 128>(a, stream);
    return launch<64, 64>(a, stream);
}

// C-ABI face of the fp32 TN GEMM (tests and micro-benchmarks; the hot path calls qt_sgemm_tn directly).
extern "C" size_t qt_sgemm_tn_f32_workspace_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return (size_t)32 * M * N * 4 + 256;
}

extern "C" int qt_sgemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* Cin,
                               int64_t ldcin, float* Cout, int64_t ldcout, int M, int N, int kdim, int skip_zero_k,
                               int mode, int allow_split_k, void* workspace, size_t workspace_bytes,
                               qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(A && B && Cout && M > 0 && N > 0 && kdim >= 0, "qt_sgemm_tn_f32: bad arguments");
    QT_CHECK_ARG(mode == SG_MODE_SUB || mode == SG_MODE_SET || mode == SG_MODE_NEG, "qt_sgemm_tn_f32: mode %d", mode);
    QT_CHECK_ARG(mode != SG_MODE_SUB || Cin, "qt_sgemm_tn_f32: MODE_SUB needs Cin");
    SgemmArgs g;
    g.A = A; g.lda = lda;
    g.B = B; g.ldb = ldb;
    g.Cin = Cin; g.ldcin = ldcin;
    g.Cout = Cout; g.ldcout = ldcout;
    g.M = M; g.N = N; g.kdim = kdim;
    g.k_mode = skip_zero_k ? SG_K_FROM_N0 : SG_K_FULL;
    g.mode = mode;
    if (allow_split_k && workspace) {
        g.split_ws = (float*)qt_align_up((size_t)workspace, 256);
        g.split_ws_bytes = workspace_bytes >= 256 ? workspace_bytes - 256 : 0;
    }
    return qt_sgemm_tn(g, stream);
}
