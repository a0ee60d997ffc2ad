// See gemm3_tn.h.  The kernel is xtx.hip's pipeline (unit = 16 k-rows x (256 A + 256 B columns) = 16 KiB,
// ring of 8 units, LDS-DMA of unit u+6 issued in phase u behind a counted vmcnt(10), waves 4-7 one barrier
// behind waves 0-3; hazard analysis at xtx_kernel) with two changes:
//   * the A and B panels of a unit come from two plane sets, and the unit sequence of an item walks
//     k-chunk (128 rows) -> plane product (6) -> 8 units, so one 8-phase body stays inside one plane pair;
//   * the epilogue applies the tile to C (C -= acc / C = acc) or stores a slab for the ordered reduction.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>

#include "gemm3_tn.h"
#include "ring_pipe.h"

namespace {

constexpr int BT = 256;
constexpr int UT = 16;
constexpr int UNIT_BYTES = UT * 2 * BT * 2;
constexpr int RING = 8;
constexpr int NTHREADS = 512;
constexpr int NUM_CU = 256;
constexpr int CH_ROWS = G3_CHUNK_ROWS;          // k rows per chunk (64)
constexpr int CH_UNITS = 6 * (CH_ROWS / UT);    // 24 units: 6 plane products x 4 units
static_assert(CH_ROWS / UT == 4, "the phase code below recomputes the source pointers every 4 units");
// plane (0 = hi, 1 = mid, 2 = lo) of the A / B operand in product pr, two bits each, smallest product first:
//   pr:  0      1      2       3       4       5
//   A :  lo     hi     mid     mid     hi      hi
//   B :  hi     lo     mid     hi      mid     hi
constexpr unsigned PA_BITS = 2u | (0u << 2) | (1u << 4) | (1u << 6) | (0u << 8) | (0u << 10);
constexpr unsigned PB_BITS = 0u | (2u << 2) | (1u << 4) | (0u << 6) | (1u << 8) | (0u << 10);

struct G3Params {
    const char* Apl;
    const char* Bpl;
    int64_t plane_bytes, plane_bytesB;
    int64_t ld2, ld2B;    // row pitch in bytes (A planes / B planes)
    int64_t rowA0, rowB0;
    int colA0, colB0, colmax, colmaxB;
    int M, N;
    float* C;
    int64_t ldc;
    float* slabs;
    const G3Item* items;
    int64_t bsA_bytes, bsB_bytes, bsC, bsSlabs;   // problem blockIdx.y of a batch
};

template <int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void gemm3_kernel(G3Params p) {
    constexpr int LEAD = 6;
    __shared__ __attribute__((aligned(16))) char ring[RING * UNIT_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;
    const bool group_b = wave >= 4;

    if (blockIdx.y) {       // problem b of a batch of identical shapes
        p.Apl += (size_t)blockIdx.y * p.bsA_bytes;
        p.Bpl += (size_t)blockIdx.y * p.bsB_bytes;
        p.C += (size_t)blockIdx.y * p.bsC;
        p.slabs += (size_t)blockIdx.y * p.bsSlabs;
    }
    const G3Item* itp = p.items + blockIdx.x;
    const int it_tile = __builtin_amdgcn_readfirstlane(itp->tile);
    const int c_lo = __builtin_amdgcn_readfirstlane(itp->c_lo);
    const int c_hi = __builtin_amdgcn_readfirstlane(itp->c_hi);
    const int slab_idx = __builtin_amdgcn_readfirstlane(itp->slab);
    const int ti = it_tile >> 16, tj = it_tile & 0xFFFF;
    const int nu = (c_hi - c_lo) * CH_UNITS;   // a multiple of 24 (items are >= 2 chunks: nu >= 48)

    // staging geometry: as xtx_kernel (wave w fills k rows 4*(w&3)..+3 of column group w>>2; XOR swizzle of
    // the 16-B chunk index on the SOURCE address)
    const int rsub = lane >> 4;
    const int lch = (lane & 15) ^ (rsub << 2);
    const int trow = 4 * (wave & 3) + rsub;
    int colA = p.colA0 + ti * BT + (wave >> 2) * 128 + lch * 8;
    int colB = p.colB0 + tj * BT + (wave >> 2) * 128 + lch * 8;
    colA = colA > p.colmax - 8 ? p.colmax - 8 : colA;   // edge tiles: clamp (masked at the store)
    colB = colB > p.colmaxB - 8 ? p.colmaxB - 8 : colB;
    const unsigned voffA = (unsigned)((size_t)trow * (size_t)p.ld2 + (size_t)colA * 2);
    const unsigned voffB = (unsigned)((size_t)trow * (size_t)p.ld2B + (size_t)colB * 2);
    const unsigned ring_lds = (unsigned)(size_t)(QT_LDS char*)ring;
    const unsigned dst_wave =
        __builtin_amdgcn_readfirstlane(ring_lds + (wave >> 2) * 4096 + (wave & 3) * 1024);

    // unit i of the item -> (k chunk, plane product, 16-row slice): scalar source pointers of both panels
    const int64_t ustride = (int64_t)UT * p.ld2, ustrideB = (int64_t)UT * p.ld2B;
    auto unit_src = [&](int i, const char*& a, const char*& b) {
        const int g = i >> 2, j = i & 3;
        const int c = g / 6, pr = g - 6 * c;
        const int64_t plA = (PA_BITS >> (2 * pr)) & 3u, plB = (PB_BITS >> (2 * pr)) & 3u;
        const int64_t row = (int64_t)(c_lo + c) * CH_ROWS + j * UT;
        a = p.Apl + plA * p.plane_bytes + (p.rowA0 + row) * p.ld2;
        b = p.Bpl + plB * p.plane_bytesB + (p.rowB0 + row) * p.ld2B;
    };
    auto issue = [&](int i, int slot) {
        const char *a, *b;
        unit_src(i, a, b);
        const unsigned d = dst_wave + (unsigned)slot * UNIT_BYTES;
        glds16_pair2(voffA, voffB, a, b, d, d + 8192);
    };
    // steady state: running pointers; a body of 8 phases issues units u+6 .. u+13, the plane pair changes
    // at units u+8 and u+12 (phase slots 2 and 6), where the pointers are recomputed
    const char *runA = nullptr, *runB = nullptr;

    // fragment read geometry: as xtx_kernel
    const int g = lane >> 4, il = lane & 15, q = il >> 2, pp = il & 3;
    const int rowpart = (8 * (g >> 1) + q) * 256 + 32 * (g & 1) + 8 * pp;
    int aoff[4], boff[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) aoff[mi] = wave_m * 4096 + rowpart + 64 * (mi ^ q);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
        boff[ni] = (2 + (wave_n >> 1)) * 4096 + rowpart + 64 * ((((wave_n & 1) << 1) + ni) ^ q);

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    s16x8 fa[4], fb[2];
    auto drain_wait = [&](int u) {
        const int later = nu - u - 2;
        if (later >= 5) wait_vmcnt<10>();
        else if (later == 4) wait_vmcnt<8>();
        else if (later == 3) wait_vmcnt<6>();
        else if (later == 2) wait_vmcnt<4>();
        else if (later == 1) wait_vmcnt<2>();
        else wait_vmcnt<0>();
    };
    auto phase = [&](auto slot_c, auto steady_c, int u) {
        constexpr int S = decltype(slot_c)::value;
        constexpr bool STEADY = decltype(steady_c)::value;
        constexpr int ISLOT = (S + LEAD) & (RING - 1);
        const char* base = ring + S * UNIT_BYTES;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa[mi] = tr_load8(base + aoff[mi]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fb[ni] = tr_load8(base + boff[ni]);
        if (STEADY) {
            if (S == 2 || S == 6) unit_src(u + LEAD, runA, runB);
            const unsigned d = dst_wave + (unsigned)ISLOT * UNIT_BYTES;
            glds16_pair2(voffA, voffB, runA, runB, d, d + 8192);
            runA += ustride;
            runB += ustrideB;
            wait_vmcnt<10>();
        } else if (u + LEAD < nu) {
            issue(u + LEAD, ISLOT);
            wait_vmcnt<10>();
        } else {
            drain_wait(u);
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[mi][ni] = mfma16<false>(fa[mi], fb[ni], acc[mi][ni]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto body8 = [&](auto steady_c, int u) {
        phase(std::integral_constant<int, 0>{}, steady_c, u);
        phase(std::integral_constant<int, 1>{}, steady_c, u + 1);
        phase(std::integral_constant<int, 2>{}, steady_c, u + 2);
        phase(std::integral_constant<int, 3>{}, steady_c, u + 3);
        phase(std::integral_constant<int, 4>{}, steady_c, u + 4);
        phase(std::integral_constant<int, 5>{}, steady_c, u + 5);
        phase(std::integral_constant<int, 6>{}, steady_c, u + 6);
        phase(std::integral_constant<int, 7>{}, steady_c, u + 7);
    };

    if (nu > 0) {
#pragma unroll
        for (int i = 0; i < LEAD; ++i) issue(i, i);   // nu >= 24 > LEAD
        wait_vmcnt<2 * (LEAD - 1)>();
        __builtin_amdgcn_s_barrier();
        if (group_b) __builtin_amdgcn_s_barrier();   // stagger: group B runs one interval behind
        __builtin_amdgcn_sched_barrier(0);
        unit_src(LEAD, runA, runB);
        int u = 0;
        for (; u + 8 + LEAD <= nu; u += 8) body8(std::true_type{}, u);
        for (; u + 8 <= nu; u += 8) body8(std::false_type{}, u);
        if (!group_b) __builtin_amdgcn_s_barrier();   // pairs with group B's last barrier
        wait_vmcnt<0>();
    }

    // ---- epilogue ----
    const int jl = lane & 31, ih = 4 * (lane >> 5);
    if (slab_idx < 0) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int gj = tj * BT + wave_n * 64 + ni * 32 + jl;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gi = ti * BT + wave_m * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + ih;
                    if (gi < p.M && gj < p.N) {
                        float* dst = p.C + (size_t)gi * p.ldc + gj;
                        if (MODE == G3_SUB) *dst = *dst - acc[mi][ni][r];
                        else if (MODE == G3_ADD) *dst = *dst + acc[mi][ni][r];
                        else *dst = acc[mi][ni][r];
                    }
                }
            }
    } else {
        float* slab = p.slabs + (size_t)slab_idx * (size_t)(BT * BT);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i_loc = wave_m * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + ih;
                    const int j_loc = wave_n * 64 + ni * 32 + jl;
                    slab[i_loc * BT + j_loc] = acc[mi][ni][r];
                }
    }
}

// C[tile] (-)= sum of the tile's slabs in table order (one float4 per thread per step)
__global__ __launch_bounds__(256) void gemm3_reduce_kernel(const float* __restrict__ slabs,
                                                           const G3Red* __restrict__ red, float* C, int64_t ldc,
                                                           int M, int N, int mode, int64_t bsSlabs, int64_t bsC) {
    if (blockIdx.z) {
        slabs += (size_t)blockIdx.z * bsSlabs;
        C += (size_t)blockIdx.z * bsC;
    }
    const G3Red t = red[blockIdx.x];
    const int ti = t.tile >> 16, tj = t.tile & 0xFFFF;
    const int part = blockIdx.y;   // 16 parts of 16 rows
    const size_t tile_elems = (size_t)BT * BT;
    for (int e = threadIdx.x; e < 16 * (BT / 4); e += blockDim.x) {
        const int i_loc = part * 16 + e / (BT / 4);
        const int j_loc = (e % (BT / 4)) * 4;
        const int gi = ti * BT + i_loc, gj = tj * BT + j_loc;
        if (gi >= M || gj >= N) continue;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < t.count; ++sp) {
            const f32x4 v = *(const f32x4*)(slabs + (size_t)(t.first + sp) * tile_elems + (size_t)i_loc * BT + j_loc);
            s += v;
        }
        float* dst = C + (size_t)gi * ldc + gj;
        if (gj + 3 < N && (ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0) {
            if (mode == G3_SUB) {
                f32x4 o = *(f32x4*)dst;
                o -= s;
                *(f32x4*)dst = o;
            } else if (mode == G3_ADD) {
                f32x4 o = *(f32x4*)dst;
                o += s;
                *(f32x4*)dst = o;
            } else {
                *(f32x4*)dst = s;
            }
        } else {
            for (int c = 0; c < 4 && gj + c < N; ++c)
                dst[c] = (mode == G3_SUB) ? dst[c] - s[c] : (mode == G3_ADD) ? dst[c] + s[c] : s[c];
        }
    }
}

__device__ __forceinline__ unsigned short bf16_bits_rne(float v) {
    return __builtin_bit_cast(unsigned short, (__bf16)v);
}

__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src, int64_t ld_src, int cols,
                                                     unsigned short* __restrict__ planes, int64_t ld_pl,
                                                     int64_t plane_stride, int mask_upper, int row_g0, int col_g0,
                                                     int64_t bs_src, int64_t bs_planes) {
    if (blockIdx.z) {
        src += (size_t)blockIdx.z * bs_src;
        planes += (size_t)blockIdx.z * bs_planes;
    }
    const int c4 = (blockIdx.x * 256 + threadIdx.x) * 4;
    const int r = blockIdx.y;
    if (c4 >= cols) return;
    f32x4 v = *(const f32x4*)(src + (size_t)r * ld_src + c4);
    typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
    u16x4 hi, mid, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = v[e];
        if (mask_upper && col_g0 + c4 + e > row_g0 + r) x = 0.0f;
        const unsigned short h = bf16_bits_rne(x);
        const float r1 = x - qt_bf16_to_f32(h);            // exact
        const unsigned short m = bf16_bits_rne(r1);
        const float r2 = r1 - qt_bf16_to_f32(m);           // exact
        hi[e] = h;
        mid[e] = m;
        lo[e] = bf16_bits_rne(r2);
    }
    unsigned short* dst = planes + (size_t)r * ld_pl + c4;
    *(u16x4*)dst = hi;
    *(u16x4*)(dst + plane_stride) = mid;
    *(u16x4*)(dst + 2 * plane_stride) = lo;
}

}  // namespace

int qt_gemm3_launch(const G3Args& a, hipStream_t stream) {
    if (a.n_items <= 0) return QT_OK;
    QT_CHECK_ARG(a.ld % 8 == 0 && a.colmax >= 8, "qt_gemm3_launch: plane pitch %lld must be a multiple of 8", (long long)a.ld);
    QT_CHECK_ARG(((uintptr_t)a.Apl & 15) == 0 && ((uintptr_t)a.Bpl & 15) == 0 && (a.plane_stride % 8) == 0,
                 "qt_gemm3_launch: planes must be 16-byte aligned");
    // per-lane source offsets are 32-bit: 16 rows of one unit must span < 4 GiB
    QT_CHECK_ARG((uint64_t)a.ld * 2 * UT + (uint64_t)a.colmax * 2 < ((uint64_t)1 << 32), "qt_gemm3_launch: pitch too large");
    G3Params p;
    p.Apl = (const char*)a.Apl;
    p.Bpl = (const char*)a.Bpl;
    p.plane_bytes = a.plane_stride * 2;
    p.ld2 = a.ld * 2;
    p.plane_bytesB = (a.plane_strideB ? a.plane_strideB : a.plane_stride) * 2;
    p.ld2B = (a.ldB ? a.ldB : a.ld) * 2;
    p.colmaxB = a.colmaxB ? a.colmaxB : a.colmax;
    QT_CHECK_ARG(p.ld2B % 16 == 0 && p.colmaxB >= 8 && (p.plane_bytesB % 16) == 0 &&
                     (uint64_t)p.ld2B * UT + (uint64_t)p.colmaxB * 2 < ((uint64_t)1 << 32),
                 "qt_gemm3_launch: B-side pitch %lld / columns %d unsupported", (long long)(p.ld2B / 2), p.colmaxB);
    p.rowA0 = a.rowA0;
    p.rowB0 = a.rowB0;
    p.colA0 = a.colA0;
    p.colB0 = a.colB0;
    p.colmax = a.colmax;
    p.M = a.M;
    p.N = a.N;
    p.C = a.C;
    p.ldc = a.ldc;
    p.slabs = a.slabs;
    p.items = a.items;
    const int nb = a.batch > 1 ? a.batch : 1;
    QT_CHECK_ARG(nb <= 65535 && (a.bsApl % 8) == 0 && (a.bsBpl % 8) == 0, "qt_gemm3_launch: batch %d / plane strides unsupported", nb);
    p.bsA_bytes = a.bsApl * 2;
    p.bsB_bytes = a.bsBpl * 2;
    p.bsC = a.bsC;
    p.bsSlabs = a.bsSlabs;
    const dim3 grid(a.n_items, nb);
    if (a.mode == G3_SUB) hipLaunchKernelGGL((gemm3_kernel<G3_SUB>), grid, dim3(NTHREADS), 0, stream, p);
    else if (a.mode == G3_ADD) hipLaunchKernelGGL((gemm3_kernel<G3_ADD>), grid, dim3(NTHREADS), 0, stream, p);
    else hipLaunchKernelGGL((gemm3_kernel<G3_SET>), grid, dim3(NTHREADS), 0, stream, p);
    QT_LAUNCH_CHECK();
    if (a.n_red > 0) {
        hipLaunchKernelGGL(gemm3_reduce_kernel, dim3(a.n_red, 16, nb), dim3(256), 0, stream, (const float*)a.slabs, a.red,
                           a.C, a.ldc, a.M, a.N, a.mode, a.bsSlabs, a.bsC);
        QT_LAUNCH_CHECK();
    }
    return QT_OK;
}

int qt_split3_launch(const float* src, int64_t ld_src, int rows, int cols, unsigned short* planes, int64_t ld_pl,
                     int64_t plane_stride, int mask_upper, int row_g0, int col_g0, hipStream_t stream, int batch,
                     int64_t bs_src, int64_t bs_planes) {
    if (rows <= 0 || cols <= 0) return QT_OK;
    QT_CHECK_ARG(cols % 4 == 0 && ld_src % 4 == 0 && ld_pl % 4 == 0 && plane_stride % 4 == 0 &&
                     ((uintptr_t)src & 15) == 0 && ((uintptr_t)planes & 7) == 0,
                 "qt_split3_launch: columns / pitches must be multiples of 4 and the pointers aligned");
    QT_CHECK_ARG(bs_src % 4 == 0 && bs_planes % 4 == 0, "qt_split3_launch: batch strides must be multiples of 4");
    hipLaunchKernelGGL(split3_kernel, dim3((cols / 4 + 255) / 256, rows, batch > 1 ? batch : 1), dim3(256), 0, stream, src,
                       ld_src, cols, planes, ld_pl, plane_stride, mask_upper, row_g0, col_g0, bs_src, bs_planes);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

// k ranges are counted in chunks of G3_CHUNK_ROWS rows; a 256-column tile column tj of a lower-triangular B
// starts at chunk tj * (256 / G3_CHUNK_ROWS)
long g3_row_chunks(int Tm, int Tn, int c_end, int tri) {
    constexpr int TRI_STEP = 256 / G3_CHUNK_ROWS;
    long total = 0;
    for (int tj = 0; tj < Tn; ++tj) total += (long)Tm * std::max(0, c_end - (tri ? TRI_STEP * tj : 0));
    return total;
}

void g3_plan_row(int Tm, int Tn, int c_end, int tri, std::vector<G3Item>& items, std::vector<G3Red>& red, int target_items) {
    constexpr int TRI_STEP = 256 / G3_CHUNK_ROWS;
    items.clear();
    red.clear();
    const long total = g3_row_chunks(Tm, Tn, c_end, tri);
    const int cap = target_items > 0 && target_items < NUM_CU ? target_items : NUM_CU;   // items per product (at most one round of the CUs)
    // smallest piece length (in chunks, >= 2) with which all pieces fit one round of the 256 CUs
    auto count_items = [&](int per) {
        long n_items = 0;
        for (int tj = 0; tj < Tn; ++tj) {
            const int n = c_end - (tri ? TRI_STEP * tj : 0);
            if (n > 0) n_items += (long)Tm * ((n + per - 1) / per);
        }
        return n_items;
    };
    int per = std::max(2, (int)((total + cap - 1) / cap));
    while (count_items(per) > cap && per < c_end) ++per;   // more tiles than the cap: one item per tile
    int next_slab = 0;
    for (int ti = 0; ti < Tm; ++ti)
        for (int tj = 0; tj < Tn; ++tj) {
            const int lo = tri ? TRI_STEP * tj : 0, n = c_end - lo;
            if (n <= 0) continue;
            const int pieces = (n + per - 1) / per;
            const int tile = (ti << 16) | tj;
            if (pieces == 1) {
                items.push_back({tile, lo, c_end, -1});
                continue;
            }
            red.push_back({tile, next_slab, pieces, 0});
            for (int s = 0; s < pieces; ++s) {
                const int a = lo + (int)((long)n * s / pieces), b = lo + (int)((long)n * (s + 1) / pieces);
                items.push_back({tile, a, b, next_slab++});
            }
        }
    // longest items first (they all start in the first round; the order only matters beyond 256 items)
    std::stable_sort(items.begin(), items.end(),
                     [](const G3Item& x, const G3Item& y) { return (x.c_hi - x.c_lo) > (y.c_hi - y.c_lo); });
    // Deal the items to the XCDs (workgroup i of a launch runs on XCD i % 8; the <= 256 items of a product start
    // together and stream their k ranges in step): items that read the SAME k rows -- the same piece of different
    // tiles -- go to one XCD, where the A panel of a tile row is then fetched once for all its tile columns and a B
    // panel once for both tile rows.  The slab of an item and the reduction order do not change: same bits.
    // QT_G3_DEAL=0: the order above (rounds 2-3).
    static const bool deal = [] { const char* e = getenv("QT_G3_DEAL"); return !(e && atoi(e) == 0); }();
    if (deal && items.size() > 8) {
        std::vector<int> starts;
        for (const G3Item& it : items) starts.push_back(it.c_lo);
        std::sort(starts.begin(), starts.end());
        starts.erase(std::unique(starts.begin(), starts.end()), starts.end());
        const int P = (int)starts.size();                       // distinct k-range starts
        const int g = P >= 8 ? 1 : 8 / P;                       // XCDs per k range when there are fewer ranges than XCDs
        std::vector<std::vector<G3Item>> cls(8);
        for (const G3Item& it : items) {
            const int rank = (int)(std::lower_bound(starts.begin(), starts.end(), it.c_lo) - starts.begin());
            const int tj = it.tile & 0xFFFF;
            cls[(rank * g + (g > 1 ? tj % g : 0)) % 8].push_back(it);
        }
        std::vector<G3Item> dealt;
        dealt.reserve(items.size());
        for (size_t q = 0; dealt.size() < items.size(); ++q)
            for (int x = 0; x < 8; ++x)
                if (q < cls[x].size()) dealt.push_back(cls[x][q]);
        items.swap(dealt);
    }
}

// ---- C-ABI face (tests and micro-benchmarks; the hot path calls qt_gemm3_launch from cholesky.hip) ----
// C (-)= A^T B with A [k][lda], B [k][ldb] fp32, k a multiple of 128.  kind 0: every tile whole (C -= A^T B);
// kind 1: split along k into slabs (C = A^T B), as the inverse's block-row product.  Synchronous table upload.
static size_t g3_test_layout(int M, int N, int k, int64_t& ldp, size_t& planes_bytes, size_t& slabs_bytes) {
    ldp = (int64_t)qt_align_up((size_t)std::max(M, N), 256);
    planes_bytes = (size_t)3 * k * ldp * 2;
    slabs_bytes = (size_t)2 * NUM_CU * BT * BT * 4;
    return 2 * planes_bytes + slabs_bytes + (size_t)(1 << 20) + 1024;
}

extern "C" size_t qt_gemm3_tn_f32_workspace_bytes(int M, int N, int k) {
    if (M <= 0 || N <= 0 || k <= 0) return 0;
    int64_t ldp;
    size_t pb, sb;
    return g3_test_layout(M, N, k, ldp, pb, sb);
}

extern "C" int qt_gemm3_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M,
                               int N, int k, int kind, void* workspace, size_t workspace_bytes, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(A && B && C && M > 0 && N > 0 && k > 0 && k % 128 == 0, "qt_gemm3_tn_f32: k=%d must be a positive multiple of 128", k);
    QT_CHECK_ARG(M % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0, "qt_gemm3_tn_f32: M, N, lda, ldb must be multiples of 4");
    int64_t ldp;
    size_t pb, sb;
    const size_t need = g3_test_layout(M, N, k, ldp, pb, sb);
    if (!workspace || workspace_bytes < need) {
        qt_set_error("qt_gemm3_tn_f32: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    char* ws = (char*)qt_align_up((size_t)workspace, 256);
    unsigned short* Apl = (unsigned short*)ws;
    unsigned short* Bpl = (unsigned short*)(ws + pb);
    float* slabs = (float*)(ws + 2 * pb);
    char* tab = ws + 2 * pb + sb;
    QT_HIP(hipMemsetAsync(ws, 0, 2 * pb, stream));   // columns M..ldp / N..ldp of the planes
    int rc = qt_split3_launch(A, lda, k, M, Apl, ldp, (int64_t)k * ldp, 0, 0, 0, stream);
    if (rc) return rc;
    rc = qt_split3_launch(B, ldb, k, N, Bpl, ldp, (int64_t)k * ldp, 0, 0, 0, stream);
    if (rc) return rc;
    const int Tm = (M + BT - 1) / BT, Tn = (N + BT - 1) / BT, nch = k / CH_ROWS;
    std::vector<G3Item> items;
    std::vector<G3Red> red;
    if (kind == 0) {
        for (int ti = 0; ti < Tm; ++ti)
            for (int tj = 0; tj < Tn; ++tj) items.push_back({(ti << 16) | tj, 0, nch, -1});
    } else {
        // every tile cut into pieces of <= 2 chunks (bounded by the slab area)
        int next = 0;
        for (int ti = 0; ti < Tm; ++ti)
            for (int tj = 0; tj < Tn; ++tj) {
                const int pieces = std::min(nch / 2, std::max(1, 2 * NUM_CU / (Tm * Tn)));   // >= 2 chunks each
                red.push_back({(ti << 16) | tj, next, pieces, 0});
                for (int s = 0; s < pieces; ++s)
                    items.push_back({(ti << 16) | tj, (int)((long)nch * s / pieces), (int)((long)nch * (s + 1) / pieces), next++});
            }
        QT_CHECK_ARG((size_t)next * BT * BT * 4 <= sb, "qt_gemm3_tn_f32: too many tiles for the test face");
    }
    const size_t ib = items.size() * sizeof(G3Item), rb = red.size() * sizeof(G3Red);
    QT_CHECK_ARG(qt_align_up(ib, 256) + rb <= (size_t)(1 << 20), "qt_gemm3_tn_f32: table too large for the test face");
    QT_HIP(hipStreamSynchronize(stream));
    QT_HIP(hipMemcpy(tab, items.data(), ib, hipMemcpyHostToDevice));
    if (rb) QT_HIP(hipMemcpy(tab + qt_align_up(ib, 256), red.data(), rb, hipMemcpyHostToDevice));
    G3Args g;
    g.Apl = Apl;
    g.Bpl = Bpl;
    g.plane_stride = (int64_t)k * ldp;
    g.ld = ldp;
    g.rowA0 = g.rowB0 = 0;
    g.colA0 = g.colB0 = 0;
    g.colmax = (int)ldp;
    g.M = M;
    g.N = N;
    g.C = C;
    g.ldc = ldc;
    g.mode = kind == 0 ? G3_SUB : G3_SET;
    g.slabs = slabs;
    g.items = (const G3Item*)tab;
    g.n_items = (int)items.size();
    g.red = (const G3Red*)(tab + qt_align_up(ib, 256));
    g.n_red = (int)red.size();
    return qt_gemm3_launch(g, stream);
}

// ---- a7 for fp32 activations (an fp32 checkpoint): G += X^T X with X fp32 [n_tokens, K] --------------------------
// Upstream accumulates `inp.float()`: for an fp32 model that is an fp32 Gram product.  The bf16 Gram
// kernel cannot take such activations without rounding them (8 significant bits); this entry point runs the product
// on the same three-plane machinery the Cholesky chain uses: the token chunk is split into hi / mid / lo bf16 planes
// (residual <= 2^-27 |x|) and every lower-triangular 256x256 tile accumulates its six plane products in one fp32 MFMA
// accumulator, added into G from the epilogue -- fp32-accurate, 6 bf16-MFMA flops per fp32 flop.
// Tokens are processed in chunks of XF_CHUNK rows (zero-padded to a multiple of 128) so the planes stay bounded.
constexpr int XF_CHUNK = 8192;

static int64_t xf_pitch(int K) { return (int64_t)qt_align_up((size_t)K, 256); }

extern "C" size_t qt_xtx_accumulate_f32_workspace_bytes(int64_t n_tokens, int K) {
    if (n_tokens <= 0 || K <= 0) return 0;
    const int nt = (K + BT - 1) / BT;
    return (size_t)3 * XF_CHUNK * xf_pitch(K) * 2 + qt_align_up((size_t)nt * (nt + 1) / 2 * sizeof(G3Item), 256) * 2 + 1024;
}

extern "C" int qt_xtx_accumulate_f32(const float* X, int64_t n_tokens, int K, int64_t ldx, float* G, void* workspace,
                                     size_t workspace_bytes, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(X && G && K > 0 && n_tokens >= 0 && ldx >= K, "qt_xtx_accumulate_f32: bad arguments");
    QT_CHECK_ARG(K % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)X & 15) == 0,
                 "qt_xtx_accumulate_f32: K, ldx must be multiples of 4 and X 16-byte aligned (K=%d)", K);
    if (n_tokens == 0) return QT_OK;
    const size_t need = qt_xtx_accumulate_f32_workspace_bytes(n_tokens, K);
    if (!workspace || workspace_bytes < need) {
        qt_set_error("qt_xtx_accumulate_f32: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    const int64_t ldp = xf_pitch(K);
    const int64_t plane_stride = (int64_t)XF_CHUNK * ldp;
    char* ws = (char*)qt_align_up((size_t)workspace, 256);
    unsigned short* planes = (unsigned short*)ws;
    const size_t planes_bytes = (size_t)3 * XF_CHUNK * ldp * 2;
    char* tab = ws + planes_bytes;
    const int nt = (K + BT - 1) / BT;
    const size_t tab_bytes = qt_align_up((size_t)nt * (nt + 1) / 2 * sizeof(G3Item), 256);
    // item tables (one per distinct chunk length: a full chunk, and the last one), pinned per (K, chunks)
    static std::mutex m;
    static std::map<std::pair<int, int>, G3Item*> tabs;
    auto table = [&](int nch) -> const G3Item* {
        std::lock_guard<std::mutex> lock(m);
        auto key = std::make_pair(K, nch);
        auto it = tabs.find(key);
        if (it != tabs.end()) return it->second;
        G3Item* pinned = nullptr;
        if (hipHostMalloc((void**)&pinned, tab_bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
        int n = 0;
        for (int ti = 0; ti < nt; ++ti)
            for (int tj = 0; tj <= ti; ++tj) pinned[n++] = {(ti << 16) | tj, 0, nch, -1};
        tabs[key] = pinned;
        return pinned;
    };
    // columns K..ldp of the planes are never written by the split (edge tiles read them): zero them once per call
    if (ldp != K) QT_HIP(hipMemsetAsync(planes, 0, planes_bytes, stream));
    int slot = 0;
    for (int64_t t0 = 0; t0 < n_tokens; t0 += XF_CHUNK) {
        const int rows = (int)std::min<int64_t>(XF_CHUNK, n_tokens - t0);
        const int rows_pad = (rows + 127) / 128 * 128;            // items are whole chunks of 64 rows, at least two
        if (rows_pad != rows) {
            // the rows behind a ragged last chunk may hold an earlier chunk's planes: zero them
            for (int q = 0; q < 3; ++q)
                QT_HIP(hipMemsetAsync(planes + q * plane_stride + (size_t)rows * ldp, 0, (size_t)(rows_pad - rows) * ldp * 2,
                                      stream));
        }
        int rc = qt_split3_launch(X + (size_t)t0 * ldx, ldx, rows, K, planes, ldp, plane_stride, 0, 0, 0, stream);
        if (rc) return rc;
        const int nch = rows_pad / CH_ROWS;
        const G3Item* host_tab = table(nch);
        if (!host_tab) {
            qt_set_error("qt_xtx_accumulate_f32: hipHostMalloc failed");
            return QT_ERR_HIP;
        }
        char* dev_tab = tab + (size_t)(slot & 1) * tab_bytes;     // two slots: a full-chunk and a last-chunk table
        ++slot;
        QT_HIP(hipMemcpyAsync(dev_tab, host_tab, (size_t)nt * (nt + 1) / 2 * sizeof(G3Item), hipMemcpyHostToDevice, stream));
        G3Args g;
        g.Apl = planes;
        g.Bpl = planes;
        g.plane_stride = plane_stride;
        g.ld = ldp;
        g.rowA0 = g.rowB0 = 0;
        g.colA0 = g.colB0 = 0;
        g.colmax = (int)ldp;
        g.M = K;
        g.N = K;
        g.C = G;
        g.ldc = K;
        g.mode = G3_ADD;
        g.slabs = nullptr;
        g.items = (const G3Item*)dev_tab;
        g.n_items = nt * (nt + 1) / 2;
        g.red = nullptr;
        g.n_red = 0;
        rc = qt_gemm3_launch(g, stream);
        if (rc) return rc;
    }
    return QT_OK;
}

// Host-only self-check of the block-row planner (no GPU needed; tests/test_gemm3_plan.py): every k chunk of
// every tile is covered exactly once, whole-range items are direct, partial ones own distinct consecutive
// slabs listed once in the reduction table, and one round of CUs holds all items whenever that is possible.
// Returns 0, or a negative code naming the violated property.
extern "C" int qt_gemm3_plan_check(int Tm, int Tn, int c_end, int tri, int* n_items_out, int* n_slabs_out,
                                   int* longest_out) {
    if (Tm <= 0 || Tn <= 0 || c_end <= 0) return -1;
    std::vector<G3Item> items;
    std::vector<G3Red> red;
    g3_plan_row(Tm, Tn, c_end, tri, items, red);
    const int step = 256 / G3_CHUNK_ROWS;
    std::vector<int> cover((size_t)Tm * Tn * c_end, 0);
    std::vector<int> slab_seen;
    int longest = 0, n_slabs = 0;
    for (const G3Item& it : items) {
        const int ti = it.tile >> 16, tj = it.tile & 0xFFFF;
        if (ti >= Tm || tj >= Tn || it.c_lo >= it.c_hi || it.c_hi > c_end) return -2;
        const int lo = tri ? step * tj : 0;
        if (it.c_lo < lo) return -3;
        for (int c = it.c_lo; c < it.c_hi; ++c) cover[((size_t)ti * Tn + tj) * c_end + c]++;
        longest = std::max(longest, it.c_hi - it.c_lo);
        if (it.slab < 0) {
            if (it.c_lo != lo || it.c_hi != c_end) return -4;   // a direct item must own its tile's whole range
        } else {
            if ((int)slab_seen.size() <= it.slab) slab_seen.resize(it.slab + 1, 0);
            if (slab_seen[it.slab]++) return -5;
            n_slabs = std::max(n_slabs, it.slab + 1);
        }
    }
    for (int ti = 0; ti < Tm; ++ti)
        for (int tj = 0; tj < Tn; ++tj)
            for (int c = 0; c < c_end; ++c) {
                const int want = (c >= (tri ? step * tj : 0)) ? 1 : 0;
                if (cover[((size_t)ti * Tn + tj) * c_end + c] != want) return -6;
            }
    std::vector<int> in_red(n_slabs, 0);
    for (const G3Red& r : red) {
        if (r.count < 2 || r.first < 0 || r.first + r.count > n_slabs) return -7;
        for (int q = 0; q < r.count; ++q)
            if (in_red[r.first + q]++) return -8;
        // the slabs of a reduction entry belong to its tile, in ascending k order
        int prev_hi = -1;
        for (int q = 0; q < r.count; ++q) {
            bool found = false;
            for (const G3Item& it : items)
                if (it.slab == r.first + q) {
                    if (it.tile != r.tile || (prev_hi >= 0 && it.c_lo != prev_hi)) return -9;
                    prev_hi = it.c_hi;
                    found = true;
                }
            if (!found) return -10;
        }
    }
    for (int q = 0; q < n_slabs; ++q)
        if (!in_red[q]) return -11;
    long tiles = 0;
    for (int tj = 0; tj < Tn; ++tj) tiles += (c_end > (tri ? step * tj : 0)) ? Tm : 0;
    if (tiles <= NUM_CU && (long)items.size() > NUM_CU) return -12;
    if (n_items_out) *n_items_out = (int)items.size();
    if (n_slabs_out) *n_slabs_out = n_slabs;
    if (longest_out) *longest_out = longest;
    return 0;
}
