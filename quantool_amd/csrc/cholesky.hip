// a8: U = chol((H + damp I)^-1, upper)   (upstream quantize_weight:
// cholesky -> cholesky_inverse -> cholesky(upper), reached through gptq.py:86 / base.py:161).
//
// One factorisation instead of three.  With A = flat-reverse(Hd) (A[i][j] = Hd[K-1-i][K-1-j],
// built by qt_hessian_prepare):   A = R^T R (R upper)  =>  Hd = (J R^T J)(J R J) = Ut Ut^T with
// Ut = J R^T J upper, so Hd^-1 = (Ut^-1)^T (Ut^-1) and, by uniqueness of the Cholesky factor,
//     U = Ut^-1 = J R^-T J = flat-reverse(Y),   Y = R^-T (lower).
// Cost 2/3 K^3 flops (vs 4/3 K^3) and every block recurrence below is a k-major "TN" product.  Two block
// sizes: NB = 128 for the latency-bound panel kernels, NBO / NBI = 256 for everything that carries the K^3.
//   potrf over NBO-wide outer blocks J, left-looking inside one (all in place in A):
//     block row j of J :  A[j, j:] -= sum_{p in J, p<j} R[p, j]^T R[p, j:]   (sgemm SUB, k <= NBO - 128)
//                         R_jj = chol(A_jj), 32x32 inverses                   (potf2_kernel)
//                         R[j, j+1:] = R_jj^-T A[j, j+1:]                     (trsm_rt_kernel)
//     outer level, small K (no product long enough for the bf16 MFMA) -- RIGHT-looking:
//       after block J  :  A[J1:, J1:] -= R[J, J1:]^T R[J, J1:]                (sgemm SUB, k = NBO, upper tiles only)
//     outer level, large K -- LEFT-looking, so the K^3/3 runs as long-k block-row products:
//       before block J :  A[J, J0:] -= R[:J0, J]^T R[:J0, J0:]                (gemm3 SUB, k = J0; sgemm where short)
//       after block J  :  bf16 plane copies of R[J, J0:]                      (split3_kernel)
//   R^-T by NBI-row block rows I (Y lower, so the k range of a tile starts at its first column):
//     inside I         :  Y_II by the 128-row recurrence  T = sum_{p in I, p<i} R[p, i]^T Y[p, I0:i];
//                         Y[i, I0:i] = -Dinv_i^T T ;  Y[i, i] = Dinv_i^T     (k <= NBI - 128)
//     left of I        :  T_I = sum_{p<I0} R[p, I]^T Y[p, :I0]               (gemm3 SET / sgemm SET, k = I0)
//                         Y[I, :I0] = -Y_II T_I = -(Y_II^T)^T T_I             (transpose + sgemm NEG, k = NBI)
//                         bf16 plane copies of Y[I, :I1]                      (split3_kernel, large K only)
// gemm3 (gemm3_tn.h) = fp32-accurate product on the bf16 MFMA from three bf16 planes per operand; sgemm
// (sgemm_tn.h) = the f32-MFMA fmaf chain.
#include <stdlib.h>
#include <string.h>

#include <map>
#include <tuple>
#include <vector>

#include "common.h"
#include "gemm3_tn.h"
#include "sgemm_tn.h"

namespace {

constexpr int NB = 128;
constexpr int NBO_MAX = 1024; // workspace is sized for the widest outer block
// Outer block of the factorisation / block-row height of the inverse (multiples of 128).  QT_CHOL_NBO /
// QT_CHOL_NBI force them (A/B runs); otherwise 256 on the f32 chain (measured at K = 14336 / 4096: 128: 40.5 /
// 4.30 ms, 256: 35.3 / 4.04, 512: 36.4 / 4.05, 1024: 38.5 / 4.40) and 512 where the bf16x3 products are in
// use -- two tile rows per product halve the k-split and the slab traffic (K = 14336 / 8192: 23.7 / 9.6 ms at
// 256 / 256, 23.0 / 9.2 at 512 / 256, 22.8 / 8.9 at 512 / 512; bench 84.5 -> 82.9 ms/step).
static int chol_block_env(const char* name) {
    const char* e = getenv(name);
    if (!e) return 0;
    int x = atoi(e) / 128 * 128;
    return x < 128 ? 128 : (x > NBO_MAX ? NBO_MAX : x);
}
// ---- bf16x3 block-row products (gemm3_tn.h) ---------------------------------------------------------
// Where a block-row product of the chain has enough k-chunks to fill the chip with xtx-class items, it
// runs on the bf16 MFMA from three-plane copies of R and Y (measured in the same process, k = 14080,
// 256 x 14080 outputs: 0.53 ms vs 0.90 ms on the f32 MFMA; tools/gemm3_bench.py).  The factorisation is
// then LEFT-looking at the outer level (block row J gathers  A[J, J:] -= R[:J0, J]^T R[:J0, J:]  in one
// long-k product: each element of A is read and written once, where the right-looking k = 256 update of
// the whole trailing matrix is bound by that read-modify-write), and the inverse's T_I product is the same
// shape.  QT_CHOL_G3=0 disables; QT_CHOL_G3_MIN_CHUNKS = 128-row k-chunks a product needs (256 =
// one per CU in rounds 2-3; K = 14336 / 8192 / 4096: 24.2 / 9.5 / 3.6 ms at 64...256, 24.6 at 640, 24.9 at 768; off: 34.0 /
// 10.7 / 3.6.  Round 4: 64 -- with the short f32 products split less (sgemm_tn.hip) and the chains batched, more of
// the K = 4096 steps pay on the bf16 MFMA: single chain 3.82 -> 3.70 ms, three batched 5.55 -> 5.23, ten batched
// 10.9 -> 10.4; K = 8192 / 14336 unchanged).  Item tables for every step are built once per (K, block sizes) in pinned host memory and
// uploaded with one async copy per call.
struct G3Step {
    int n_items = 0, n_red = 0;
    size_t item_off = 0, red_off = 0;   // bytes into the table blob
};
struct CholG3Plan {
    bool any = false;
    std::vector<G3Step> fac, inv;       // per outer block / per block row (n_items == 0: sgemm_tn)
    char* blob = nullptr;               // pinned
    size_t blob_bytes = 0;
    int max_slabs = 0;
};

static int chol_g3_min_chunks() {   // read per call: tests switch the path on for small K
    const char* on = getenv("QT_CHOL_G3");
    if (on && atoi(on) == 0) return 0;
    const char* e = getenv("QT_CHOL_G3_MIN_CHUNKS");
    const int x = e ? atoi(e) : 64;
    return x < 1 ? 1 : x;
}

// Items per product.  A product's k range is cut so that its items fill at most ONE round of the 256 CUs; below K = 8192
// the cap is 96 (QT_G3_TARGET_ITEMS forces a value for every K): those widths come as batches in practice (3 per Llama
// layer, 10 per Mixtral layer), a batch of n multiplies the items by n, and fewer, longer items mean fewer slabs to
// write and reduce.  K = 4096, cap 256 / 128 / 96 / 64: one chain 2.88 / 2.88 / 2.92 / 3.04 ms, three batched 4.15 / 3.92 /
// 3.84 / 3.59, ten batched 8.67 / 7.63 / 7.30 / 7.15; K = 8192 x 3: 13.4 / 13.6 / 13.8 / 13.4 (no gain: stays at 256).
// The cap is part of the bits and one value for single and batched calls (a batch stays bit-identical to its members).
static int chol_g3_target_items(int K) {
    const char* e = getenv("QT_G3_TARGET_ITEMS");
    if (e && atoi(e) > 0) return atoi(e);
    return K < 8192 ? 96 : 256;
}

static const CholG3Plan* chol_g3_plan(int K, int NBO, int NBI) {
    static std::mutex m;
    static std::map<std::tuple<int, int, int, int>, CholG3Plan*> plans;
    const int min_chunks = chol_g3_min_chunks();
    const int target_items = chol_g3_target_items(K);
    std::lock_guard<std::mutex> lock(m);
    const auto key = std::make_tuple(K, NBO, NBI, min_chunks * 1024 + target_items);
    auto it = plans.find(key);
    if (it != plans.end()) return it->second;
    CholG3Plan* pl = new CholG3Plan();
    std::vector<char> bytes;
    auto add = [&](int Tm, int Tn, int c_end, int tri) {
        G3Step st;
        if (min_chunks <= 0 || K % 8 != 0 || g3_row_chunks(Tm, Tn, c_end, tri) * G3_CHUNK_ROWS < (long)min_chunks * 128) return st;
        std::vector<G3Item> items;
        std::vector<G3Red> red;
        g3_plan_row(Tm, Tn, c_end, tri, items, red, target_items);
        st.n_items = (int)items.size();
        st.n_red = (int)red.size();
        st.item_off = qt_align_up(bytes.size(), 256);
        bytes.resize(st.item_off + items.size() * sizeof(G3Item));
        memcpy(bytes.data() + st.item_off, items.data(), items.size() * sizeof(G3Item));
        st.red_off = qt_align_up(bytes.size(), 256);
        bytes.resize(st.red_off + red.size() * sizeof(G3Red));
        if (!red.empty()) memcpy(bytes.data() + st.red_off, red.data(), red.size() * sizeof(G3Red));
        for (const G3Red& r : red) pl->max_slabs = pl->max_slabs > r.first + r.count ? pl->max_slabs : r.first + r.count;
        pl->any = true;
        return st;
    };
    for (int J0 = 0; J0 < K; J0 += NBO) {
        const int J1 = (K - J0 < NBO) ? K : J0 + NBO;
        pl->fac.push_back(J0 == 0 ? G3Step() : add((J1 - J0 + 255) / 256, (K - J0 + 255) / 256, J0 / G3_CHUNK_ROWS, 0));
    }
    for (int I0 = 0; I0 < K; I0 += NBI) {
        const int I1 = (K - I0 < NBI) ? K : I0 + NBI;
        pl->inv.push_back(I0 == 0 ? G3Step() : add((I1 - I0 + 255) / 256, (I0 + 255) / 256, I0 / G3_CHUNK_ROWS, 1));
    }
    if (pl->any) {
        pl->blob_bytes = qt_align_up(bytes.size(), 256);
        if (hipHostMalloc((void**)&pl->blob, pl->blob_bytes, hipHostMallocDefault) != hipSuccess) {
            pl->any = false;   // no pinned memory: stay on the f32 path
            pl->blob = nullptr;
        } else {
            memcpy(pl->blob, bytes.data(), bytes.size());
        }
    }
    if (!pl->any) {
        pl->blob_bytes = 0;
        pl->max_slabs = 0;
        for (G3Step& st : pl->fac) st = G3Step();
        for (G3Step& st : pl->inv) st = G3Step();
    }
    plans[key] = pl;
    return pl;
}
static size_t chol_g3_ws_bytes(const CholG3Plan* pl, int K) {
    if (!pl->any) return 0;
    return 2 * qt_align_up((size_t)3 * K * K * 2, 256) + (size_t)pl->max_slabs * 256 * 256 * 4 + pl->blob_bytes + 256;
}

// block sizes and item tables of one call
static const CholG3Plan* chol_blocks(int K, int& NBO, int& NBI) {
    const int eo = chol_block_env("QT_CHOL_NBO"), ei = chol_block_env("QT_CHOL_NBI");
    NBO = eo ? eo : 512;
    NBI = ei ? ei : 512;
    const CholG3Plan* pl = chol_g3_plan(K, NBO, NBI);
    if (pl->any) return pl;
    NBO = eo ? eo : 256;
    NBI = ei ? ei : 256;
    return chol_g3_plan(K, NBO, NBI);
}

constexpr int LDP = NB + 1;  // padded LDS leading dimension

// ---- panel kernels --------------------------------------------------------------------------
// All three keep a 128-long column in REGISTERS.  Registers cannot be indexed at run time, so a
// column is held as four 32-entry arrays: the 32 steps inside a sub-block are unrolled (static
// indices) and the sub-block loop stays a run-time loop with wave-uniform guards.
constexpr int LDT = NB + 4;  // LDS leading dimension with 16-byte aligned rows
constexpr int RD_STRIDE = NB * NB + 4 * 32 * 32;  // dense R block + four 32x32 sub-block inverses

#define QT_SEL4(kb, v0, v1, v2, v3) ((kb) == 0 ? (v0) : (kb) == 1 ? (v1) : (kb) == 2 ? (v2) : (v3))

// potf2: one workgroup (512 threads), Cholesky (upper, A = R^T R) of an n x n block (n <= 128),
// blocked by 32 inside the workgroup so that only 4 x 32 steps are sequential and everything
// else is f32 MFMA on LDS-resident tiles:
//   for kb = 0..3:
//     (i)   wave 0: factor the 32x32 diagonal sub-block with one column per lane in registers;
//           row c reaches the other lanes through v_readlane (no LDS, no barrier), then the
//           32-step back substitution for X_kb = R_kk^-1 the same way;
//     (ii)  R[kb, a] = X_kb^T A[kb, a]           for a > kb          (16 MFMAs per 32x32 tile)
//     (iii) A[a, a'] -= R[kb, a]^T R[kb, a']     for kb < a <= a'
//   P     : input block, upper triangle read, row stride ldp
//   Rout  : receives R (upper triangle written; strict lower untouched), stride ldr
//   Rd    : dense 128x128 row-major copy of R (zeros below the diagonal; identity padding >= n)
//           followed by D32[4][32][32], D32[b][k][i] = X_b[k][i]
//   info  : first non-positive pivot (1-based global index) if any; col0 = global offset
constexpr int LDA = NB + 4;

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// acc (+)= sum_k A[k][i] * B[k][n] over k = 0..31 for one 32x32 tile (TN form, LDS operands)
__device__ __forceinline__ void mma32_tn(const float* A, int lda, const float* B, int ldb, f32x16& acc,
                                         int l31, int h, bool negate_a) {
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
        float a = A[(2 * kk + h) * lda + l31];
        const float b = B[(2 * kk + h) * ldb + l31];
        if (negate_a) a = -a;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
}

// P and Rout may be the same block (the factorisation runs in place): no __restrict__ on them.
constexpr int POTF2_THREADS = 512, POTF2_WAVES = POTF2_THREADS / 64;   // 8 waves: the six trailing tiles of a step in one round
__global__ __launch_bounds__(POTF2_THREADS) void potf2_kernel(const float* P, int64_t ldp, int n,
                                                    float* Rout, int64_t ldr,
                                                    float* __restrict__ Rd, int32_t* info, int col0, int prio,
                                                    int64_t bsP, int64_t bsRd) {
    qt_set_chain_prio(prio);
    if (blockIdx.x) {      // problem b of a batch: one workgroup each
        P += (size_t)blockIdx.x * bsP;
        Rout += (size_t)blockIdx.x * bsP;
        Rd += (size_t)blockIdx.x * bsRd;
        info += blockIdx.x;
    }
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* As = sm;                // [NB][LDA]
    float* Xs = sm + NB * LDA;     // [4][32][32]  Xs[b][k][i] = X_b[k][i]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    // Symmetric fill from the upper triangle (identity padding beyond n).  The 16 row-major float4
    // loads of a thread are all issued before the first is used: a load -> wait -> store loop pays one
    // memory round trip per element (64 of them were ~40 % of this kernel).  Needs ldp % 4 == 0 (then
    // n % 4 == 0 too: n is 128 or K % 128) and a 16-byte aligned P; odd K takes the scalar loop.
    if ((ldp & 3) == 0 && (n & 3) == 0 && (((uintptr_t)P) & 15) == 0) {
        constexpr int NV = NB * NB / 4 / POTF2_THREADS;
        f32x4 v[NV];
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int e4 = tid + POTF2_THREADS * r;       // float4 index in the 128 x 32 grid
            const int i = e4 >> 5, j = (e4 & 31) * 4;
            const int ic = i < n ? i : n - 1, jc = j < n ? j : n - 4;
            v[r] = *(const f32x4*)(P + (size_t)ic * ldp + jc);
        }
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int e4 = tid + POTF2_THREADS * r;
            const int i = e4 >> 5, j = (e4 & 31) * 4;
            // only the upper triangle is ever read back: (i) takes R[c][i] from lane i's column (i > c), (ii) and
            // (iii) touch blocks on or above the diagonal, and what (i) / (iii) compute below it is never stored.
            // Rows are written whole (float4, conflict-free); the strict lower part just holds the input's values.
            f32x4 x;
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = (i < n && j + q < n) ? v[r][q] : (i == j + q ? 1.0f : 0.0f);
            *(f32x4*)(As + i * LDA + j) = x;
        }
    } else {
        for (int e = tid; e < NB * NB; e += POTF2_THREADS) {
            const int i = e / NB, j = e % NB;
            const int lo = i < j ? i : j, hi = i < j ? j : i;
            float v = (i == j) ? 1.0f : 0.0f;
            if (hi < n) v = P[(size_t)lo * ldp + hi];
            As[i * LDA + j] = v;
        }
    }
    __syncthreads();

    int bad = 0;
#pragma unroll 1
    for (int kb = 0; kb < 4; ++kb) {
        float* Akk = As + (32 * kb) * LDA + 32 * kb;
        // ---- (i) diagonal sub-block: factor AND inverse in one pass, wave 0 ----
        // Lanes 0..31 hold the columns of the 32x32 block, lanes 32..63 the columns of an identity.  One
        // right-looking step c -- rcj = col[c] / r_cc on every lane, then col[i] -= R[c][i] * rcj for i > c
        // with R[c][i] broadcast from lane i -- is the Cholesky step on the left half and the forward
        // substitution R^T Z = I on the right half: the same instructions, so Z = R^-T (= X^T) costs no
        // second 32-step loop.  Afterwards lane j < 32 holds R[i][j] in col[i], lane 32 + j holds X[j][c] in col[c].
        if (wave == 0) {
            // Both 32x32 blocks live in the f32 MFMA's accumulator layout (element (i, j) on lane j + 32 h(i), register
            // rho(i); i = (r & 3) + 8 (r >> 2) + 4 h): C1 = the diagonal block, C2 = an identity.  Step c:
            //     r[.] = C1[c][.] / sqrt(C1[c][c]),  z[.] = C2[c][.] / sqrt(C1[c][c])        (row c, one register of one half)
            //     C1[i][.] -= r[i] r[.],  C2[i][.] -= r[i] z[.]   for i > c                  (ONE MFMA each: a rank-1 update)
            // -- the Cholesky step on C1 and the forward substitution R^T Z = I on C2.  The A operand of both MFMAs is
            // the row itself: lane i + 32 k holds A[i][k], and row c sits on the lanes of half h(c) already, so
            // a = -r on those lanes (zero for i <= c and on the other half, whose k-slice then adds 0 * 0), b = r / z on
            // the same lanes.  Per element that is fmaf(-r_i, r_j, C1[i][j]), the operation of the lane-per-column loop this
            // replaces (31 v_readlane + fma pairs per step: ~6 us per 32x32 block; two MFMAs per step: ~2 us).
            f32x16 c1, c2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                c1[r] = Akk[i * LDA + l31];
                c2[r] = (i == l31) ? 1.0f : 0.0f;
            }
#pragma unroll
            for (int c = 0; c < 32; ++c) {
                const int rc = (c & 3) + 4 * (c >> 3);           // register of row c ...
                const int hc = (c >> 2) & 1;                     // ... on the lanes of this half
                float piv = readlane_f(c1[rc], c + 32 * hc);
                const bool isbad = !(piv > 0.0f);
                bad = (isbad && bad == 0 && 32 * kb + c < n) ? 32 * kb + c + 1 : bad;
                piv = isbad ? 1.0f : piv;
                float rs = __builtin_amdgcn_rsqf(piv);
                rs = rs * fmaf(-0.5f * piv * rs, rs, 1.5f);   // one Newton step: the inverse inherits rs
                const bool mine = h == hc;
                const float rr = c1[rc] * rs, zz = c2[rc] * rs;  // row c of R / of Z on the lanes of half hc
                c1[rc] = mine ? rr : c1[rc];
                c2[rc] = mine ? zz : c2[rc];
                if (c < 31) {
                    const float a = (mine && l31 > c) ? -rr : 0.0f;
                    // columns left of the diagonal hold whatever the caller's strict lower triangle held (never
                    // initialised): they must not meet the zero lanes of `a` (0 * NaN would reach finished rows)
                    const float b1 = (mine && l31 >= c) ? rr : 0.0f, b2 = mine ? zz : 0.0f;
                    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b2, c2, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                Akk[i * LDA + l31] = (i <= l31) ? c1[r] : 0.0f;                       // R_kk, zeros below
                Xs[(kb * 32 + l31) * 32 + i] = (i >= l31) ? c2[r] : 0.0f;             // X[j][c] = Z[c][j]: row c = i, column j = l31
            }
        }
        __syncthreads();
        if (kb == 3) break;
        // ---- (ii) R[kb, a] = X_kb^T A[kb, a], a > kb: one tile per wave (round robin) ----
        for (int a = kb + 1 + wave; a < 4; a += POTF2_WAVES) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
            float* Bt = As + (32 * kb) * LDA + 32 * a;
            mma32_tn(Xs + kb * 1024, 32, Bt, LDA, acc, l31, h, false);
#pragma unroll
            for (int r = 0; r < 16; ++r) Bt[((r & 3) + 8 * (r >> 2) + 4 * h) * LDA + l31] = acc[r];
        }
        __syncthreads();
        // ---- (iii) A[a, a'] -= R[kb, a]^T R[kb, a'], kb < a <= a' ----
        {
            int t = 0;
            for (int a = kb + 1; a < 4; ++a)
                for (int a2 = a; a2 < 4; ++a2, ++t) {
                    if ((t % POTF2_WAVES) != wave) continue;
                    float* Ct = As + (32 * a) * LDA + 32 * a2;
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = Ct[((r & 3) + 8 * (r >> 2) + 4 * h) * LDA + l31];
                    mma32_tn(As + (32 * kb) * LDA + 32 * a, LDA, As + (32 * kb) * LDA + 32 * a2, LDA, acc, l31, h,
                             true);
#pragma unroll
                    for (int r = 0; r < 16; ++r) Ct[((r & 3) + 8 * (r >> 2) + 4 * h) * LDA + l31] = acc[r];
                }
        }
        __syncthreads();
    }
    if (bad != 0 && tid == 0) atomicCAS(info, 0, col0 + bad);
    // R out (upper triangle; float4 groups that cross the diagonal carry zeros below it -- nothing reads the
    // strict lower part of a factored diagonal block) and the dense copy for trsm / the block inverses
    const bool vec_out = (ldr & 3) == 0 && (n & 3) == 0 && (((uintptr_t)Rout) & 15) == 0;
    for (int e4 = tid; e4 < NB * NB / 4; e4 += POTF2_THREADS) {
        const int i = e4 >> 5, j = (e4 & 31) * 4;
        f32x4 v = *(const f32x4*)(As + i * LDA + j);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (j + q >= i) ? v[q] : 0.0f;
        *(f32x4*)(Rd + i * NB + j) = v;
        if (i < n && j + 3 >= i && j < n) {
            if (vec_out) {
                *(f32x4*)(Rout + (size_t)i * ldr + j) = v;
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (j + q >= i && j + q < n) Rout[(size_t)i * ldr + j + q] = v[q];
            }
        }
    }
    for (int e4 = tid; e4 < 4 * 32 * 32 / 4; e4 += POTF2_THREADS) *(f32x4*)(Rd + NB * NB + 4 * e4) = *(const f32x4*)(Xs + 4 * e4);
}

// trsm: Z = R^-T B for a 128 x ncols panel on the f32 MFMA, by BLOCKED forward substitution
// with the 32x32 inverses X_b of R's diagonal sub-blocks (no divisions, no 128-long chain):
//     Z_b = X_b^T B_b ;   B_a -= R[b, a]^T Z_b   for a > b            (b = 0..3)
// Each wave owns 32 columns and keeps its four 32x32 row-block tiles in accumulators.  A result
// tile feeds the next MFMA directly as its B operand (accumulator register r holds row
// rho(r) = (r&3) + 8*(r>>2) on lanes 0-31 and rho(r)+4 on lanes 32-63, which is just a pairing
// of the k indices: the k order of these sums is free), so Z never round-trips through LDS;
// only the A operands (X_b, -R[b,a], k-major as stored) are read from LDS, one b32 per MFMA.
//   Rd : dense 128x128 R followed by D32[4][32][32], D32[b][k][i] = X_b[k][i] (potf2_kernel)
// B and Z may be the same panel (in place: a thread reads its columns before it writes them).
__global__ __launch_bounds__(256) void trsm_rt_kernel(const float* __restrict__ Rd, int n,
                                                      const float* B, int64_t ldb,
                                                      float* Z, int64_t ldz, int ncols, int prio, int64_t bsRd,
                                                      int64_t bsB) {
    qt_set_chain_prio(prio);
    if (blockIdx.y) {      // problem b of a batch
        Rd += (size_t)blockIdx.y * bsRd;
        B += (size_t)blockIdx.y * bsB;
        Z += (size_t)blockIdx.y * bsB;
    }
    extern __shared__ __attribute__((aligned(16))) float Rs[];  // [RD_STRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* D32 = Rs + NB * NB;
    const int l31 = lane & 31, h = lane >> 5;
    const int col = blockIdx.x * 128 + wave * 32 + l31;
    const bool cok = col < ncols;
    const int colc = cok ? col : ncols - 1;

    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
            acc[a][r] = (row < n) ? B[(size_t)row * ldb + colc] : 0.0f;
        }
    // R block + sub-block inverses into LDS: all 20 float4 loads of a thread in flight before the first is stored.  As a
    // load -> wait -> store loop (what `for (e = tid; ...) Rs[e] = Rd[e]` compiles to) the copy paid 20 memory round trips
    // one after the other -- more than half of this kernel's 17 us (ISA read in round 4; potf2's fill had the same shape).
    {
        constexpr int NV = RD_STRIDE / 4 / 256;
        static_assert(NV * 4 * 256 == RD_STRIDE, "whole float4s per thread");
        f32x4 v[NV];
#pragma unroll
        for (int r = 0; r < NV; ++r) v[r] = ((const f32x4*)Rd)[tid + 256 * r];
#pragma unroll
        for (int r = 0; r < NV; ++r) ((f32x4*)Rs)[tid + 256 * r] = v[r];
    }
    __syncthreads();

#pragma unroll
    for (int b = 0; b < 4; ++b) {
        // Z_b[i][col] = sum_k X_b[k][i] * B_b[k][col]
        f32x16 zb;
#pragma unroll
        for (int r = 0; r < 16; ++r) zb[r] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float av = D32[(b * 32 + k) * 32 + l31];  // X_b[k][i = l31]
            zb = __builtin_amdgcn_mfma_f32_32x32x2f32(av, acc[b][r], zb, 0, 0, 0);
        }
        acc[b] = zb;
#pragma unroll
        for (int a = b + 1; a < 4; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float av = -Rs[(32 * b + k) * NB + 32 * a + l31];  // -R[32b+k][32a+i]
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, zb[r], acc[a], 0, 0, 0);
            }
        }
    }
    if (cok) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < n) Z[(size_t)row * ldz + col] = acc[a][r];
            }
    }
}

// Batched inverse of the diagonal blocks (one workgroup per block, off the factorisation's
// critical path): X = R^-1, thread j < 128 owns column j (x = e_j) and runs the column-oriented
// back substitution p = 127..0: x_p /= R_pp; x_i -= R_ip x_p (i < p), reading column p of R as a
// contiguous row of the transposed LDS copy.
//   Dinv[b]  : R^-1 dense 128x128 row-major (zeros outside the triangle / n)
//   Ydiag(b) : (R^-1)^T dense n x n at Y + (b*128)*(ldy+1) (zeros above the diagonal)
__global__ __launch_bounds__(128) void trinv_batched_kernel(const float* __restrict__ Rd_all, int K,
                                                            float* __restrict__ Dinv_all,
                                                            float* __restrict__ Y, int64_t ldy, int64_t bsW, int64_t bsY) {
    if (blockIdx.y) {      // problem b of a batch
        Rd_all += (size_t)blockIdx.y * bsW;
        Dinv_all += (size_t)blockIdx.y * bsW;
        Y += (size_t)blockIdx.y * bsY;
    }
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Rt = sm;             // [NB][LDT]  Rt[p][i] = R[i][p]
    float* Xs = sm + NB * LDT;  // [NB][LDP]
    const int b = blockIdx.x;
    const int n = (K - b * NB < NB) ? K - b * NB : NB;
    const float* Rd = Rd_all + (size_t)b * RD_STRIDE;
    const int j = threadIdx.x;
    for (int e = j; e < NB * NB; e += NB) {
        const int i = e / NB, c = e % NB;
        Rt[c * LDT + i] = Rd[e];
    }
    __syncthreads();
    float x0[32], x1[32], x2[32], x3[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        x0[r] = (r == j) ? 1.0f : 0.0f;
        x1[r] = (32 + r == j) ? 1.0f : 0.0f;
        x2[r] = (64 + r == j) ? 1.0f : 0.0f;
        x3[r] = (96 + r == j) ? 1.0f : 0.0f;
    }
#pragma unroll 1
    for (int pb = 3; pb >= 0; --pb) {
#pragma unroll
        for (int pp = 31; pp >= 0; --pp) {
            const int p = 32 * pb + pp;
            const float* rp = Rt + p * LDT;  // rp[i] = R[i][p]
            const float xp = QT_SEL4(pb, x0[pp], x1[pp], x2[pp], x3[pp]) / rp[p];
            if (pb == 0) x0[pp] = xp;
            if (pb == 1) x1[pp] = xp;
            if (pb == 2) x2[pp] = xp;
            if (pb == 3) x3[pp] = xp;
#define QT_XUPD(arr, a, r_to)                                                 \
    _Pragma("unroll") for (int r4 = 0; r4 < (r_to); r4 += 4) {                 \
        const f32x4 v4 = *(const f32x4*)(rp + 32 * (a) + r4);                 \
        _Pragma("unroll") for (int e = 0; e < 4; ++e)                         \
            if (r4 + e < (r_to)) arr[r4 + e] = fmaf(-v4[e], xp, arr[r4 + e]); \
    }
            if (pb == 0) { QT_XUPD(x0, 0, pp) } else { QT_XUPD(x0, 0, 32) }
            if (pb == 1) { QT_XUPD(x1, 1, pp) } else if (pb > 1) { QT_XUPD(x1, 1, 32) }
            if (pb == 2) { QT_XUPD(x2, 2, pp) } else if (pb > 2) { QT_XUPD(x2, 2, 32) }
            if (pb == 3) { QT_XUPD(x3, 3, pp) }
#undef QT_XUPD
        }
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        Xs[r * LDP + j] = x0[r];
        Xs[(32 + r) * LDP + j] = x1[r];
        Xs[(64 + r) * LDP + j] = x2[r];
        Xs[(96 + r) * LDP + j] = x3[r];
    }
    __syncthreads();
    float* Dinv = Dinv_all + (size_t)b * NB * NB;
    float* Yd = Y + (size_t)(b * NB) * ldy + b * NB;
    for (int e = j; e < NB * NB; e += NB) {
        const int i = e / NB, jj = e % NB;
        const bool in = (i < n && jj < n);
        Dinv[e] = (in && jj >= i) ? Xs[i * LDP + jj] : 0.0f;
        if (in) Yd[(size_t)i * ldy + jj] = (jj <= i) ? Xs[jj * LDP + i] : 0.0f;
    }
}

// The same outputs on the f32 MFMA (round 4; QT_CHOL_TRINV=div selects the kernel above): Z = R^-T I by trsm_rt_kernel's
// blocked forward substitution with the 32x32 inverses potf2 already computed -- four stages of MFMAs instead of 128
// sequential divide-and-update steps (97 -> ~20 us per launch at K = 4096, 137 -> ~25 at K = 14336; one launch per chain).
// Z is lower triangular: Ydiag(b) = Z, Dinv[b] = Z^T.  Wave w owns the columns 32w .. 32w + 31 of the identity.
__global__ __launch_bounds__(256) void trinv_mfma_kernel(const float* __restrict__ Rd_all, int K,
                                                         float* __restrict__ Dinv_all,
                                                         float* __restrict__ Y, int64_t ldy, int64_t bsW, int64_t bsY) {
    if (blockIdx.y) {      // problem b of a batch
        Rd_all += (size_t)blockIdx.y * bsW;
        Dinv_all += (size_t)blockIdx.y * bsW;
        Y += (size_t)blockIdx.y * bsY;
    }
    extern __shared__ __attribute__((aligned(16))) float Rs[];  // [RD_STRIDE]
    const int blk = blockIdx.x;
    const int n = (K - blk * NB < NB) ? K - blk * NB : NB;
    const float* Rd = Rd_all + (size_t)blk * RD_STRIDE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // as in trsm_rt_kernel: every load of the copy in flight before the first store
        constexpr int NV = RD_STRIDE / 4 / 256;
        f32x4 v[NV];
#pragma unroll
        for (int r = 0; r < NV; ++r) v[r] = ((const f32x4*)Rd)[tid + 256 * r];
#pragma unroll
        for (int r = 0; r < NV; ++r) ((f32x4*)Rs)[tid + 256 * r] = v[r];
    }
    const float* D32 = Rs + NB * NB;
    const int l31 = lane & 31, h = lane >> 5;
    const int col = wave * 32 + l31;

    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
            acc[a][r] = (row == col) ? 1.0f : 0.0f;
        }
    __syncthreads();

#pragma unroll
    for (int b = 0; b < 4; ++b) {
        f32x16 zb;
#pragma unroll
        for (int r = 0; r < 16; ++r) zb[r] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float av = D32[(b * 32 + k) * 32 + l31];  // X_b[k][i = l31]
            zb = __builtin_amdgcn_mfma_f32_32x32x2f32(av, acc[b][r], zb, 0, 0, 0);
        }
        acc[b] = zb;
#pragma unroll
        for (int a = b + 1; a < 4; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float av = -Rs[(32 * b + k) * NB + 32 * a + l31];  // -R[32b+k][32a+i]
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, zb[r], acc[a], 0, 0, 0);
            }
        }
    }
    float* Dinv = Dinv_all + (size_t)blk * NB * NB;
    float* Yd = Y + (size_t)(blk * NB) * ldy + blk * NB;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
            const bool in = row < n && col < n;
            const float z = (in && row >= col) ? acc[a][r] : 0.0f;      // Z[row][col] = (R^-1)[col][row]
            Dinv[col * NB + row] = z;
            if (in) Yd[(size_t)row * ldy + col] = z;
        }
}

// XT[k][m] = Y_II[m][k] for a w x w diagonal block of the lower-triangular Y (entries above the diagonal
// of Y are not initialised outside the 128-blocks on the diagonal: written as zeros here)
__global__ __launch_bounds__(256) void transpose_lower_block_kernel(const float* __restrict__ Y, int64_t ldy, int w,
                                                                    float* __restrict__ XT, int ldx, int64_t bsY,
                                                                    int64_t bsXT) {
    if (blockIdx.z) {
        Y += (size_t)blockIdx.z * bsY;
        XT += (size_t)blockIdx.z * bsXT;
    }
    __shared__ float tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;   // XT rows k = by.., cols m = bx..
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int m = bx + r, k = by + tx;                   // read Y_II[m][k], coalesced along k
        tile[r][tx] = (m < w && k < w && k <= m) ? Y[(size_t)m * ldy + k] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = by + r, m = bx + tx;
        if (k < w && m < w) XT[(size_t)k * ldx + m] = tile[tx][r];
    }
}

// In-place flat reversal of the lower-triangular Y into the upper-triangular U:
// U[i][j] = Y[K-1-i][K-1-j] for j >= i, strict lower triangle of U = 0.
__global__ __launch_bounds__(256) void flat_reverse_lower_to_upper_kernel(float* __restrict__ U, int K, int64_t bsU) {
    U += (size_t)blockIdx.z * bsU;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= K || j < i) return;
    const size_t e = (size_t)i * K + j;
    const size_t pe = (size_t)(K - 1 - i) * K + (K - 1 - j);
    if (i == j) {
        if (e <= pe) {
            const float t = U[e];
            U[e] = U[pe];
            U[pe] = t;
        }
    } else {
        U[e] = U[pe];
        U[pe] = 0.0f;
    }
}

// Upstream's LinAlgError fallback, applied on the device so the host need not synchronise:
// if the factorisation hit a non-positive pivot, U = I (plain round-to-nearest).
__global__ __launch_bounds__(256) void identity_if_failed_kernel(float* __restrict__ U, int K,
                                                                 const int32_t* __restrict__ info, int64_t bsU) {
    // a fixed, small grid striding over the rows: in the normal case (info == 0) every workgroup leaves at once, and a
    // launch of K * K / 256 workgroups that all do so took 141 us at K = 14336 (16 us at K = 4096) for nothing
    if (info[blockIdx.z] == 0) return;
    U += (size_t)blockIdx.z * bsU;
    for (int i = blockIdx.x; i < K; i += gridDim.x)
        for (int j = threadIdx.x; j < K; j += blockDim.x) U[(size_t)i * K + j] = (i == j) ? 1.0f : 0.0f;
}

}  // namespace

// bytes of ONE problem's share of the workspace (a multiple of 256), without the bf16x3 item tables (shared by a batch)
static size_t chol_problem_ws_bytes(int K, const CholG3Plan* g3) {
    const size_t nb = (K + NB - 1) / NB;
    // T panel [128, K] + T_I panel [512, K] + X_II [512, 512] + Rd, Dinv [nb][128*128] each + split-K slabs
    // (a split product writes splits * M * N floats with splits <= 2048 workgroups / tiles: <= 2048 * 128 * 128)
    const size_t split = (size_t)2048 * NB * NB * 4 + (size_t)NBO_MAX * K * 4;
    size_t n = (size_t)NB * K * 4 + (size_t)NBO_MAX * K * 4 + (size_t)NBO_MAX * NBO_MAX * 4 + nb * (NB * NB + RD_STRIDE) * 4 + split + 256;
    if (g3->any) n += 2 * qt_align_up((size_t)3 * K * K * 2, 256) + (size_t)g3->max_slabs * 256 * 256 * 4 + 256;
    return qt_align_up(n, 256);
}

extern "C" size_t qt_cholesky_inverse_upper_batched_workspace_bytes(int K, int n_problems) {
    if (K <= 0 || n_problems <= 0) return 0;
    int nbo_, nbi_;
    const CholG3Plan* g3 = chol_blocks(K, nbo_, nbi_);
    return (size_t)n_problems * chol_problem_ws_bytes(K, g3) + g3->blob_bytes + 512;
}

extern "C" size_t qt_cholesky_inverse_upper_workspace_bytes(int K) {
    return qt_cholesky_inverse_upper_batched_workspace_bytes(K, 1);
}

// n problems of one size K in every launch of the chain: problem b's matrices at A + b * strideA, U + b * strideU
// (elements), its pivot flag at info[b].  Per problem the kernels, their launch shapes and the order of every sum are
// those of a single-problem call (tile sizes and split-K decisions come from one problem's shape), so the factors are
// bit-identical to n separate calls; what changes is that the latency-bound panel kernels (potf2: ONE workgroup per
// problem; trsm, the short folds) and the block-row products serve all n problems per launch.
static int chol_run(float* A, int64_t strideA, int K, float* U, int64_t strideU, int32_t* info, int nprob, void* workspace,
                    size_t workspace_bytes, hipStream_t stream) {
    const size_t need = qt_cholesky_inverse_upper_batched_workspace_bytes(K, nprob);
    if (!workspace || workspace_bytes < need) {
        qt_set_error("qt_cholesky_inverse_upper: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    const int nblk = (K + NB - 1) / NB;
    char* ws = (char*)qt_align_up((size_t)workspace, 256);
    int NBO, NBI;
    const CholG3Plan* g3 = chol_blocks(K, NBO, NBI);
    const size_t prob_bytes = chol_problem_ws_bytes(K, g3);
    const int64_t sW = (int64_t)(prob_bytes / 4);            // one problem's workspace, in floats
    const int64_t sA = nprob > 1 ? strideA : 0, sU = nprob > 1 ? strideU : 0;
    float* T = (float*)ws;
    float* TI = T + (size_t)NB * K;
    float* XT = TI + (size_t)NBO_MAX * K;
    float* Rd = XT + (size_t)NBO_MAX * NBO_MAX;
    float* Dinv = Rd + (size_t)nblk * RD_STRIDE;
    float* split_ws = Dinv + (size_t)nblk * NB * NB;
    const size_t split_ws_bytes = (size_t)2048 * NB * NB * 4 + (size_t)NBO_MAX * K * 4;
    // bf16x3 products: plane copies of R and Y, slabs (per problem), item tables (one copy for the batch)
    unsigned short *Rpl = nullptr, *Ypl = nullptr;
    float* g3_slabs = nullptr;
    char* g3_tab = nullptr;
    const int64_t plane_stride = (int64_t)K * K;
    if (g3->any) {
        char* q = (char*)qt_align_up((size_t)((char*)split_ws + split_ws_bytes), 256);
        Rpl = (unsigned short*)q;
        q += qt_align_up((size_t)3 * K * K * 2, 256);
        Ypl = (unsigned short*)q;
        q += qt_align_up((size_t)3 * K * K * 2, 256);
        g3_slabs = (float*)q;
        g3_tab = (char*)qt_align_up((size_t)(ws + (size_t)nprob * prob_bytes), 256);
        QT_HIP(hipMemcpyAsync(g3_tab, g3->blob, g3->blob_bytes, hipMemcpyHostToDevice, stream));
    }
    // where a block-row product's C lives decides its batch stride: A (factor phase) or the workspace (T_I)
    auto g3_row = [&](const G3Step& st, const unsigned short* Bplanes, int col_a0, int col_b0, int M, int N, float* C,
                      int64_t bsC, int mode) -> int {
        G3Args a;
        a.Apl = Rpl;
        a.Bpl = Bplanes;
        a.plane_stride = plane_stride;
        a.ld = K;
        a.rowA0 = a.rowB0 = 0;
        a.colA0 = col_a0;
        a.colB0 = col_b0;
        a.colmax = K;
        a.M = M;
        a.N = N;
        a.C = C;
        a.ldc = K;
        a.mode = mode;
        a.slabs = g3_slabs;
        a.items = (const G3Item*)(g3_tab + st.item_off);
        a.n_items = st.n_items;
        a.red = (const G3Red*)(g3_tab + st.red_off);
        a.n_red = st.n_red;
        a.batch = nprob;
        a.bsApl = a.bsBpl = sW * 2;       // plane copies live in the problem's workspace (bf16 elements)
        a.bsC = bsC;
        a.bsSlabs = sW;
        return qt_gemm3_launch(a, stream);
    };
    // an f32 product of the chain, for all problems: strides by where each operand lives
    auto sgemm = [&](SgemmArgs& g, int64_t bsA_, int64_t bsB_, int64_t bsC_) -> int {
        g.batch = nprob;
        g.bsA = bsA_;
        g.bsB = bsB_;
        g.bsCin = g.bsCout = bsC_;
        g.bs_split = sW;
        return qt_sgemm_tn(g, stream);
    };
    float* Y = U;
    const int64_t sY = sU;
    const size_t inv_lds = (size_t)(NB * LDT + NB * LDP) * sizeof(float);
    const size_t potf2_lds = (size_t)(NB * LDA + 4 * 32 * 32) * sizeof(float);
    static QtOncePerDevice lds_attr;
    QT_HIP(lds_attr.run([&] {
        hipError_t e = hipFuncSetAttribute((const void*)potf2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)potf2_lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)trsm_rt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(RD_STRIDE * sizeof(float)));
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)trinv_batched_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)inv_lds);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)trinv_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)(RD_STRIDE * sizeof(float)));
        return e;
    }));
    QT_HIP(hipMemsetAsync(info, 0, sizeof(int32_t) * nprob, stream));
    const int prio = qt_chain_prio();

    // ---- A = R^T R over NBO-wide outer blocks, left-looking inside one; in place.  Outer level: right-looking
    // (one k = NBO update of the trailing matrix per block) on the f32 MFMA, or -- when the bf16x3 products
    // are in use for this K -- left-looking (block row J gathers all earlier block rows in one long-k product)
    for (int J0 = 0, Jb = 0; J0 < K; J0 += NBO, ++Jb) {
        const int J1 = (K - J0 < NBO) ? K : J0 + NBO;
        if (g3->any && J0 > 0) {
            // A[J, J0:] -= R[:J0, J]^T R[:J0, J0:]
            const G3Step& st = g3->fac[Jb];
            float* C = A + (size_t)J0 * K + J0;
            if (st.n_items > 0) {
                int rc = g3_row(st, Rpl, J0, J0, J1 - J0, K - J0, C, sA, G3_SUB);
                if (rc) return rc;
            } else {
                SgemmArgs g;
                g.A = A + J0; g.lda = K;
                g.B = A + J0; g.ldb = K;
                g.Cin = C; g.ldcin = K;
                g.Cout = C; g.ldcout = K;
                g.M = J1 - J0; g.N = K - J0; g.kdim = J0; g.k_mode = SG_K_FULL; g.mode = SG_MODE_SUB;
                g.split_ws = split_ws; g.split_ws_bytes = split_ws_bytes;
                int rc = sgemm(g, sA, sA, sA);
                if (rc) return rc;
            }
        }
        for (int j0 = J0; j0 < J1; j0 += NB) {
            const int j = j0 / NB, nbj = (K - j0 < NB) ? K - j0 : NB;
            float* Ajj = A + (size_t)j0 * K + j0;
            if (j0 > J0) {
                // rows J0..j0 of this outer block are final: fold them into block row j
                SgemmArgs g;
                g.A = A + (size_t)J0 * K + j0; g.lda = K;
                g.B = A + (size_t)J0 * K + j0; g.ldb = K;
                g.Cin = Ajj; g.ldcin = K;
                g.Cout = Ajj; g.ldcout = K;
                g.M = nbj; g.N = K - j0; g.kdim = j0 - J0; g.k_mode = SG_K_FULL; g.mode = SG_MODE_SUB;
                int rc = sgemm(g, sA, sA, sA);
                if (rc) return rc;
            }
            hipLaunchKernelGGL(potf2_kernel, dim3(nprob), dim3(POTF2_THREADS), potf2_lds, stream, (const float*)Ajj, (int64_t)K, nbj,
                               Ajj, (int64_t)K, Rd + (size_t)j * RD_STRIDE, info, j0, prio, sA, sW);
            QT_LAUNCH_CHECK();
            const int rest = K - j0 - nbj;
            if (rest > 0) {
                hipLaunchKernelGGL(trsm_rt_kernel, dim3((rest + 127) / 128, nprob), dim3(256), RD_STRIDE * sizeof(float),
                                   stream, (const float*)(Rd + (size_t)j * RD_STRIDE), nbj, (const float*)(Ajj + nbj),
                                   (int64_t)K, Ajj + nbj, (int64_t)K, rest, prio, sW, sA);
                QT_LAUNCH_CHECK();
            }
        }
        const int rem = K - J1;
        if (g3->any) {
            // plane copies of the finished block row (columns from its diagonal block on; only entries above
            // the diagonal are ever read back)
            if (rem > 0) {
                int rc = qt_split3_launch(A + (size_t)J0 * K + J0, K, J1 - J0, K - J0, Rpl + (size_t)J0 * K + J0, K,
                                          plane_stride, 0, 0, 0, stream, nprob, sA, sW * 2);
                if (rc) return rc;
            }
        } else if (rem > 0) {
            // the K^3/3 of the factorisation: one SYRK-shaped update per outer block, upper tiles only
            SgemmArgs g;
            g.A = A + (size_t)J0 * K + J1; g.lda = K;
            g.B = A + (size_t)J0 * K + J1; g.ldb = K;
            g.Cin = A + (size_t)J1 * K + J1; g.ldcin = K;
            g.Cout = A + (size_t)J1 * K + J1; g.ldcout = K;
            g.M = rem; g.N = rem; g.kdim = J1 - J0; g.k_mode = SG_K_FULL; g.mode = SG_MODE_SUB;
            g.upper_only = 1;
            int rc = sgemm(g, sA, sA, sA);
            if (rc) return rc;
        }
    }
    // ---- all diagonal-block inverses at once ----
    {
        const char* e = getenv("QT_CHOL_TRINV");   // read per call: the tests compare the two kernels in one process
        if (e && strcmp(e, "div") == 0)
            hipLaunchKernelGGL(trinv_batched_kernel, dim3(nblk, nprob), dim3(NB), inv_lds, stream, (const float*)Rd, K, Dinv, Y,
                               (int64_t)K, sW, sY);
        else
            hipLaunchKernelGGL(trinv_mfma_kernel, dim3(nblk, nprob), dim3(256), RD_STRIDE * sizeof(float), stream,
                               (const float*)Rd, K, Dinv, Y, (int64_t)K, sW, sY);
    }
    QT_LAUNCH_CHECK();
    // ---- Y = R^-T by NBI-row block rows ----
    for (int I0 = 0, Ib = 0; I0 < K; I0 += NBI, ++Ib) {
        const int I1 = (K - I0 < NBI) ? K : I0 + NBI;
        const int Wi = I1 - I0;
        // the diagonal 512-block Y_II by the 128-row recurrence (short k)
        for (int i0 = I0 + NB; i0 < I1; i0 += NB) {
            const int i = i0 / NB, nbi = (K - i0 < NB) ? K - i0 : NB;
            SgemmArgs g;
            g.A = A + (size_t)I0 * K + i0; g.lda = K;
            g.B = Y + (size_t)I0 * K + I0; g.ldb = K;
            g.Cin = nullptr; g.ldcin = 0;
            g.Cout = T; g.ldcout = K;
            g.M = nbi; g.N = i0 - I0; g.kdim = i0 - I0; g.k_mode = SG_K_FROM_N0; g.mode = SG_MODE_SET;
            int rc = sgemm(g, sA, sY, sW);
            if (rc) return rc;
            SgemmArgs t;
            t.A = Dinv + (size_t)i * NB * NB; t.lda = NB;
            t.B = T; t.ldb = K;
            t.Cin = nullptr; t.ldcin = 0;
            t.Cout = Y + (size_t)i0 * K + I0; t.ldcout = K;
            t.M = nbi; t.N = i0 - I0; t.kdim = nbi; t.k_mode = SG_K_FULL; t.mode = SG_MODE_NEG;
            rc = sgemm(t, sW, sW, sY);
            if (rc) return rc;
        }
        // plane copy of the finished block row of Y (zeros above the diagonal), for the later rows' products
        auto split_row = [&]() -> int {
            if (!g3->any || I1 >= K) return QT_OK;
            return qt_split3_launch(Y + (size_t)I0 * K, K, Wi, I1, Ypl + (size_t)I0 * K, K, plane_stride, 1, I0, 0, stream,
                                    nprob, sY, sW * 2);
        };
        if (I0 == 0) {
            int rc0 = split_row();
            if (rc0) return rc0;
            continue;
        }
        // left of the diagonal block: the K^3/3 of the inverse, one product per block row
        int rc;
        if (g3->any && g3->inv[Ib].n_items > 0) {
            rc = g3_row(g3->inv[Ib], Ypl, I0, 0, Wi, I0, TI, sW, G3_SET);
        } else {
            SgemmArgs g;
            g.A = A + I0; g.lda = K;
            g.B = Y; g.ldb = K;
            g.Cin = nullptr; g.ldcin = 0;
            g.Cout = TI; g.ldcout = K;
            g.M = Wi; g.N = I0; g.kdim = I0; g.k_mode = SG_K_FROM_N0; g.mode = SG_MODE_SET;
            g.split_ws = split_ws; g.split_ws_bytes = split_ws_bytes;
            rc = sgemm(g, sA, sY, sW);
        }
        if (rc) return rc;
        hipLaunchKernelGGL(transpose_lower_block_kernel, dim3((Wi + 31) / 32, (Wi + 31) / 32, nprob), dim3(256), 0, stream,
                           (const float*)(Y + (size_t)I0 * K + I0), (int64_t)K, Wi, XT, NBO_MAX, sY, sW);
        QT_LAUNCH_CHECK();
        SgemmArgs t;
        t.A = XT; t.lda = NBO_MAX;
        t.B = TI; t.ldb = K;
        t.Cin = nullptr; t.ldcin = 0;
        t.Cout = Y + (size_t)I0 * K; t.ldcout = K;
        t.M = Wi; t.N = I0; t.kdim = Wi; t.k_mode = SG_K_FULL; t.mode = SG_MODE_NEG;
        rc = sgemm(t, sW, sW, sY);
        if (rc) return rc;
        rc = split_row();
        if (rc) return rc;
    }
    hipLaunchKernelGGL(flat_reverse_lower_to_upper_kernel, dim3((K + 255) / 256, K, nprob), dim3(256), 0, stream, U, K, sU);
    QT_LAUNCH_CHECK();
    hipLaunchKernelGGL(identity_if_failed_kernel, dim3(K < 1024 ? K : 1024, 1, nprob), dim3(256), 0, stream, U, K,
                       (const int32_t*)info, sU);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

extern "C" int qt_cholesky_inverse_upper(float* A, int K, float* U, int32_t* info, void* workspace,
                                         size_t workspace_bytes, qt_stream_t stream_) {
    QT_CHECK_ARG(K > 0 && A && U && info, "qt_cholesky_inverse_upper: bad arguments");
    return chol_run(A, 0, K, U, 0, info, 1, workspace, workspace_bytes, (hipStream_t)stream_);
}

extern "C" int qt_cholesky_inverse_upper_batched(float* A, int64_t strideA, int K, float* U, int64_t strideU, int32_t* info,
                                                 int n_problems, void* workspace, size_t workspace_bytes,
                                                 qt_stream_t stream_) {
    QT_CHECK_ARG(K > 0 && A && U && info && n_problems >= 1 && n_problems <= SG_MAX_BATCH,
                 "qt_cholesky_inverse_upper_batched: bad arguments (1 <= n_problems <= %d)", SG_MAX_BATCH);
    QT_CHECK_ARG(n_problems == 1 || (strideA >= (int64_t)K * K && strideU >= (int64_t)K * K && strideA % 4 == 0 && strideU % 4 == 0),
                 "qt_cholesky_inverse_upper_batched: strides must be multiples of 4 elements and >= K*K");
    return chol_run(A, strideA, K, U, strideU, info, n_problems, workspace, workspace_bytes, (hipStream_t)stream_);
}
