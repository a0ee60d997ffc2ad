// a11: the GPTQ column sweep (upstream quantize_weight's block loop,
// reached through gptq.py:86 / base.py:161).
//
// Rows of W are independent given U, so one lane owns one row.  Per 128-column block:
//   sweep_block_kernel : the 128 sequential quantise / error-feedback steps.  The U block and
//                        the lane's 128 weights live in LDS; the active 32-column sub-block
//                        lives in registers (statically indexed), the other columns receive
//                        rank-32 updates with float4 broadcasts of U.  Per element the update
//                        sequence is exactly upstream's: ascending source column, each step
//                        "w = w - (err * u)" with two roundings (no contraction).
//   sgemm_tn (SUB)     : W[:, i2:] -= Err1 @ U[i1:i2, i2:] as an ascending-k fmaf chain from 0
//                        on the f32 MFMA, then one subtraction -- the oracle's fixed order.
// Bit-exact against oracle/gptq_oracle.c:orc_gptq_sweep.
#include <stdlib.h>

#include <map>
#include <tuple>
#include <vector>

#include "common.h"
#include "gemm3_tn.h"
#include "sgemm_tn.h"

#pragma clang fp contract(off)

namespace {

constexpr int BS = 128;    // upstream block_size default
constexpr int SWEEP_MAX_BATCH = 8;  // blocks whose far update may be merged into one pass over W (workspace is sized for it)
constexpr int SB = 32;     // register sub-block
constexpr int ROWS = 128;  // rows (lanes) per workgroup: 2 waves
constexpr size_t SWEEP_LDS = (size_t)(BS * BS + BS * ROWS + 2 * BS) * sizeof(float);

// Row groups of a stacked sweep (qt_gptq_sweep_grouped): rows [row_end[g-1], row_end[g]) belong to problem g, whose factor
// is U + g * bsU and whose column groups are g_idx + g * K.  n == 0: one problem.
struct SweepGroups {
    int n;
    int64_t bsU;
    int row_end[SG_MAX_GROUPS];
};
__device__ __forceinline__ int sweep_group_of(const SweepGroups& sg, int row0) {
    int g = 0;
    while (g + 1 < sg.n && row0 >= sg.row_end[g]) ++g;
    return g;
}

__global__ __launch_bounds__(ROWS) void sweep_block_kernel(float* __restrict__ W, int R, int K,
                                                           const float* __restrict__ U,
                                                           const float* __restrict__ scale_t,
                                                           const float* __restrict__ zp_t,
                                                           const int32_t* __restrict__ g_idx, int i1, int cnt,
                                                           float qmin, float qmax, int8_t* __restrict__ Qt,
                                                           float* __restrict__ ErrT, float* __restrict__ loss,
                                                           int prio, SweepGroups sg) {
    qt_set_chain_prio(prio);
    if (sg.n > 1) {
        const int g = sweep_group_of(sg, blockIdx.x * ROWS);
        U += (size_t)g * sg.bsU;
        g_idx += (size_t)g * K;
    }
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Un = sm;                  // [BS][BS]   Un[i][j] = U[i1+i][i1+j], zero outside j>=i / cnt
    float* wl = sm + BS * BS;        // [BS cols][ROWS]
    float* dd = wl + BS * ROWS;      // [BS] diag, [BS] diag^2
    const int tid = threadIdx.x;
    const int row = blockIdx.x * ROWS + tid;
    const bool valid = row < R;
    const int rowc = valid ? row : R - 1;

    // Prologue loads are issued in batches of 16 independent float4 loads per thread and only then
    // written to LDS: a load -> wait -> store loop costs one memory round trip per iteration (32 + 128
    // of them used to be half of this kernel's time).  Out-of-block elements of a ragged last block are
    // read from a clamped in-block address and replaced by the inert value afterwards (cnt % 4 == 0
    // because K % 4 == 0, so a float4 is either wholly inside or wholly outside).
    {
        const float* ublk = U + (size_t)i1 * K + i1;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e4 = tid + ROWS * (half * 16 + r);      // float4 index in the 128 x 32 grid
                const int i = e4 >> 5, j = (e4 & 31) * 4;
                const int ic = i < cnt ? i : cnt - 1, jc = j < cnt ? j : cnt - 4;
                v[r] = *(const f32x4*)(ublk + (size_t)ic * K + jc);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e4 = tid + ROWS * (half * 16 + r);
                const int i = e4 >> 5, j = (e4 & 31) * 4;
                const bool inside = i < cnt && j < cnt;
#pragma unroll
                for (int q = 0; q < 4; ++q) Un[i * BS + j + q] = (inside && j + q >= i) ? v[r][q] : 0.0f;
            }
        }
    }
    {
        const float d = (tid < cnt) ? U[(size_t)(i1 + tid) * K + (i1 + tid)] : 1.0f;
        dd[tid] = d;
        dd[BS + tid] = 1.0f / (d * d);   // only the loss uses it (tolerance 1e-6, not part of the bit-exact contract)
    }
    {
        const float* wrow = W + (size_t)rowc * K + i1;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x4 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (half * 16 + r) * 4;
                v[r] = *(const f32x4*)(wrow + (c < cnt ? c : cnt - 4));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (half * 16 + r) * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) wl[(c + q) * ROWS + tid] = (c < cnt) ? v[r][q] : 0.0f;
            }
        }
    }
    __syncthreads();

    float blk_loss = 0.0f;
    for (int sb = 0; sb < BS / SB; ++sb) {
        const int cb = sb * SB;
        if (cb >= cnt) break;
        float w[SB], er[SB], sc[SB], zz[SB];
#pragma unroll
        for (int t = 0; t < SB; ++t) w[t] = wl[(cb + t) * ROWS + tid];
#pragma unroll
        for (int t = 0; t < SB; ++t) {
            const int c = cb + t;
            const int g = (c < cnt) ? g_idx[i1 + c] : g_idx[i1];
            sc[t] = scale_t[(size_t)g * R + rowc];
            zz[t] = zp_t[(size_t)g * R + rowc];
        }
#pragma unroll
        for (int t = 0; t < SB; ++t) {
            const int c = cb + t;
            {  // columns >= cnt of a ragged last block run as inert padding (w=0, U=0, d=1)
                const float d = dd[c], rd2 = dd[BS + c];
                const float wv = w[t];
                float x = wv / sc[t];
                x = x + zz[t];
                x = fminf(fmaxf(x, qmin), qmax);
                const float q = rintf(x);
                const float dq = (q - zz[t]) * sc[t];
                const float diff = wv - dq;
                blk_loss = blk_loss + (diff * diff) * rd2;
                const float e = diff / d;
                er[t] = e;
                w[t] = dq;
                if (valid && c < cnt) {
                    Qt[(size_t)(i1 + c) * R + row] = (int8_t)q;
                    ErrT[(size_t)c * R + row] = e;
                }
                const float* urow = Un + c * BS + cb;
                // two columns per instruction (v_pk_mul_f32 / v_pk_add_f32: the same IEEE single
                // operations, so the roundings are unchanged); an odd first column goes alone
                if ((t + 1) & 1) {
                    const float pr = e * urow[t + 1];
                    w[t + 1] = w[t + 1] - pr;
                }
                const f32x2 e2 = {e, e};
#pragma unroll
                for (int u = (t + 2) & ~1; u < SB; u += 2) {
                    const f32x2 uu = *(const f32x2*)(urow + u);
                    const f32x2 pr = e2 * uu;
                    f32x2 wp = {w[u], w[u + 1]};
                    wp = wp - pr;
                    w[u] = wp[0];
                    w[u + 1] = wp[1];
                }
            }
            // keep each column step's LDS broadcasts next to their use (without this hipcc
            // hoists all 496 U reads of the triangle and spills)
            __builtin_amdgcn_sched_barrier(0);
        }
        // dequantised values of this sub-block back to W (upstream: W[:, i1:i2] = Q1)
        if (valid) {
            float* wrow = W + (size_t)row * K + i1 + cb;
#pragma unroll
            for (int t = 0; t < SB; t += 4) {
                if (cb + t + 3 < cnt) {
                    f32x4 v = {w[t], w[t + 1], w[t + 2], w[t + 3]};
                    *(f32x4*)(wrow + t) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (cb + t + e < cnt) wrow[t + e] = w[t + e];
                }
            }
        }
        // rank-32 update of the block's remaining columns (ascending source column per element)
        for (int j4 = cb + SB; j4 < cnt; j4 += 4) {
            f32x2 wa = {wl[(j4 + 0) * ROWS + tid], wl[(j4 + 1) * ROWS + tid]};
            f32x2 wb = {wl[(j4 + 2) * ROWS + tid], wl[(j4 + 3) * ROWS + tid]};
#pragma unroll
            for (int t = 0; t < SB; ++t) {
                const f32x4 u = *(const f32x4*)(Un + (cb + t) * BS + j4);
                const f32x2 e2 = {er[t], er[t]};
                const f32x2 ua = {u[0], u[1]}, ub = {u[2], u[3]};
                const f32x2 pa = e2 * ua, pb = e2 * ub;
                wa = wa - pa;
                wb = wb - pb;
            }
            wl[(j4 + 0) * ROWS + tid] = wa[0];
            wl[(j4 + 1) * ROWS + tid] = wa[1];
            wl[(j4 + 2) * ROWS + tid] = wb[0];
            wl[(j4 + 3) * ROWS + tid] = wb[1];
        }
    }
    if (valid) loss[row] = loss[row] + blk_loss / 2.0f;
}

// ---------------------------------------------------------------------------------------------------
// sweep_quad_kernel: the same 128 sequential steps with FOUR LANES PER ROW.
//
// The row-per-lane kernel above is bound by what ONE wave can issue (a lone wave issues a VALU op every 4
// cycles; per column: ~30 dependent ops of the quantise step + 2 ops per later column of the block), and a
// launch has only R / 64 waves (64 waves for R = 4096 on a chip with 1024 SIMDs).  Here lane (r, p),
// p = lane & 3, owns the block's columns 4m + p (m = 0..31) of row r, all 32 in registers: no weight tile
// in LDS, no sub-blocks.  At step c = 4 m0 + p0 every lane of the quad runs the quantise step on its own
// column-m0 register (only the owner's, p == p0, is final and kept), the owner's error is broadcast inside
// the quad by one DPP move, and every lane updates its own later columns -- a quarter of the row's update
// work per lane, four times the waves.  Per ELEMENT the operation sequence is unchanged: ascending source
// column, each step  w = w - (err * u)  with two roundings, IEEE division, round-half-even -- so the
// outputs are bit for bit those of the row-per-lane kernel and of oracle/gptq_oracle.c:orc_gptq_sweep.
// The per-row loss is summed in column order too (every lane of a quad carries the same running sum).
//
// LDS: only the U block, laid out per (step c, lane class p): Up[c][p][m] = U[i1+c][i1+4m+p] for
// 4m+p > c, else 0 (a lane reads its 32 - m0 values of a step as float4s; 36-float rows keep the four
// classes of a quad on disjoint banks).  73 KiB per workgroup of 64 rows, so two workgroups share a CU
// with room left for a GEMM workgroup.
constexpr int QL = 4;                 // lanes per row
constexpr int QROWS = 64;             // rows per workgroup (128: half as many workgroups holding a CU's LDS, but the
                                      // stage is 9 % slower alone and the bench does not move: 76.7-77.0 vs 76.5-76.9 ms)
constexpr int QTHREADS = QROWS * QL;  // 256
constexpr int QM = BS / QL;           // columns per lane: 32
constexpr int UP_P = QM + 4;          // floats per (c, p) row
constexpr int UP_C = QL * UP_P;       // floats per step c
constexpr size_t QUAD_LDS = (size_t)(BS * UP_C + 2 * BS) * sizeof(float);

template <int P0>
__device__ __forceinline__ float quad_bcast(float v) {
    // every lane of a quad gets lane P0's value (DPP quad_perm [P0, P0, P0, P0])
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), P0 * 0x55, 0xF, 0xF, true));
}

constexpr int QPF = 3;   // macro-steps (of 4 columns) a lane's scale / zero-point loads run ahead of their use

struct QuadState {
    float w[QM], sc[QM], zz[QM];   // sc / zz: only entries M0 .. M0 + QPF are live at macro-step M0
    float u[2][QM], d[2];          // this lane's U values and the diagonal entry of step C in u[C & 1] / d[C & 1]: read one step ahead
    float lsum, qv, ev, dv;
};

// this lane's U values of step C (columns 4m + p, m >= C / 4) and U[C][C], from LDS into the registers of parity C & 1
template <int C>
__device__ __forceinline__ void quad_fetch(QuadState& s, const float* __restrict__ Up, const float* __restrict__ dd, int p) {
    if constexpr (C < BS) {
        constexpr int M0 = C >> 2;
        const f32x4* urow = (const f32x4*)(Up + C * UP_C + p * UP_P);
#pragma unroll
        for (int k4 = M0 >> 2; k4 < QM / 4; ++k4) {
            const f32x4 t = urow[k4];
            s.u[C & 1][4 * k4 + 0] = t[0]; s.u[C & 1][4 * k4 + 1] = t[1]; s.u[C & 1][4 * k4 + 2] = t[2]; s.u[C & 1][4 * k4 + 3] = t[3];
        }
        s.d[C & 1] = dd[C];
    }
}

template <int M0, int P0>
__device__ __forceinline__ void quad_step(QuadState& s, const float* __restrict__ Up, const float* __restrict__ dd,
                                          int p, float qmin, float qmax) {
    constexpr int C = 4 * M0 + P0;
    // The LDS reads of step C + 1 are issued before step C's arithmetic (a lone wave per SIMD has nothing else to cover
    // their latency with; rows beyond a ragged block's last column are zero-filled, so the read ahead is always valid)
#if !defined(QT_SWEEP_LAB) || QT_SWEEP_LAB != 4     // lab build 4: the reads of a step at its own start, as up to round 4
    quad_fetch<C + 1>(s, Up, dd, p);
#else
    quad_fetch<C>(s, Up, dd, p);
#endif
    const float* u = s.u[C & 1];
    const float d = s.d[C & 1];
    // the quantise step on this lane's own column-M0 value (final only in the owner lane)
    const float wv = s.w[M0];
    float x = wv / s.sc[M0];
    x = x + s.zz[M0];
    x = fminf(fmaxf(x, qmin), qmax);
    const float q = rintf(x);
    const float dq = (q - s.zz[M0]) * s.sc[M0];
    const float diff = wv - dq;
    const float e = diff / d;
    const bool own = p == P0;
    s.dv = own ? diff : s.dv;
    s.qv = own ? q : s.qv;
    s.ev = own ? e : s.ev;
    const float eb = quad_bcast<P0>(e);
    // column M0 of the lanes behind the owner; the owner's becomes the dequantised value; lanes before it keep theirs
    {
        const float pr = eb * u[M0];
        const float t = s.w[M0] - pr;
        s.w[M0] = p > P0 ? t : (own ? dq : s.w[M0]);
    }
    // every later column of this lane (two per instruction where a pair is available)
    constexpr int MS = M0 + 1;
    if (MS < QM && (MS & 1)) {
        const float pr = eb * u[MS];
        s.w[MS] = s.w[MS] - pr;
    }
    const f32x2 e2 = {eb, eb};
#pragma unroll
    for (int m = (MS + 1) & ~1; m < QM; m += 2) {
        const f32x2 uu = {u[m], u[m + 1]};
        const f32x2 pr = e2 * uu;
        f32x2 wp = {s.w[m], s.w[m + 1]};
        wp = wp - pr;
        s.w[m] = wp[0];
        s.w[m + 1] = wp[1];
    }
    __builtin_amdgcn_sched_barrier(0);   // keep a step's LDS reads next to their use (register pressure)
}

struct QuadScales {
    const float* scale_t;   // + rowc
    const float* zp_t;      // + rowc
    const int32_t* gcol;    // g_idx + i1 + p
    int R, nm;
};

template <int M>
__device__ __forceinline__ void quad_load_scales(QuadState& s, const QuadScales& q) {
#if defined(QT_SWEEP_LAB) && QT_SWEEP_LAB == 3     // lab build 3 (timing only): no scale / zero-point loads inside the loop
    if constexpr (M >= QPF && M < QM) {
        s.sc[M] = s.sc[M - QPF];
        s.zz[M] = s.zz[M - QPF];
        return;
    }
#endif
    if constexpr (M < QM) {
        const int g = q.gcol[M < q.nm ? 4 * M : 0];
        s.sc[M] = q.scale_t[(size_t)g * q.R];
        s.zz[M] = q.zp_t[(size_t)g * q.R];
    }
}

template <int M0>
__device__ __forceinline__ void quad_macro_step(QuadState& s, const QuadScales& qs, const float* __restrict__ Up,
                                                const float* __restrict__ dd,
                                                int p, float qmin, float qmax, bool valid, int row, int R, int i1,
                                                int8_t* __restrict__ Qt, float* __restrict__ ErrT) {
    quad_load_scales<M0 + QPF>(s, qs);
    quad_step<M0, 0>(s, Up, dd, p, qmin, qmax);
    quad_step<M0, 1>(s, Up, dd, p, qmin, qmax);
    quad_step<M0, 2>(s, Up, dd, p, qmin, qmax);
    quad_step<M0, 3>(s, Up, dd, p, qmin, qmax);
    // every lane now holds the level, the error and the rounding residual of its own column 4 M0 + p
    // the row's loss: the four columns' terms added in column order, in every lane of the quad alike (the
    // row-per-lane kernel's and the oracle's order: ascending column)
    {
        const float t = (s.dv * s.dv) * dd[BS + 4 * M0 + p];
        s.lsum = s.lsum + quad_bcast<0>(t);
        s.lsum = s.lsum + quad_bcast<1>(t);
        s.lsum = s.lsum + quad_bcast<2>(t);
        s.lsum = s.lsum + quad_bcast<3>(t);
    }
#if defined(QT_SWEEP_LAB) && QT_SWEEP_LAB == 2     // lab build 2 (timing only): no stores inside the loop
    if (valid && M0 == QM - 1) {
#else
    if (valid) {
#endif
        const int c = 4 * M0 + p;
        Qt[(size_t)(i1 + c) * R + row] = (int8_t)s.qv;
        ErrT[(size_t)c * R + row] = s.ev;
    }
}

template <int M0>
__device__ __forceinline__ void quad_run(QuadState& s, const QuadScales& qs, const float* __restrict__ Up,
                                         const float* __restrict__ dd, int p,
                                         float qmin, float qmax, bool valid, int row, int R, int i1, int nm,
                                         int8_t* __restrict__ Qt, float* __restrict__ ErrT) {
    if constexpr (M0 < QM) {
        if (M0 >= nm) return;       // ragged last block: cnt = 4 nm columns (wave-uniform)
        quad_macro_step<M0>(s, qs, Up, dd, p, qmin, qmax, valid, row, R, i1, Qt, ErrT);
        quad_run<M0 + 1>(s, qs, Up, dd, p, qmin, qmax, valid, row, R, i1, nm, Qt, ErrT);
    }
}

__global__ __launch_bounds__(QTHREADS) void sweep_quad_kernel(float* __restrict__ W, int R, int K,
                                                              const float* __restrict__ U,
                                                              const float* __restrict__ scale_t,
                                                              const float* __restrict__ zp_t,
                                                              const int32_t* __restrict__ g_idx, int i1, int cnt,
                                                              float qmin, float qmax, int8_t* __restrict__ Qt,
                                                              float* __restrict__ ErrT, float* __restrict__ loss,
                                                              int prio, SweepGroups sg) {
    qt_set_chain_prio(prio);
    if (sg.n > 1) {
        const int g = sweep_group_of(sg, blockIdx.x * QROWS);
        U += (size_t)g * sg.bsU;
        g_idx += (size_t)g * K;
    }
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Up = sm;                      // [BS][QL][UP_P]
    float* dd = sm + BS * UP_C;          // [BS] diag, [BS] 1 / diag^2
    const int tid = threadIdx.x;
    const int p = tid & (QL - 1);
    const int row = blockIdx.x * QROWS + (tid >> 2);
    const bool valid = row < R;
    const int rowc = valid ? row : R - 1;
    const int nm = cnt >> 2;             // cnt % 4 == 0 (K % 4 == 0)

    {   // the U block, strictly above the diagonal, per (step, lane class); NV independent float4 loads per thread
        const float* ublk = U + (size_t)i1 * K + i1;
        constexpr int NV = BS * BS / 4 / QTHREADS;
        f32x4 v[NV];
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int e4 = tid + QTHREADS * r;                 // float4 index in the 128 x 32 grid
            const int i = e4 >> 5, j = (e4 & 31) * 4;
            const int ic = i < cnt ? i : cnt - 1, jc = j < cnt ? j : cnt - 4;
            v[r] = *(const f32x4*)(ublk + (size_t)ic * K + jc);
        }
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const int e4 = tid + QTHREADS * r;
            const int i = e4 >> 5, m = e4 & 31, j = m * 4;
            const bool inside = i < cnt && j < cnt;
#pragma unroll
            for (int q = 0; q < 4; ++q) Up[i * UP_C + q * UP_P + m] = (inside && j + q > i) ? v[r][q] : 0.0f;
        }
    }
    if (tid < BS) {
        const float d = (tid < cnt) ? U[(size_t)(i1 + tid) * K + (i1 + tid)] : 1.0f;
        dd[tid] = d;
        dd[BS + tid] = 1.0f / (d * d);   // only the loss uses it (tolerance 1e-6, not part of the bit-exact contract)
    }
    QuadState s;
    {
        const float* wrow = W + (size_t)rowc * K + i1 + p;
#pragma unroll
        for (int m = 0; m < QM; ++m) s.w[m] = m < nm ? wrow[4 * m] : 0.0f;
    }
    // the row's running loss, read with the row itself: as `loss[row] = loss[row] + ...` at the end it was one more
    // memory round trip behind the last column step of every block launch
    const float loss_in = (valid && p == 0) ? loss[row] : 0.0f;
    const QuadScales qs = {scale_t + rowc, zp_t + rowc, g_idx + i1 + p, R, nm};
    quad_load_scales<0>(s, qs);
    quad_load_scales<1>(s, qs);
    quad_load_scales<2>(s, qs);
    static_assert(QPF == 3, "the prologue loads the first QPF macro-steps' scales");
    s.lsum = 0.0f;
    s.qv = 0.0f;
    s.ev = 0.0f;
    s.dv = 0.0f;
    __syncthreads();
    // consume the early loss read here, where the row's own loads are waited for anyway: its use at the end would
    // otherwise carry an s_waitcnt vmcnt(0) that also waits for every store of the write-back to be acknowledged
    asm volatile("" ::"v"(loss_in));
    quad_fetch<0>(s, Up, dd, p);

#if !defined(QT_SWEEP_LAB) || QT_SWEEP_LAB != 1   // lab build 1 (tools/sweep_lab.sh, timing only): prologue + epilogue alone
    quad_run<0>(s, qs, Up, dd, p, qmin, qmax, valid, row, R, i1, nm, Qt, ErrT);
#endif

    if (valid) {   // dequantised values back to W (upstream: W[:, i1:i2] = Q1)
        float* wrow = W + (size_t)row * K + i1 + p;
#pragma unroll
        for (int m = 0; m < QM; ++m)
            if (m < nm) wrow[4 * m] = s.w[m];
    }
    if (valid && p == 0) loss[row] = loss_in + s.lsum / 2.0f;
}

}  // namespace

// OPT-IN, NOT the parity contract (QT_SWEEP_FAR=bf16x3): the far update as a three-plane bf16 product (gemm3_tn.h) instead
// of the f32-MFMA chain.  Every product is fp32-accurate (what the planes drop is <= 2^-24 of it) but the sum is not the
// ascending-k fmaf chain the oracle fixes (that order is this repo's choice, upstream leaves it to BLAS),
// so the outputs agree with the oracle's to a rate, not to the bit -- as they already do across two factorisations.
// What it buys: the far update is 1.5 TFLOP of f32 MFMA per Llama-3-8B layer (~13 ms); as 9 TFLOP of bf16 MFMA ~7 ms.
static bool sweep_far_bf16x3(int R, int K) {
    const char* e = getenv("QT_SWEEP_FAR");
    return e && e[0] == 'b' && K % 8 == 0 && R % 4 == 0;
}
struct FarPlan {
    int64_t ldU, ldE;          // plane pitches (elements): U planes [K][ldU], error planes [kmax][ldE]
    size_t u_bytes, e_bytes, tab_bytes;
    int Tm, Tn_max, kmax;
};
static FarPlan far_plan(int R, int K, int blocksize) {
    FarPlan f;
    f.kmax = SWEEP_MAX_BATCH * blocksize;
    f.ldU = (int64_t)qt_align_up((size_t)K, 256);
    f.ldE = (int64_t)qt_align_up((size_t)R, 256);
    f.u_bytes = qt_align_up((size_t)3 * K * f.ldU * 2, 256);
    f.e_bytes = qt_align_up((size_t)3 * f.kmax * f.ldE * 2, 256);
    f.Tm = (R + 255) / 256;
    f.Tn_max = (K + 255) / 256;
    f.tab_bytes = qt_align_up((size_t)f.Tm * f.Tn_max * sizeof(G3Item), 256);
    return f;
}

extern "C" size_t qt_gptq_sweep_workspace_bytes(int R, int K, int blocksize) {
    if (R <= 0 || blocksize <= 0) return 0;
    size_t n = qt_align_up((size_t)SWEEP_MAX_BATCH * blocksize * R * 4, 256) + 256;  // ErrT[batch][blocksize][R]
    if (sweep_far_bf16x3(R, K)) {
        const FarPlan f = far_plan(R, K, blocksize);
        n += f.u_bytes + f.e_bytes + f.tab_bytes;
    }
    return n;
}

static int sweep_run(float* W, int R, int K, const float* U, const float* scale_t, const float* zp_t, int G,
                     const int32_t* g_idx, int blocksize, int num_bits, int8_t* Qt, float* loss, void* workspace,
                     size_t workspace_bytes, hipStream_t stream, const SweepGroups& sg) {
    QT_CHECK_ARG(W && U && scale_t && zp_t && g_idx && Qt && loss, "qt_gptq_sweep: null pointer");
    QT_CHECK_ARG(R > 0 && K > 0 && G > 0, "qt_gptq_sweep: bad shape R=%d K=%d G=%d", R, K, G);
    QT_CHECK_ARG(blocksize == BS, "qt_gptq_sweep: blocksize=%d unsupported (this build: 128)", blocksize);
    QT_CHECK_ARG(num_bits >= 2 && num_bits <= 8, "qt_gptq_sweep: num_bits=%d", num_bits);
    QT_CHECK_ARG(K % 4 == 0 && ((uintptr_t)W & 15) == 0, "qt_gptq_sweep: K %% 4 and 16-byte aligned W required");
    const size_t need = qt_gptq_sweep_workspace_bytes(R, K, blocksize);
    if (!workspace || workspace_bytes < need) {
        qt_set_error("qt_gptq_sweep: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    float* ErrT = (float*)qt_align_up((size_t)workspace, 256);
    const bool far3 = sweep_far_bf16x3(R, K) && sg.n <= 1;      // the opt-in three-plane far update: single problem only
    FarPlan fp{};
    unsigned short *Upl = nullptr, *Epl = nullptr;
    G3Item* far_tab = nullptr;
    if (far3) {
        fp = far_plan(R, K, blocksize);
        char* q = (char*)ErrT + qt_align_up((size_t)SWEEP_MAX_BATCH * blocksize * R * 4, 256);
        Upl = (unsigned short*)q;
        q += fp.u_bytes;
        Epl = (unsigned short*)q;
        q += fp.e_bytes;
        far_tab = (G3Item*)q;
    }
    const float qmin = -(float)(1 << (num_bits - 1)), qmax = (float)((1 << (num_bits - 1)) - 1);
    static QtOncePerDevice lds_attr;
    QT_HIP(lds_attr.run([&] {
        return hipFuncSetAttribute((const void*)sweep_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)SWEEP_LDS);
    }));
    static QtOncePerDevice lds_attr_q;
    QT_HIP(lds_attr_q.run([&] {
        return hipFuncSetAttribute((const void*)sweep_quad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)QUAD_LDS);
    }));
    // QT_SWEEP_BLOCK=row: the row-per-lane block kernel (the round-1/2 form; A/B and cross-check); default: four lanes per row
    const char* blk_env = getenv("QT_SWEEP_BLOCK");
    const bool quad = !(blk_env && blk_env[0] == 'r');
    QT_HIP(hipMemsetAsync(loss, 0, (size_t)R * 4, stream));
    // Lazy far update: the blocks of a batch update only the batch's own later columns right away
    // (k = 128, few columns); everything to the right of the batch gets the batch's chains in ONE
    // pass over W (chain_len = 128: same roundings, in the same order, as one pass per block), so
    // the read-modify-write traffic on W drops by the batch size.
    // Batch size: 4 blocks, 8 from K = 8192 up -- a longer batch halves the far update's traffic on W again but puts
    // more near-update columns between two block kernels (K = 14336 / R = 4096: 11.74 -> 11.35 ms; K = 4096 /
    // R = 28672: 5.9 -> 6.0 ms; K = 8192: +-0.5 %, profiles/r03_sweep_batch_ab.txt).  The bits do not depend on it.  QT_SWEEP_BATCH forces it.
    const int batch_blocks = [&] {
        const char* e = getenv("QT_SWEEP_BATCH");
        const int b = e ? atoi(e) : (K >= 8192 ? 8 : 4);
        return b < 1 ? 1 : (b > SWEEP_MAX_BATCH ? SWEEP_MAX_BATCH : b);
    }();
    const int prio = qt_chain_prio();
    // stacked problems: an update product's B operand (rows of U) depends on the row group of the output tile
    auto set_groups = [&](SgemmArgs& g) {
        if (sg.n <= 1) return;
        g.n_groups = sg.n;
        g.group_bsB = sg.bsU;
        for (int i = 0; i < sg.n; ++i) g.group_m_end[i] = sg.row_end[i];
    };
    if (far3) {
        // planes of U once (pad columns zeroed: edge tiles read them), the tile list once: column-major in tiles, so a
        // far update over the first Tn' tile columns right of the batch is a prefix of it
        QT_HIP(hipMemsetAsync(Upl, 0, fp.u_bytes, stream));
        QT_HIP(hipMemsetAsync(Epl, 0, fp.e_bytes, stream));
        int rc = qt_split3_launch(U, K, K, K, Upl, fp.ldU, (int64_t)K * fp.ldU, 0, 0, 0, stream);
        if (rc) return rc;
        static std::mutex m;
        static std::map<std::tuple<int, int, int>, G3Item*> tabs;      // pinned, per (Tm, Tn_max, chunks)
        const int c_end = BS * batch_blocks / G3_CHUNK_ROWS;
        G3Item* host = nullptr;
        {
            std::lock_guard<std::mutex> lock(m);
            const auto key = std::make_tuple(fp.Tm, fp.Tn_max, c_end);
            auto it = tabs.find(key);
            if (it == tabs.end()) {
                if (hipHostMalloc((void**)&host, fp.tab_bytes, hipHostMallocDefault) != hipSuccess) {
                    qt_set_error("qt_gptq_sweep: no pinned memory for the far update's tile table");
                    return QT_ERR_HIP;
                }
                int n = 0;
                for (int tj = 0; tj < fp.Tn_max; ++tj)
                    for (int ti = 0; ti < fp.Tm; ++ti) host[n++] = {(ti << 16) | tj, 0, c_end, -1};
                tabs[key] = host;
            } else {
                host = it->second;
            }
        }
        QT_HIP(hipMemcpyAsync(far_tab, host, (size_t)fp.Tm * fp.Tn_max * sizeof(G3Item), hipMemcpyHostToDevice, stream));
    }
    for (int b0 = 0; b0 < K; b0 += BS * batch_blocks) {
        const int bend = (b0 + BS * batch_blocks < K) ? b0 + BS * batch_blocks : K;   // first column right of the batch
        for (int i1 = b0; i1 < bend; i1 += BS) {
            const int i2 = (i1 + BS < K) ? i1 + BS : K;
            const int cnt = i2 - i1;
            float* err_blk = ErrT + (size_t)(i1 - b0) * R;
            qt_prof_mark(QT_PROF_SWEEP_BLOCK, stream);
            if (quad)
                hipLaunchKernelGGL(sweep_quad_kernel, dim3((R + QROWS - 1) / QROWS), dim3(QTHREADS), QUAD_LDS, stream, W,
                                   R, K, U, scale_t, zp_t, g_idx, i1, cnt, qmin, qmax, Qt, err_blk, loss, prio, sg);
            else
                hipLaunchKernelGGL(sweep_block_kernel, dim3((R + ROWS - 1) / ROWS), dim3(ROWS), SWEEP_LDS, stream, W, R,
                                   K, U, scale_t, zp_t, g_idx, i1, cnt, qmin, qmax, Qt, err_blk, loss, prio, sg);
            qt_prof_mark(QT_PROF_SWEEP_BLOCK, stream);
            QT_LAUNCH_CHECK();
            if (i2 < bend) {   // near update: the rest of this batch
                SgemmArgs g;
                g.A = err_blk; g.lda = R;
                g.B = U + (size_t)i1 * K + i2; g.ldb = K;
                g.Cin = W + i2; g.ldcin = K;
                g.Cout = W + i2; g.ldcout = K;
                g.M = R; g.N = bend - i2; g.kdim = cnt; g.k_mode = SG_K_FULL; g.mode = SG_MODE_SUB;
                set_groups(g);
                const int rc = qt_sgemm_tn(g, stream);
                if (rc) return rc;
            }
        }
        if (bend < K && far3 && bend - b0 == BS * batch_blocks) {
            // far update as a three-plane product: error rows of the batch -> planes, then W[:, bend:] -= E^T U[b0:bend, bend:]
            int rc = qt_split3_launch(ErrT, R, bend - b0, R, Epl, fp.ldE, (int64_t)fp.kmax * fp.ldE, 0, 0, 0, stream);
            if (rc) return rc;
            G3Args a;
            a.Apl = Epl; a.ld = fp.ldE; a.plane_stride = (int64_t)fp.kmax * fp.ldE; a.rowA0 = 0; a.colA0 = 0; a.colmax = (int)fp.ldE;
            a.Bpl = Upl; a.ldB = fp.ldU; a.plane_strideB = (int64_t)K * fp.ldU; a.rowB0 = b0; a.colB0 = bend; a.colmaxB = (int)fp.ldU;
            a.M = R; a.N = K - bend; a.C = W + bend; a.ldc = K; a.mode = G3_SUB;
            a.slabs = nullptr; a.red = nullptr; a.n_red = 0;
            a.items = far_tab; a.n_items = fp.Tm * ((K - bend + 255) / 256);
            rc = qt_gemm3_launch(a, stream);
            if (rc) return rc;
        } else if (bend < K) {        // far update: every chain of the batch, one pass
            SgemmArgs g;
            g.A = ErrT; g.lda = R;
            g.B = U + (size_t)b0 * K + bend; g.ldb = K;
            g.Cin = W + bend; g.ldcin = K;
            g.Cout = W + bend; g.ldcout = K;
            g.M = R; g.N = K - bend; g.kdim = bend - b0; g.k_mode = SG_K_FULL; g.mode = SG_MODE_SUB;
            g.chain_len = BS;
            set_groups(g);
            const int rc = qt_sgemm_tn(g, stream);
            if (rc) return rc;
        }
    }
    return QT_OK;
}

extern "C" int qt_gptq_sweep(float* W, int R, int K, const float* U, const float* scale_t, const float* zp_t, int G,
                             const int32_t* g_idx, int blocksize, int num_bits, int8_t* Qt, float* loss,
                             void* workspace, size_t workspace_bytes, qt_stream_t stream_) {
    SweepGroups sg{};
    return sweep_run(W, R, K, U, scale_t, zp_t, G, g_idx, blocksize, num_bits, Qt, loss, workspace, workspace_bytes,
                     (hipStream_t)stream_, sg);
}

extern "C" int qt_gptq_sweep_grouped(float* W, int R, int K, const float* U, int64_t strideU, int n_groups,
                                     const int32_t* row_end, const float* scale_t, const float* zp_t, int G,
                                     const int32_t* g_idx, int blocksize, int num_bits, int8_t* Qt, float* loss,
                                     void* workspace, size_t workspace_bytes, qt_stream_t stream_) {
    QT_CHECK_ARG(n_groups >= 1 && n_groups <= SG_MAX_GROUPS && row_end, "qt_gptq_sweep_grouped: 1 <= n_groups <= %d", SG_MAX_GROUPS);
    SweepGroups sg{};
    sg.n = n_groups;
    sg.bsU = strideU;
    int prev = 0;
    for (int g = 0; g < n_groups; ++g) {
        // 128: the update products' tile height (no tile may straddle two factors); the block kernels need 64
        QT_CHECK_ARG(row_end[g] > prev && (row_end[g] % 128 == 0 || g == n_groups - 1),
                     "qt_gptq_sweep_grouped: group %d ends at row %d (ascending; every boundary but the last a multiple of 128)", g,
                     row_end[g]);
        sg.row_end[g] = row_end[g];
        prev = row_end[g];
    }
    QT_CHECK_ARG(prev == R, "qt_gptq_sweep_grouped: the last group ends at row %d, R = %d", prev, R);
    QT_CHECK_ARG(n_groups == 1 || strideU >= (int64_t)K * K, "qt_gptq_sweep_grouped: strideU < K*K");
    return sweep_run(W, R, K, U, scale_t, zp_t, G, g_idx, blocksize, num_bits, Qt, loss, workspace, workspace_bytes,
                     (hipStream_t)stream_, sg);
}
