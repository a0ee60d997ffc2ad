// a7: Gram accumulation G += X^T X  (upstream accumulate_hessian, reached
// through src/quantool/methods/llm_compressor/base.py:161).
//
// MFMA-bound (arithmetic intensity ~K flop/B).  X is [tokens][channels] bf16, so BOTH MFMA
// operands are "k-strided" (the reduction index is the row index): X is staged row-major into
// LDS by LDS-DMA (global_load_lds, 16 B/lane) with the swizzle on the SOURCE address, and read
// back with the hardware transposing read ds_read_b64_tr_b16.
//
// Loop structure.  Because the reduction index is the token index and X rows are
// channel-contiguous, the natural staging unit is a slice of TOKENS, not of channels:
//   unit  = 16 tokens x (256 A-panel + 256 B-panel channels) = 16 KiB = exactly one
//           v_mfma_f32_32x32x16_bf16 k-step for the whole 256x256 tile;
//   ring  = 8 units of LDS (128 KiB); unit u is consumed in phase u and its slot is re-filled
//           with unit u+8; the LDS-DMA of unit u+6 is issued in phase u, so FIVE units (80 KiB per
//           CU) are always in flight behind a COUNTED s_waitcnt vmcnt(10) -- never 0 in the loop;
//   phase = { 12 transposing reads of unit u ; issue unit u+6 ; vmcnt(10) ; s_barrier ;
//             lgkmcnt(0) ; 8 MFMAs ; s_barrier };
//   waves 4-7 run the same program one barrier behind waves 0-3, so on every SIMD one wave's
//   LDS reads / DMA issue run under its partner's 8 MFMAs (ping-pong), and a unit's slot is free
//   two phases after it was read (hazard analysis at the kernel).
//
// Work decomposition: lower-triangular 256x256 output tiles.  Whole "rounds" of 256 tiles run one
// workgroup per tile over ALL tokens and add their tile straight into G from the epilogue (no slab
// round trip); the remaining tiles are split over token chunks into fp32 slabs that
// xtx_reduce_kernel sums in fixed order (deterministic; no atomics).  Workgroups are mapped so that
// the 32 resident on one XCD form a 4x8 block of tiles and all 256 of a round form a 16x16 block
// walking the tokens together: a panel missed by one XCD's L2 is in the Infinity Cache for the rest.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <type_traits>
#include <vector>

#include "common.h"
#include "ring_pipe.h"

namespace {

constexpr int BT = 256;                       // output tile edge (channels)
constexpr int BKT = 64;                       // tokens per token tile (split / tail granularity)
constexpr int UT = 16;                        // tokens per unit = one MFMA k-step
constexpr int UNIT_BYTES = UT * 2 * BT * 2;   // A + B panels, 16 KiB
constexpr int RING = 8;                       // units resident in LDS (128 KiB)
constexpr int NTHREADS = 512;                 // 8 waves: 2 (M) x 4 (N), 128x64 outputs per wave
constexpr int NUM_CU = 256;

struct XtxParams {
    const void* X;       // [tokens, ldx] 16-bit elements (bf16 or fp16: the kernel is instantiated for each)
    const void* tail;    // zero-padded [64, K] staging of the ragged last token tile
    int64_t ldx;
    int K;
    int n_tt;       // token tiles (of 64) including the tail tile
    int has_tail;
    const int* tile_tab;  // [n_tiles] (ti << 16) | tj in locality order (xtx_tile_order)
    int n_direct;   // logical items [0, n_direct): one tile each over all tokens, added into G
    int n_rem;      // tiles after those, each split over s2 token chunks -> slabs
    int s2;
    float* slabs;   // [s2][n_rem][256*256]
    float* G;
    int map_mode;   // 0: rounds of 256 with XCD-contiguous blocks of 32; 1: identity
    int wrap_units; // > 0: timing-only locality ablation (xtx_kernel<true>)
    unsigned* progress;  // [rounds][256] progress words of the direct items (zeroed per launch), or null
    int thr_win;         // throttle: units a workgroup may lead the slowest started member by
    int thr_nap;         // throttle: s_sleep argument of one nap (x64 cycles)
    int thr_chk;         // throttle: units between two progress checks (a power of two, 8..256; 32 by default)
    // DOT variant of xtx16_kernel (qt_xtx_frobenius): instead of storing its tile, every item writes
    // <H tile, its partial tile> (lower triangle, off-diagonal entries twice) to dot_partials[item]
    const float* H;
    double* dot_partials;
    int batch_items;          // DOT: work items per problem (the grid holds n_batch problems back to back) ...
    int64_t x_batch_bytes;    // ... whose X matrices lie this many bytes apart
};

// WRAP = true is a TIMING-ONLY ablation (wrong results; only in builds with -DQT_XTX_ABLATION, then
// QT_XTX_ABLATE_WRAP=<units>): the source
// pointer wraps every wrap_units units, so the footprint every workgroup streams is that window --
// L2-resident for small windows, Infinity-Cache-resident for medium ones (profiles/r02_xtx_locality.md).
//
// THROTTLE (speed only, never correctness): the 256 workgroups of a round stream the same ~32 panels.
// While they stay within a few thousand tokens of each other, a panel missed by one XCD's L2 is in the
// 256 MiB Infinity Cache for the other seven; once they drift apart it comes from HBM, and the chip
// -- power-limited in this kernel -- lowers its clock (profiles/r02_xtx_locality_sweep.txt: 1.31 PF with
// HBM-served misses, 1.43-1.50 with on-die ones).  Every thr_chk (32) units wave 0 of a direct item
// publishes its unit index and snapshots the round's 256 progress words (one 1 KiB LDS-DMA, read
// thr_chk units later, when the counted vmcnt has long retired it); a workgroup more than thr_win
// units ahead of the slowest STARTED, unfinished member sleeps a bounded while (its other waves wait
// at the phase barrier).  No workgroup ever waits FOR another: no spin, no dependence on residency.

template <bool WRAP, bool F16, bool THROTTLE>
__global__ __launch_bounds__(NTHREADS, 2) void xtx_kernel(XtxParams p) {
    constexpr int LEAD = 6;     // unit u+LEAD is issued in phase u  (LEAD <= RING-2, see the hazard analysis)
    // ONE LDS object: the ring (+ 1 KiB of scratch for the throttle's progress snapshot).
    // Unit image: [4 channel groups: A-lo, A-hi, B-lo, B-hi][16 tokens][256 B].
    __shared__ __attribute__((aligned(16))) char ring[RING * UNIT_BYTES + (THROTTLE ? 1024 : 0)];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;
    const bool group_b = wave >= 4;  // wave-uniform

    // ---- workgroup -> work item ------------------------------------------------------------
    // Blocks are dealt round-robin over the 8 XCDs (b and b+8 share one; speed only, never
    // correctness).  Within each round of 256 consecutive blocks, the 32 blocks of one XCD take 32
    // consecutive logical items = one 4x8 block of tiles of the table.
    int logical;
    {
        const int b = blockIdx.x, nwg = gridDim.x;
        if (p.map_mode == 0) {
            const int round = b >> 8, rb = b & 255;
            const int m = min(256, nwg - (round << 8));
            const int xcd = rb & 7, idx = rb >> 3, q8 = m >> 3, r8 = m & 7;
            logical = (round << 8) + (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        } else {
            logical = b;
        }
    }
    int tile_idx, tt0, cnt, slab_idx = -1;
    if (logical < p.n_direct) {
        tile_idx = logical;
        tt0 = 0;
        cnt = p.n_tt;
    } else {
        const int l2 = logical - p.n_direct;
        const int chunk = l2 / p.n_rem;
        tile_idx = p.n_direct + (l2 - chunk * p.n_rem);
        const int base_cnt = p.n_tt / p.s2, rem = p.n_tt % p.s2;
        tt0 = chunk * base_cnt + (chunk < rem ? chunk : rem);
        cnt = base_cnt + (chunk < rem ? 1 : 0);
        slab_idx = l2;
    }
    const int tt_packed = p.tile_tab[tile_idx];
    const int ti = tt_packed >> 16, tj = tt_packed & 0xFFFF;
    const int nu = cnt * (BKT / UT);  // units of this item (a multiple of 4)
    const int K = p.K;
    const size_t ld2 = (size_t)p.ldx * 2;  // row pitch in bytes (the tail staging uses the same pitch)

    // ---- staging geometry: two LDS-DMA instructions per thread per unit (A panel, B panel) ----
    // wave w fills token rows 4*(w&3)..+3 of channel group (w>>2) (+2 for the B panel): 1 KiB,
    // lane-linear; the XOR swizzle of the 16-B chunk index is applied to the SOURCE address.
    const int rsub = lane >> 4;                       // token row inside the wave's 4-row piece
    const int lch = (lane & 15) ^ (rsub << 2);        // logical 16-B chunk held at physical chunk lane&15
    const int trow = 4 * (wave & 3) + rsub;           // token row inside the unit
    int colA = ti * BT + (wave >> 2) * 128 + lch * 8;
    int colB = tj * BT + (wave >> 2) * 128 + lch * 8;
    colA = colA > K - 8 ? K - 8 : colA;               // edge tiles: clamp (masked at the store)
    colB = colB > K - 8 ? K - 8 : colB;
    const unsigned voffA = (unsigned)((size_t)trow * ld2 + (size_t)colA * 2);
    const unsigned voffB = (unsigned)((size_t)trow * ld2 + (size_t)colB * 2);
    const unsigned ring_lds = (unsigned)(size_t)(QT_LDS char*)ring;
    const unsigned dst_wave =
        __builtin_amdgcn_readfirstlane(ring_lds + (wave >> 2) * 4096 + (wave & 3) * 1024);  // wave-uniform

    // scalar source pointers: first token row of unit i.  The ragged last token tile (if this item
    // reaches it) is read from the zero-padded staging; everything before it is contiguous in X.
    const bool ends_in_tail = p.has_tail && (tt0 + cnt == p.n_tt);
    const int i_tail = ends_in_tail ? nu - (BKT / UT) : 0x7fffffff;  // first unit taken from the staging
    const size_t ustride = (size_t)UT * ld2;
    auto unit_src = [&](int i) -> const char* {
        return i >= i_tail ? (const char*)p.tail + (size_t)(i - i_tail) * ustride
                           : (const char*)p.X + ((size_t)tt0 * BKT + (size_t)i * UT) * ld2;
    };
    const char* run_src = (const char*)p.X + (size_t)tt0 * BKT * ld2;  // pointer of the next unit to issue
    auto issue = [&](int i, int slot) {
        const unsigned d = dst_wave + (unsigned)slot * UNIT_BYTES;
        glds16_pair(voffA, voffB, unit_src(i), d, d + 8192);
    };
    // steady-state issue: no tail test, the source pointer just advances (units are consecutive rows)
    int wrap_cnt = 0;
    auto issue_running = [&](int slot) {
        const unsigned d = dst_wave + (unsigned)slot * UNIT_BYTES;
        glds16_pair(voffA, voffB, run_src, d, d + 8192);
        run_src += ustride;
        if (WRAP && ++wrap_cnt == p.wrap_units) {
            wrap_cnt = 0;
            run_src -= (size_t)p.wrap_units * ustride;
        }
    };

    // ---- throttle state (wave 0 of a direct item only) ----
    unsigned* prog_round = nullptr;
    int prog_member = 0;
    if (THROTTLE && p.progress && slab_idx < 0) {
        prog_round = p.progress + (size_t)(logical >> 8) * 256;
        prog_member = logical & 255;
    }
    const bool throttled = THROTTLE && prog_round != nullptr && wave == 0;   // wave-uniform
    bool snap_pending = false;
    auto throttle_step = [&](int u) {
        if (snap_pending) {
            // the snapshot issued thr_chk units ago: 4 progress words per lane (0 = not started)
            const unsigned* sp = (const unsigned*)(ring + RING * UNIT_BYTES) + lane * 4;
            unsigned m = 0xffffffffu;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned v = sp[e];
                m = (v != 0 && v < m) ? v : m;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned o = (unsigned)__shfl_xor((int)m, off);
                m = o < m ? o : m;
            }
            const int slowest = (int)__builtin_amdgcn_readfirstlane(m);     // 0x7fffffff: finished members
            const int lag = (u - p.thr_chk + 1) - slowest;                    // in units, at snapshot time
            if (lag > p.thr_win) {
                int naps = (lag - p.thr_win) / 24 + 1;                        // one nap of 127 ~ 8 k cycles ~ 20 units
                naps = naps > 6 ? 6 : naps;
                for (int i = 0; i < naps; ++i) {
                    if (p.thr_nap >= 96) __builtin_amdgcn_s_sleep(127);
                    else if (p.thr_nap >= 48) __builtin_amdgcn_s_sleep(64);
                    else __builtin_amdgcn_s_sleep(32);
                }
            }
        }
        if (lane == 0) {
            unsigned* dst = prog_round + prog_member;
            const unsigned val = (unsigned)(u + 1);
            asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(val) : "memory");
        }
        glds16_snapshot((unsigned)lane * 16u, prog_round, ring_lds + RING * UNIT_BYTES);
        snap_pending = true;
    };

    // ---- fragment read geometry (per lane), byte offsets inside a unit ----
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

    // ---- the pipeline --------------------------------------------------------------------------
    // Intervals between consecutive workgroup barriers are numbered; waves 0-3 (group A) run
    // LOAD(u) in interval 2u and MATH(u) in 2u+1, waves 4-7 (group B) one interval later.
    //   RAW: unit u is read in intervals 2u (A) / 2u+1 (B).  Every wave waits for its own DMA of
    //        unit u (counted vmcnt) in LOAD(u-1), i.e. in intervals 2u-2 / 2u-1, and the barrier
    //        that ends interval 2u-1 follows both: "read one phase after the wait that retires it".
    //   WAR: unit u's reads retire at the lgkmcnt(0) that opens MATH(u): intervals 2u+1 (A) / 2u+2 (B).
    //        Its slot is re-filled with unit u+8, issued in LOAD(u+2): intervals 2u+4 / 2u+5, after
    //        the barriers that end 2u+2 and 2u+3.  (LEAD 7 would put group A's issue into 2u+2, beside
    //        group B's outstanding reads.)
    //   In flight at every wait: 5 units = 10 LDS-DMA instructions per wave (80 KiB per CU).
    s16x8 fa[4], fb[2];
    auto drain_wait = [&](int u) {
        // no unit beyond nu-1 exists: allow exactly the units after u+1 to stay in flight
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
        // ---- LOAD ----
        const char* base = ring + S * UNIT_BYTES;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa[mi] = tr_load8(base + aoff[mi]);
#ifdef QT_XTX_ABL_LDS    // lab builds only, TIMING-ONLY (wrong results): five fragment reads per phase instead of six
        fb[0] = tr_load8(base + boff[0]);
        fb[1] = fb[0];
#else
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fb[ni] = tr_load8(base + boff[ni]);
#endif
        if (STEADY) {
            issue_running(ISLOT);
            wait_vmcnt<10>();   // everything up to unit u+1 has landed; 5 units stay in flight
        } else if (u + LEAD < nu) {
            issue(u + LEAD, ISLOT);
            wait_vmcnt<10>();
        } else {
            drain_wait(u);
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MATH ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#ifdef QT_XTX_ABL_HALF   // lab builds only, TIMING-ONLY (wrong results): half of the MFMAs of every phase
        constexpr int MI_N = 2;
#else
        constexpr int MI_N = 4;
#endif
#pragma unroll
        for (int mi = 0; mi < MI_N; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                acc[mi][ni] = mfma16<F16>(fa[mi], fb[ni], acc[mi][ni]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto body8 = [&](auto steady_c, int u) {
        if (THROTTLE && throttled && (u & (p.thr_chk - 1)) == 0) throttle_step(u);
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
        // prologue: units 0..LEAD-1 in flight, unit 0 landed
#pragma unroll
        for (int i = 0; i < LEAD; ++i)
            if (i < nu) issue(i, i);
        if (nu >= LEAD) wait_vmcnt<2 * (LEAD - 1)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (group_b) __builtin_amdgcn_s_barrier();  // stagger: group B runs one interval behind
        __builtin_amdgcn_sched_barrier(0);

        // steady body: every phase issues a unit that exists and does not come from the tail staging
        const int steady_end = (nu < i_tail ? nu : i_tail);
        run_src += (size_t)LEAD * ustride;
        int u = 0;
        for (; u + 8 + LEAD <= steady_end; u += 8) body8(std::true_type{}, u);
        for (; u + 8 <= nu; u += 8) body8(std::false_type{}, u);
        if (u < nu) {  // nu is a multiple of 4: one half body left
            phase(std::integral_constant<int, 0>{}, std::false_type{}, u);
            phase(std::integral_constant<int, 1>{}, std::false_type{}, u + 1);
            phase(std::integral_constant<int, 2>{}, std::false_type{}, u + 2);
            phase(std::integral_constant<int, 3>{}, std::false_type{}, u + 3);
        }
        if (!group_b) __builtin_amdgcn_s_barrier();  // pairs with group B's last barrier
        wait_vmcnt<0>();
        if (THROTTLE && throttled && lane == 0) {   // finished: never the slowest member again
            unsigned* dst = prog_round + prog_member;
            const unsigned val = 0x7fffffffu;
            asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(val) : "memory");
        }
    }

    // ---- epilogue ----------------------------------------------------------------------------
    const int jl = lane & 31, ih = 4 * (lane >> 5);
    if (slab_idx < 0) {
        // the only workgroup of this tile in this launch: G += acc (launches on a stream are ordered)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) {
                const int gj = tj * BT + wave_n * 64 + ni * 32 + jl;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gi = ti * BT + wave_m * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + ih;
                    if (gi < K && gj < K) {
                        float* dst = p.G + (size_t)gi * K + gj;
                        *dst = *dst + acc[mi][ni][r];
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

// ---------------------------------------------------------------------------------------------
// The same pipeline on v_mfma_f32_16x16x32_{bf16,f16} (the default wherever the throttle is not in play;
// QT_XTX_SHAPE=16|32 forces a shape).  The chip is power-limited in
// this kernel and can hold a higher clock on the 16x16x32 shape than on 32x32x16 at equal cycles per
// flop (guide, DVFS give-back item 7), so the shape is decided by wall time, not by cycles.
// k = 32 tokens per MFMA, so a PHASE consumes a pair of units ("double": 32 tokens, 32 KiB) and the ring
// grows to 10 units = 5 doubles (160 KiB: all of the CU's LDS, so this variant carries no throttle --
// a register-snapshot form of it was built and measured slower, -1 % at K = 14336).
// Lane groups 0,1 of a fragment read tokens 0-15 from the first unit of the pair, groups 2,3 tokens
// 16-31 from the second; inside a 32-lane half the two groups read token rows r and r+8 of one unit,
// so the source-side swizzle also folds token-row bit 3 into the chunk index (conflict-free: the 32
// lanes of a half then cover all 16 chunks x 2 halves of a 256-byte bank row exactly once).
//   RAW / WAR: as above with "unit" := double, 5 slots, LEAD 3 doubles: two doubles (8 LDS-DMA
//   instructions per wave, 64 KiB per CU) stay in flight behind a counted vmcnt(8).
constexpr int RING16 = 10;          // units
constexpr int DBL_BYTES = 2 * UNIT_BYTES;

template <bool F16>
__device__ __forceinline__ f32x4 mfma32(s16x8 a, s16x8 b, f32x4 c) {
    if constexpr (F16)
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool F16, bool DOT = false>
__global__ __launch_bounds__(NTHREADS, 2) void xtx16_kernel(XtxParams p) {
    constexpr int ND = 5;       // doubles resident in LDS
    constexpr int LEAD = 3;     // double d+LEAD is issued in phase d  (LEAD <= ND-2)
    __shared__ __attribute__((aligned(16))) char ring[RING16 * UNIT_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;
    const bool group_b = wave >= 4;

    int logical;
    {
        const int b = blockIdx.x, nwg = gridDim.x;
        if (p.map_mode == 0) {
            const int round = b >> 8, rb = b & 255;
            const int m = min(256, nwg - (round << 8));
            const int xcd = rb & 7, idx = rb >> 3, q8 = m >> 3, r8 = m & 7;
            logical = (round << 8) + (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
        } else {
            logical = b;
        }
    }
    // DOT: several problems (one X each, same K / n_tokens / H) in one launch; item index inside the problem
    const int item_global = logical;
    const char* Xb = (const char*)p.X;
    if (DOT) {
        const int prob = logical / p.batch_items;
        logical -= prob * p.batch_items;
        Xb += (size_t)prob * (size_t)p.x_batch_bytes;
    }
    int tile_idx, tt0, cnt, slab_idx = -1;
    if (logical < p.n_direct) {
        tile_idx = logical;
        tt0 = 0;
        cnt = p.n_tt;
    } else {
        const int l2 = logical - p.n_direct;
        const int chunk = l2 / p.n_rem;
        tile_idx = p.n_direct + (l2 - chunk * p.n_rem);
        const int base_cnt = p.n_tt / p.s2, rem = p.n_tt % p.s2;
        tt0 = chunk * base_cnt + (chunk < rem ? chunk : rem);
        cnt = base_cnt + (chunk < rem ? 1 : 0);
        slab_idx = l2;
    }
    const int tt_packed = p.tile_tab[tile_idx];
    const int ti = tt_packed >> 16, tj = tt_packed & 0xFFFF;
    const int nd = cnt * (BKT / (2 * UT));  // doubles of this item (2 per 64-token tile)
    const int K = p.K;
    const size_t ld2 = (size_t)p.ldx * 2;

    // staging: as xtx_kernel, with token-row bit 3 folded into the chunk swizzle
    const int rsub = lane >> 4;
    const int row_hi = (wave & 3) >> 1;                                  // bit 3 of the token row inside the unit
    const int lch = (lane & 15) ^ (rsub << 2) ^ (row_hi << 1);
    const int trow = 4 * (wave & 3) + rsub;
    int colA = ti * BT + (wave >> 2) * 128 + lch * 8;
    int colB = tj * BT + (wave >> 2) * 128 + lch * 8;
    colA = colA > K - 8 ? K - 8 : colA;
    colB = colB > K - 8 ? K - 8 : colB;
    const unsigned voffA = (unsigned)((size_t)trow * ld2 + (size_t)colA * 2);
    const unsigned voffB = (unsigned)((size_t)trow * ld2 + (size_t)colB * 2);
    const unsigned ring_lds = (unsigned)(size_t)(QT_LDS char*)ring;
    const unsigned dst_wave = __builtin_amdgcn_readfirstlane(ring_lds + (wave >> 2) * 4096 + (wave & 3) * 1024);

    const bool ends_in_tail = p.has_tail && (tt0 + cnt == p.n_tt);
    const int d_tail = ends_in_tail ? nd - 2 : 0x7fffffff;             // first double taken from the staging
    const size_t ustride = (size_t)UT * ld2;
    auto dbl_src = [&](int d) -> const char* {
        return d >= d_tail ? (const char*)p.tail + (size_t)(d - d_tail) * 2 * ustride
                           : Xb + ((size_t)tt0 * BKT + (size_t)d * 2 * UT) * ld2;
    };
    const char* run_src = Xb + (size_t)tt0 * BKT * ld2;
    auto issue_at = [&](const char* src, int slot) {
        const unsigned d0 = dst_wave + (unsigned)slot * DBL_BYTES;
        glds16_pair(voffA, voffB, src, d0, d0 + 8192);
        glds16_pair(voffA, voffB, src + ustride, d0 + UNIT_BYTES, d0 + UNIT_BYTES + 8192);
    };

    // fragment geometry: lane group g = lane >> 4 takes tokens 8g..8g+7 of the double
    const int g = lane >> 4, il = lane & 15, q = il >> 2, pp = il & 3;
    const int swz = ((q << 1) ^ (g & 1)) & 7;                            // XOR on the 16-channel block index
    const int lane_base = (g >> 1) * UNIT_BYTES + (8 * (g & 1) + q) * 256 + 16 * (pp >> 1) + 8 * (pp & 1);
    int aoff[8], boff[4];
#pragma unroll
    for (int ai = 0; ai < 8; ++ai) aoff[ai] = wave_m * 4096 + lane_base + 32 * (ai ^ swz);
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
        boff[bj] = (2 + (wave_n >> 1)) * 4096 + lane_base + 32 * ((((wave_n & 1) << 2) + bj) ^ swz);

    f32x4 acc[8][4];
#pragma unroll
    for (int ai = 0; ai < 8; ++ai)
#pragma unroll
        for (int bj = 0; bj < 4; ++bj)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ai][bj][r] = 0.0f;

    s16x8 fa[8], fb[4];
    auto drain_wait = [&](int d) {
        const int later = nd - d - 2;      // doubles after d+1 that exist
        if (later >= 2) wait_vmcnt<8>();
        else if (later == 1) wait_vmcnt<4>();
        else wait_vmcnt<0>();
    };
    auto phase = [&](auto slot_c, auto steady_c, int d) {
        constexpr int S = decltype(slot_c)::value;
        constexpr bool STEADY = decltype(steady_c)::value;
        constexpr int ISLOT = (S + LEAD) % ND;
        const char* base = ring + S * DBL_BYTES;
#pragma unroll
        for (int ai = 0; ai < 8; ++ai) fa[ai] = tr_load8(base + aoff[ai]);
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) fb[bj] = tr_load8(base + boff[bj]);
        if (STEADY) {
            issue_at(run_src, ISLOT);
            run_src += 2 * ustride;
            wait_vmcnt<8>();        // everything up to double d+1 has landed; 2 doubles stay in flight
        } else if (d + LEAD < nd) {
            issue_at(dbl_src(d + LEAD), ISLOT);
            wait_vmcnt<8>();
        } else {
            drain_wait(d);
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ai = 0; ai < 8; ++ai)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) acc[ai][bj] = mfma32<F16>(fa[ai], fb[bj], acc[ai][bj]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    if (nd > 0) {
#pragma unroll
        for (int i = 0; i < LEAD; ++i)
            if (i < nd) issue_at(dbl_src(i), i);
        if (nd >= LEAD) wait_vmcnt<4 * (LEAD - 1)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (group_b) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);

        const int steady_end = (nd < d_tail ? nd : d_tail);
        run_src += (size_t)LEAD * 2 * ustride;
        int d = 0;
        for (; d + ND + LEAD <= steady_end; d += ND) {
            phase(std::integral_constant<int, 0>{}, std::true_type{}, d);
            phase(std::integral_constant<int, 1>{}, std::true_type{}, d + 1);
            phase(std::integral_constant<int, 2>{}, std::true_type{}, d + 2);
            phase(std::integral_constant<int, 3>{}, std::true_type{}, d + 3);
            phase(std::integral_constant<int, 4>{}, std::true_type{}, d + 4);
        }
        // d is a multiple of 5 here: the remaining phases take slots 0, 1, 2, ... in turn
        for (; d < nd; d += ND) {
            phase(std::integral_constant<int, 0>{}, std::false_type{}, d);
            if (d + 1 < nd) phase(std::integral_constant<int, 1>{}, std::false_type{}, d + 1);
            if (d + 2 < nd) phase(std::integral_constant<int, 2>{}, std::false_type{}, d + 2);
            if (d + 3 < nd) phase(std::integral_constant<int, 3>{}, std::false_type{}, d + 3);
            if (d + 4 < nd) phase(std::integral_constant<int, 4>{}, std::false_type{}, d + 4);
        }
        if (!group_b) __builtin_amdgcn_s_barrier();
        wait_vmcnt<0>();
    }

    // epilogue: 16x16 tiles, lane = column (B channel), registers = 4 consecutive rows (A channels)
    const int jl = lane & 15, ih = 4 * (lane >> 4);
    if (DOT) {
        // <H, X^T X> is linear in the token chunks, so every item -- whole tile or chunk of one -- contributes
        // <H tile, its accumulators>: no slab, no store of the product.  fp64 partial sums in a fixed order
        // (registers, lanes by xor-shuffle, waves 0..7), one double per item, summed in item order afterwards.
        double part = 0.0;
#pragma unroll
        for (int ai = 0; ai < 8; ++ai)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) {
                const int gj = tj * BT + wave_n * 64 + bj * 16 + jl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = ti * BT + wave_m * 128 + ai * 16 + ih + r;
                    if (gi < K && gj <= gi) {
                        const double v = (double)p.H[(size_t)gi * K + gj] * (double)acc[ai][bj][r];
                        part += (gj < gi) ? 2.0 * v : v;
                    }
                }
            }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
        double* red = (double*)ring;     // the ring is idle: every wave is past its last read
        if (lane == 0) red[wave] = part;
        __syncthreads();
        if (tid == 0) {
            double t = red[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) t += red[w];
            p.dot_partials[item_global] = t;
        }
        return;
    }
    if (slab_idx < 0) {
#pragma unroll
        for (int ai = 0; ai < 8; ++ai)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) {
                const int gj = tj * BT + wave_n * 64 + bj * 16 + jl;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = ti * BT + wave_m * 128 + ai * 16 + ih + r;
                    if (gi < K && gj < K) {
                        float* dst = p.G + (size_t)gi * K + gj;
                        *dst = *dst + acc[ai][bj][r];
                    }
                }
            }
    } else {
        float* slab = p.slabs + (size_t)slab_idx * (size_t)(BT * BT);
#pragma unroll
        for (int ai = 0; ai < 8; ++ai)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i_loc = wave_m * 128 + ai * 16 + ih + r;
                    const int j_loc = wave_n * 64 + bj * 16 + jl;
                    slab[i_loc * BT + j_loc] = acc[ai][bj][r];
                }
    }
}

// G[tile] += sum_s slab[s][tile]  (ascending s; one float4 per thread per step) for the split tiles
__global__ __launch_bounds__(256) void xtx_reduce_kernel(const float* __restrict__ slabs, int n_rem, int n_splits,
                                                         float* __restrict__ G, int K,
                                                         const int* __restrict__ tile_tab) {
    const int tile = blockIdx.x;
    const int tt_packed = tile_tab[tile];
    const int ti = tt_packed >> 16, tj = tt_packed & 0xFFFF;
    const int part = blockIdx.y;  // 16 parts of 16 rows
    const size_t tile_elems = (size_t)BT * BT;
    for (int e = threadIdx.x; e < 16 * (BT / 4); e += blockDim.x) {
        const int i_loc = part * 16 + e / (BT / 4);
        const int j_loc = (e % (BT / 4)) * 4;
        const int gi = ti * BT + i_loc, gj = tj * BT + j_loc;
        if (gi >= K || gj >= K) continue;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int sp = 0; sp < n_splits; ++sp) {
            const f32x4 v = *(const f32x4*)(slabs + ((size_t)sp * n_rem + tile) * tile_elems +
                                            (size_t)i_loc * BT + j_loc);
            s += v;
        }
        float* dst = G + (size_t)gi * K + gj;
        if (gj + 3 < K) {
            f32x4 o = *(f32x4*)dst;
            o += s;
            *(f32x4*)dst = o;
        } else {
            for (int c = 0; c < 4 && gj + c < K; ++c) dst[c] += s[c];
        }
    }
}

struct XtxPlan {
    int n_tiles, n_tt, has_tail;
    int n_direct, n_rem, s2;
    size_t slab_bytes, tail_bytes, tab_bytes, prog_bytes;
};

// The round-1/2 order (QT_XTX_ORDER=0): bands of 16 tile rows, 16x16 macro blocks along a band, 4 (rows) x 8 (cols)
// super-tiles inside a macro block, row-major inside a super-tile.  Off the diagonal 32 consecutive entries = one
// super-tile (12 distinct panels) and 256 = one macro block (32 panels); the ragged diagonal macro blocks shift the
// chunk boundaries, which is where the order above gains.
// Lower-triangular tiles in locality order (QT_XTX_ORDER=1; the first round-3 order):
// blocks of 8 consecutive panels, block rows walked in PAIRS (a, a + 1) column by column -- (a, b) and (a + 1, b)
// share their column panels -- diagonal blocks as row-major triangles, off-diagonal blocks as 4 x 8 super-tiles.
// What it buys: the 32 workgroups of an XCD take 32 consecutive entries, and a staged panel is an L2 hit for all
// but the first of them that wants it.  Distinct panels per 32-entry chunk, summed over the table (tools/
// xtx_tile_order_eval.py): K = 14336 (56 panels) 898 -> 701, i.e. best-case L2 hit 71.9 % -> 78.0 %; K = 28672:
// 3491 -> 2659; per round of 256 entries (what decides the HBM share of the misses) unchanged, 248 -> 255.
// Measured, same box, alternating processes: K = 14336 1337 -> 1362 TFLOP/s (+1.5...2 %), K = 4096 unchanged.
static void xtx_tile_order_pairs(int nt, std::vector<int>& tab) {
    const int m = 8, nb = (nt + m - 1) / m;
    auto blk = [&](int a, int b) {
        const int r0 = a * m, r1 = std::min(nt, r0 + m), c0 = b * m, c1 = std::min(nt, c0 + m);
        if (a == b) {
            for (int ti = r0; ti < r1; ++ti)
                for (int tj = c0; tj <= ti; ++tj) tab.push_back((ti << 16) | tj);
        } else {
            for (int si = r0; si < r1; si += 4)
                for (int ti = si; ti < std::min(si + 4, r1); ++ti)
                    for (int tj = c0; tj < c1; ++tj) tab.push_back((ti << 16) | tj);
        }
    };
    for (int a = 0; a < nb; a += 2) {
        if (a + 1 < nb) {
            for (int b = 0; b <= a + 1; ++b) {
                if (b <= a) blk(a, b);
                blk(a + 1, b);
            }
        } else {
            for (int b = 0; b <= a; ++b) blk(a, b);
        }
    }
}

// The default for K >= 5632 (QT_XTX_ORDER=2 forces it, 1 = the pairs order above, 0 = the round-1/2 order below): the pairs walk
// with every piece a multiple of 32 entries, so that a 32-entry chunk never straddles two super-tiles.  The pairs
// order emits a diagonal block as its 36-tile triangle, which shifts every later 4 x 8 super-tile by 4 entries
// against the 32-entry chunks of an XCD (a chunk then spans two super-tiles: 14 panels on average instead of 12).
// Here a diagonal block gives its first 32 tiles (rows 0..6 and the first 4 of row 7: 8 panels) as one chunk, placed
// right after its block-row pair's off-diagonal blocks; the 4 tiles left of each triangle, a ragged last block row
// (K not a multiple of 2048) and its triangle go to the end of the table, where misalignment hurts nothing after it.
// Distinct panels per 32-entry chunk, summed (tools/xtx_tile_order_eval.py): K = 14336: 701 -> 588 (best-case L2
// hit 78.0 % -> 81.6 %; 588 = 42 super-tiles x 12 + 7 triangles x 8 + 28 left-over tiles); per round of 256 entries
// 255 -> 232; K = 28672: 2659 -> 2352; K = 4096: 53 -> 48.
static void xtx_tile_order_aligned(int nt, std::vector<int>& tab) {
    const int m = 8, nb = (nt + m - 1) / m;
    const bool ragged = nt % m != 0;
    const int nfull = ragged ? nb - 1 : nb;
    std::vector<int> left;
    auto blk = [&](int a, int b) {
        const int r0 = a * m, r1 = std::min(nt, r0 + m), c0 = b * m, c1 = std::min(nt, c0 + m);
        for (int si = r0; si < r1; si += 4)
            for (int ti = si; ti < std::min(si + 4, r1); ++ti)
                for (int tj = c0; tj < c1; ++tj) tab.push_back((ti << 16) | tj);
    };
    auto tri = [&](int a) {
        const int r0 = a * m, r1 = std::min(nt, r0 + m);
        const bool full = r1 - r0 == m;
        int n = 0;
        for (int ti = r0; ti < r1; ++ti)
            for (int tj = r0; tj <= ti; ++tj, ++n) (full && n < 32 ? tab : left).push_back((ti << 16) | tj);
    };
    for (int a = 0; a < nfull; a += 2) {
        if (a + 1 < nfull) {
            for (int b = 0; b < a; ++b) {
                blk(a, b);
                blk(a + 1, b);
            }
            blk(a + 1, a);
            tri(a);
            tri(a + 1);
        } else {
            for (int b = 0; b < a; ++b) blk(a, b);
            tri(a);
        }
    }
    if (ragged) {
        for (int b = 0; b < nb - 1; ++b) blk(nb - 1, b);
        tri(nb - 1);
    }
    tab.insert(tab.end(), left.begin(), left.end());
}

// which walk: QT_XTX_ORDER, read per call (the tests and A/B tools switch it inside one process)
static int xtx_order_mode(int nt) {
    const char* e = getenv("QT_XTX_ORDER");
    const int mode = e ? atoi(e) : (nt * (nt + 1) / 2 >= NUM_CU ? 2 : 1);
    return mode < 0 || mode > 2 ? 2 : mode;
}

void xtx_tile_order(int nt, std::vector<int>& tab) {
    tab.clear();
    // below one full round of tiles every workgroup is a token slice of a tile and the 32-entry chunks wrap around
    // the table at a stride that is not a multiple of 32: alignment buys nothing there (K = 4096: the pairs order
    // measured 0-3 % faster, profiles/r03_xtx_order_ab.txt) -- see xtx_order_mode
    const int mode = xtx_order_mode(nt);
    if (mode == 1) {
        xtx_tile_order_pairs(nt, tab);
        return;
    }
    if (mode == 2) {
        xtx_tile_order_aligned(nt, tab);
        return;
    }
    for (int bi = 0; bi < nt; bi += 16)
        for (int bj = 0; bj <= bi + 15 && bj < nt; bj += 16)
            for (int si = bi; si < bi + 16 && si < nt; si += 4)
                for (int sj = bj; sj < bj + 16 && sj < nt; sj += 8)
                    for (int ti = si; ti < si + 4 && ti < nt; ++ti)
                        for (int tj = sj; tj < sj + 8 && tj <= ti; ++tj) tab.push_back((ti << 16) | tj);
}

XtxPlan xtx_plan(int64_t n_tokens, int K) {
    XtxPlan pl;
    const int nt = (K + BT - 1) / BT;
    pl.n_tiles = nt * (nt + 1) / 2;
    pl.n_tt = (int)((n_tokens + BKT - 1) / BKT);
    pl.has_tail = (n_tokens % BKT) != 0;
    // whole rounds of 256 tiles: one workgroup per tile over all tokens.  The leftover tiles are
    // split over S token chunks so that they, too, fill whole rounds of the 256 CUs.
    const int n_full = pl.n_tiles / NUM_CU * NUM_CU;
    const int rem = pl.n_tiles - n_full;
    int best = 1;
    if (rem > 0) {
        const size_t slab_cap = (size_t)1 << 30;
        double best_cost = 1.0;  // S = 1: one round at full length
        for (int S = 2; S <= 64; ++S) {
            if (pl.n_tt / S < 4) break;
            if ((size_t)S * rem * BT * BT * 4 > slab_cap) break;
            const long wgs = (long)rem * S;
            const double cost = (double)((wgs + NUM_CU - 1) / NUM_CU) / S;  // rounds x (1/S of a full item)
            if (cost < best_cost * 0.97) {
                best_cost = cost;
                best = S;
            }
        }
    }
    pl.s2 = best;
    if (pl.s2 == 1) {
        pl.n_direct = pl.n_tiles;
        pl.n_rem = 0;
    } else {
        pl.n_direct = n_full;
        pl.n_rem = rem;
    }
    pl.slab_bytes = (size_t)pl.s2 * pl.n_rem * BT * BT * 4;
    pl.tail_bytes = pl.has_tail ? qt_align_up((size_t)BKT * K * 2, 256) : 0;
    pl.tab_bytes = qt_align_up((size_t)pl.n_tiles * 4, 256);
    pl.prog_bytes = qt_align_up((size_t)((pl.n_tiles + NUM_CU - 1) / NUM_CU) * 256 * sizeof(unsigned), 256);
    return pl;
}

// Tile tables live in pinned host memory (one per K, for the life of the process), so the per-call
// hipMemcpyAsync into the caller's workspace is a real asynchronous copy, not a staged one.
const int* xtx_host_table(int K, int n_tiles) {
    static std::mutex m;
    static std::map<std::pair<int, int>, int*> tabs;      // one table per (K, walk)
    std::lock_guard<std::mutex> lock(m);
    const std::pair<int, int> key(K, xtx_order_mode((K + BT - 1) / BT));
    auto it = tabs.find(key);
    if (it != tabs.end()) return it->second;
    std::vector<int> v;
    xtx_tile_order((K + BT - 1) / BT, v);
    if ((int)v.size() != n_tiles) return nullptr;
    int* pinned = nullptr;
    if (hipHostMalloc((void**)&pinned, v.size() * sizeof(int), hipHostMallocDefault) != hipSuccess) return nullptr;
    for (size_t i = 0; i < v.size(); ++i) pinned[i] = v[i];
    tabs[key] = pinned;
    return pinned;
}

}  // namespace

// Host-only self-check of the tile table (runs without a GPU): every lower-triangular 256 x 256 tile exactly once.
extern "C" int qt_xtx_tile_table_check(int K, int* n_tiles_out, int* chunk_panels_out, int* round_panels_out) {
    if (K <= 0) return QT_ERR_INVALID;
    const int nt = (K + BT - 1) / BT;
    std::vector<int> tab;
    xtx_tile_order(nt, tab);
    if ((long)tab.size() != (long)nt * (nt + 1) / 2) return -2;
    std::vector<char> seen((size_t)nt * nt, 0);
    for (int e : tab) {
        const int ti = e >> 16, tj = e & 0xFFFF;
        if (ti >= nt || tj > ti) return -3;
        if (seen[(size_t)ti * nt + tj]++) return -4;
    }
    // distinct panels per chunk of `size` consecutive entries, summed (what one XCD / the chip stages together)
    auto panels = [&](int size) {
        int total = 0;
        std::vector<int> mark(nt, -1);
        for (size_t c = 0; c < tab.size(); c += size)
            for (size_t i = c; i < std::min(tab.size(), c + size); ++i)
                for (int x : {tab[i] >> 16, tab[i] & 0xFFFF})
                    if (mark[x] != (int)(c / size)) {
                        mark[x] = (int)(c / size);
                        ++total;
                    }
        return total;
    };
    if (n_tiles_out) *n_tiles_out = (int)tab.size();
    if (chunk_panels_out) *chunk_panels_out = panels(32);
    if (round_panels_out) *round_panels_out = panels(NUM_CU);
    return 0;
}

extern "C" size_t qt_xtx_workspace_bytes(int64_t n_tokens, int K) {
    if (n_tokens <= 0 || K <= 0) return 0;
    XtxPlan pl = xtx_plan(n_tokens, K);
    return pl.slab_bytes + pl.tail_bytes + pl.tab_bytes + pl.prog_bytes + 256;
}

extern "C" int qt_xtx_accumulate(const void* X, int x_dtype, int64_t n_tokens, int K, int64_t ldx, float* G,
                                 void* workspace, size_t workspace_bytes, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(qt_dtype_is16(x_dtype), "qt_xtx_accumulate: x_dtype %d must be QT_BF16 or QT_F16", x_dtype);
    QT_CHECK_ARG(K > 0 && K % 8 == 0, "qt_xtx_accumulate: K=%d must be a positive multiple of 8", K);
    QT_CHECK_ARG(ldx >= K && ldx % 8 == 0, "qt_xtx_accumulate: ldx=%lld must be >= K and a multiple of 8", (long long)ldx);
    QT_CHECK_ARG(n_tokens >= 0, "qt_xtx_accumulate: n_tokens < 0");
    if (n_tokens == 0) return QT_OK;
    QT_CHECK_ARG(X && G, "qt_xtx_accumulate: null pointer");
    QT_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)G & 15) == 0, "qt_xtx_accumulate: X and G must be 16-byte aligned");
    // per-lane source offsets are 32-bit: 16 token rows of one unit must span < 4 GiB
    QT_CHECK_ARG((uint64_t)ldx * 2 * UT + (uint64_t)K * 2 < ((uint64_t)1 << 32), "qt_xtx_accumulate: ldx too large");
    XtxPlan pl = xtx_plan(n_tokens, K);
    const size_t need = pl.slab_bytes + pl.tail_bytes + pl.tab_bytes + pl.prog_bytes + 256;
    if (workspace_bytes < need || !workspace) {
        qt_set_error("qt_xtx_accumulate: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    char* ws = (char*)qt_align_up((size_t)workspace, 256);
    float* slabs = (float*)ws;
    void* tail = pl.has_tail ? (void*)(ws + pl.slab_bytes) : nullptr;
    int* tile_tab = (int*)(ws + pl.slab_bytes + pl.tail_bytes);
    const int* host_tab = xtx_host_table(K, pl.n_tiles);
    if (!host_tab) {
        qt_set_error("qt_xtx_accumulate: could not build the tile table for K=%d", K);
        return QT_ERR_HIP;
    }
    QT_HIP(hipMemcpyAsync(tile_tab, host_tab, (size_t)pl.n_tiles * 4, hipMemcpyHostToDevice, stream));
    // The ragged last token tile is staged zero-padded with pitch K.  The kernel reads every unit
    // with ONE pitch (its per-lane offsets are loop constants), so the staging can stand in for the
    // last tile only when ldx == K; otherwise the call becomes two launches (below).
    XtxParams p;
    p.X = X;
    p.tail = tail;
    p.ldx = ldx;
    p.K = K;
    p.n_tt = pl.n_tt;
    p.has_tail = pl.has_tail;
    p.tile_tab = tile_tab;
    p.n_direct = pl.n_direct;
    p.n_rem = pl.n_rem;
    p.s2 = pl.s2;
    p.slabs = slabs;
    p.G = G;
    {
        const char* e = getenv("QT_XTX_MAP");
        p.map_mode = e ? atoi(e) : 0;
    }
    p.progress = nullptr;
    p.thr_win = 128;
    p.thr_nap = 127;
    p.thr_chk = 32;
    p.H = nullptr;
    p.dot_partials = nullptr;
    p.batch_items = 1;
    p.x_batch_bytes = 0;
    p.wrap_units = 0;
#ifdef QT_XTX_ABLATION   // lab builds only (tools/xtx_wrap_sweep.sh): the shipped library has no way into xtx_kernel<true>
    {
        const char* e = getenv("QT_XTX_ABLATE_WRAP");  // timing-only ablation, see xtx_kernel<true>
        p.wrap_units = e ? atoi(e) : 0;
    }
#endif
    if (pl.has_tail) {
        const int64_t full = n_tokens / BKT * BKT;
        const int64_t tail_rows = n_tokens - full;
        QT_HIP(hipMemsetAsync(tail, 0, (size_t)BKT * K * 2, stream));
        QT_HIP(hipMemcpy2DAsync(tail, (size_t)K * 2, (const char*)X + (size_t)full * ldx * 2, (size_t)ldx * 2,
                                (size_t)K * 2, (size_t)tail_rows, hipMemcpyDeviceToDevice, stream));
    }
    const char* thr_env = getenv("QT_XTX_THROTTLE");
    const bool throttle_on = thr_env ? atoi(thr_env) != 0 : true;
    unsigned* progress_ws = (unsigned*)(ws + pl.slab_bytes + pl.tail_bytes + pl.tab_bytes);
    auto launch = [&](const XtxParams& q, const XtxPlan& ql) -> int {
        const int grid = ql.n_direct + ql.n_rem * ql.s2;
        qt_prof_mark(QT_PROF_XTX, stream);
        // the throttle pays where several rounds of long items stream more than the Infinity Cache holds
        XtxParams qq = q;
        // (measured: K = 14336, 6 rounds, 5.6 GB of X: +3...4 %; K = 8192, 2 rounds: -1 %)
        const bool thr = throttle_on && ql.n_direct >= 4 * NUM_CU && q.n_tt >= 512 &&
                         (size_t)q.n_tt * BKT * (size_t)q.K * 2 > ((size_t)1 << 30);
        if (thr) {
            const size_t rounds = (size_t)(ql.n_direct + NUM_CU - 1) / NUM_CU;
            if (hipMemsetAsync(progress_ws, 0, rounds * 256 * sizeof(unsigned), stream) != hipSuccess) {
                qt_set_error("qt_xtx_accumulate: hipMemsetAsync(progress) failed");
                return QT_ERR_HIP;
            }
            qq.progress = progress_ws;
            static const int win = [] { const char* e = getenv("QT_XTX_THR_WIN"); return e ? atoi(e) : 128; }();
            static const int nap = [] { const char* e = getenv("QT_XTX_THR_NAP"); return e ? atoi(e) : 127; }();
            qq.thr_win = win;
            qq.thr_nap = nap;
            static const int chk = [] { const char* e = getenv("QT_XTX_THR_CHK"); const int v = e ? atoi(e) : 32; return (v == 8 || v == 16 || v == 64 || v == 128 || v == 256) ? v : 32; }();
            qq.thr_chk = chk;
        } else {
            qq.progress = nullptr;
        }
        // MFMA shape: 16x16x32 where the throttle is not in play (+2...3 % at K = 4096 on every box tried;
        // at K = 14336 it ranged +0.5...+6 % unthrottled, and the throttled 32x32x16 ring beat it);
        // QT_XTX_SHAPE=16|32 forces one (same-process A/B: tools/xtx_lab.py)
        const char* she = getenv("QT_XTX_SHAPE");
        const bool shape16 = she ? atoi(she) == 16 : !thr;
        if (shape16 && q.wrap_units == 0) {
            qq.progress = nullptr;
            if (x_dtype == QT_F16) hipLaunchKernelGGL((xtx16_kernel<true>), dim3(grid), dim3(NTHREADS), 0, stream, qq);
            else hipLaunchKernelGGL((xtx16_kernel<false>), dim3(grid), dim3(NTHREADS), 0, stream, qq);
        }
#ifdef QT_XTX_ABLATION
        else if (q.wrap_units > 0) hipLaunchKernelGGL((xtx_kernel<true, false, false>), dim3(grid), dim3(NTHREADS), 0, stream, qq);
#endif
        else if (thr && x_dtype == QT_F16) hipLaunchKernelGGL((xtx_kernel<false, true, true>), dim3(grid), dim3(NTHREADS), 0, stream, qq);
        else if (thr) hipLaunchKernelGGL((xtx_kernel<false, false, true>), dim3(grid), dim3(NTHREADS), 0, stream, qq);
        else if (x_dtype == QT_F16) hipLaunchKernelGGL((xtx_kernel<false, true, false>), dim3(grid), dim3(NTHREADS), 0, stream, qq);
        else hipLaunchKernelGGL((xtx_kernel<false, false, false>), dim3(grid), dim3(NTHREADS), 0, stream, qq);
        qt_prof_mark(QT_PROF_XTX, stream);
        QT_LAUNCH_CHECK();
        if (ql.n_rem > 0) {
            hipLaunchKernelGGL(xtx_reduce_kernel, dim3(ql.n_rem, 16), dim3(256), 0, stream, q.slabs, ql.n_rem, ql.s2,
                               q.G, q.K, q.tile_tab + ql.n_direct);
            QT_LAUNCH_CHECK();
        }
        return QT_OK;
    };
    if (pl.has_tail && ldx != K) {
        // two launches: the full token tiles with pitch ldx, then the 64 staged rows with pitch K
        const int64_t full = n_tokens / BKT * BKT;
        if (full > 0) {
            XtxPlan pf = xtx_plan(full, K);
            // the split plan of the shorter call must fit the slab area sized for the whole call
            if (pf.slab_bytes > pl.slab_bytes) {
                pf.s2 = 1;
                pf.n_direct = pf.n_tiles;
                pf.n_rem = 0;
                pf.slab_bytes = 0;
            }
            XtxParams pfp = p;
            pfp.n_tt = pf.n_tt;
            pfp.has_tail = 0;
            pfp.n_direct = pf.n_direct;
            pfp.n_rem = pf.n_rem;
            pfp.s2 = pf.s2;
            const int rc = launch(pfp, pf);
            if (rc != QT_OK) return rc;
        }
        XtxPlan pt = xtx_plan(BKT, K);
        pt.s2 = 1;
        pt.n_direct = pt.n_tiles;
        pt.n_rem = 0;
        XtxParams ptp = p;
        ptp.X = tail;
        ptp.ldx = K;
        ptp.n_tt = 1;
        ptp.has_tail = 0;
        ptp.n_direct = pt.n_direct;
        ptp.n_rem = 0;
        ptp.s2 = 1;
        return launch(ptp, pt);
    }
    return launch(p, pl);
}

// ---- internal: loss = scale * <H, X^T X>_F without materialising X^T X (awq.hip) -----------------------------
// H: [K, K] fp32 with a valid lower triangle.  Every work item of the usual plan (whole tiles and token chunks
// alike) writes one fp64 partial; sum_partials_kernel adds them in item order.  Needs n_tokens % 64 == 0 and
// ldx == K (the callers' D matrices); returns QT_ERR_UNSUPPORTED otherwise so the caller takes the two-pass form.
namespace {
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partials, int n, double scale,
                                                           float* __restrict__ out, int accumulate) {
    // one block per problem: partials [gridDim.x][n], out [gridDim.x]
    __shared__ double red[256];
    partials += (size_t)blockIdx.x * n;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partials[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float v = (float)(red[0] * scale);
        out[blockIdx.x] = accumulate ? out[blockIdx.x] + v : v;
    }
}
}  // namespace

// The plan of one problem; with several problems in the launch the tiles alone fill the chip, so no tile is cut
// into token chunks (every extra item is another epilogue that reads a 256 KiB tile of H)
static XtxPlan xtx_frobenius_plan(int64_t n_tokens, int K, int n_batch) {
    XtxPlan pl = xtx_plan(n_tokens, K);
    if ((int64_t)pl.n_tiles * n_batch >= 4 * NUM_CU) {
        pl.n_direct = pl.n_tiles;
        pl.n_rem = 0;
        pl.s2 = 1;
    }
    return pl;
}

size_t qt_xtx_frobenius_workspace_bytes(int64_t n_tokens, int K, int n_batch) {
    if (n_tokens <= 0 || K <= 0 || n_batch <= 0) return 0;
    XtxPlan pl = xtx_frobenius_plan(n_tokens, K, n_batch);
    const size_t grid = ((size_t)pl.n_direct + (size_t)pl.n_rem * pl.s2) * n_batch;
    return pl.tab_bytes + qt_align_up(grid * sizeof(double), 256) + 256;
}

int qt_xtx_frobenius(const void* X, int x_dtype, int64_t n_tokens, int K, int64_t ldx, const float* H, double scale,
                     float* loss_out, int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream,
                     int n_batch, int64_t x_batch_stride) {
    if (!qt_dtype_is16(x_dtype) || K <= 0 || K % 8 != 0 || ldx != K || n_tokens <= 0 || n_tokens % BKT != 0 ||
        ((uintptr_t)X & 15) != 0 || (uint64_t)ldx * 2 * UT + (uint64_t)K * 2 >= ((uint64_t)1 << 32) || n_batch <= 0 ||
        (n_batch > 1 && (x_batch_stride * 2) % 16 != 0))
        return QT_ERR_UNSUPPORTED;
    XtxPlan pl = xtx_frobenius_plan(n_tokens, K, n_batch);
    const int items = pl.n_direct + pl.n_rem * pl.s2;
    if ((int64_t)items * n_batch > (int64_t)1 << 30) return QT_ERR_UNSUPPORTED;
    const size_t need = qt_xtx_frobenius_workspace_bytes(n_tokens, K, n_batch);
    if (!workspace || workspace_bytes < need) {
        qt_set_error("qt_xtx_frobenius: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    char* ws = (char*)qt_align_up((size_t)workspace, 256);
    int* tile_tab = (int*)ws;
    double* partials = (double*)(ws + pl.tab_bytes);
    const int* host_tab = xtx_host_table(K, pl.n_tiles);
    if (!host_tab) {
        qt_set_error("qt_xtx_frobenius: could not build the tile table for K=%d", K);
        return QT_ERR_HIP;
    }
    QT_HIP(hipMemcpyAsync(tile_tab, host_tab, (size_t)pl.n_tiles * 4, hipMemcpyHostToDevice, stream));
    XtxParams p;
    p.X = X;
    p.tail = nullptr;
    p.ldx = ldx;
    p.K = K;
    p.n_tt = pl.n_tt;
    p.has_tail = 0;
    p.tile_tab = tile_tab;
    p.n_direct = pl.n_direct;
    p.n_rem = pl.n_rem;
    p.s2 = pl.s2;
    p.slabs = nullptr;
    p.G = nullptr;
    {
        const char* e = getenv("QT_XTX_MAP");
        p.map_mode = e ? atoi(e) : 0;
    }
    p.wrap_units = 0;
    p.progress = nullptr;
    p.thr_win = 128;
    p.thr_nap = 127;
    p.thr_chk = 32;
    p.H = H;
    p.dot_partials = partials;
    p.batch_items = items;
    p.x_batch_bytes = x_batch_stride * 2;
    const int grid = items * n_batch;
    qt_prof_mark(QT_PROF_XTX, stream);
    if (x_dtype == QT_F16) hipLaunchKernelGGL((xtx16_kernel<true, true>), dim3(grid), dim3(NTHREADS), 0, stream, p);
    else hipLaunchKernelGGL((xtx16_kernel<false, true>), dim3(grid), dim3(NTHREADS), 0, stream, p);
    qt_prof_mark(QT_PROF_XTX, stream);
    QT_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_partials_kernel, dim3(n_batch), dim3(256), 0, stream, (const double*)partials, items, scale,
                       loss_out, accumulate);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

// C-ABI face of the fused product (tests; awq.hip calls qt_xtx_frobenius directly)
extern "C" size_t qt_xtx_dot_workspace_bytes(int64_t n_tokens, int K) { return qt_xtx_frobenius_workspace_bytes(n_tokens, K, 1); }

extern "C" int qt_xtx_dot(const void* X, int x_dtype, int64_t n_tokens, int K, int64_t ldx, const float* H, double scale,
                          float* out, int accumulate, void* workspace, size_t workspace_bytes, qt_stream_t stream) {
    QT_CHECK_ARG(X && H && out, "qt_xtx_dot: null pointer");
    const int rc = qt_xtx_frobenius(X, x_dtype, n_tokens, K, ldx, H, scale, out, accumulate, workspace, workspace_bytes,
                                    (hipStream_t)stream, 1, 0);
    if (rc == QT_ERR_UNSUPPORTED)
        qt_set_error("qt_xtx_dot: needs 16-bit X, ldx == K, K %% 8 == 0 and n_tokens %% 64 == 0 (got K=%d ldx=%lld n=%lld)", K,
                     (long long)ldx, (long long)n_tokens);
    return rc;
}
