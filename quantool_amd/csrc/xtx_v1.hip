// a7 (round-1 kernel, kept for same-process A/B runs: env QT_XTX_IMPL=0): Gram accumulation
// G += X^T X with a two-stage LDS double buffer (one vmcnt(0) + barrier per 64-token tile).
// The shipped kernel is xtx.hip.
//
// MFMA-bound (arithmetic intensity ~K flop/B).  X is [tokens][channels] bf16, so BOTH MFMA
// operands are "k-strided" (the reduction index is the row index): X tiles are staged row-major
// into LDS by LDS-DMA (global_load_lds, 16 B/lane) with the swizzle on the SOURCE address, and
// read back with the hardware transposing read ds_read_b64_tr_b16.
//
// Work decomposition: lower-triangular 256x256 output tiles x S token chunks.  Every workgroup
// writes its fp32 partial tile to a slab; xtx_reduce_kernel sums the S slabs of a tile in fixed
// order and adds them into G (deterministic; no atomics).
#include <stdlib.h>

#include <map>
#include <vector>

#include "common.h"

namespace {

constexpr int BT = 256;                    // output tile edge (channels)
constexpr int BKT = 64;                    // tokens per K-step
constexpr int NTHREADS = 512;              // 8 waves: 2 (M) x 4 (N), 128x64 outputs per wave
constexpr int OP_BYTES = BKT * BT * 2;     // one operand panel in LDS (32 KiB)
constexpr int STAGE_BYTES = 2 * OP_BYTES;  // A + B; two stages = 128 KiB of LDS
constexpr int NUM_CU = 256;

struct XtxParams {
    const __bf16* X;
    const __bf16* tail;  // zero-padded [64, K] staging of the ragged last token tile (ld = K)
    int64_t ldx;
    int K;
    int n_tt;       // token tiles (of 64) including the tail tile
    int has_tail;
    int n_tiles;    // lower-triangular 256x256 tiles
    int n_splits;   // S
    float* slabs;   // [S][n_tiles][256*256]
    const int* tile_tab;  // [n_tiles] (ti << 16) | tj, in L2-friendly super-tile order
    int map_mode;         // workgroup -> (chunk, tile) mapping (see kernel)
    int tiles_per_xcd;
};

__device__ __forceinline__ bf16x8 tr_load8(const char* lds_addr) {
    // two transposing reads: tokens +0..3 and +4..7 (rows are 256 B apart -> +1024 B)
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS s16x4*)(lds_addr));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((QT_LDS s16x4*)(lds_addr + 1024));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(NTHREADS, 2) void xtx_v1_kernel(XtxParams p) {
    // Two DISTINCT LDS objects (not one array with two halves): hipcc then knows that the
    // ds_reads of one stage cannot alias the in-flight LDS-DMA writes of the other and stops
    // inserting s_waitcnt vmcnt(0) in front of every stage's first read (which serialised the
    // prefetch of tile t+1 with the compute of tile t).
    __shared__ __attribute__((aligned(16))) char stage0[STAGE_BYTES];
    __shared__ __attribute__((aligned(16))) char stage1[STAGE_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave >> 2, wave_n = wave & 3;

    // XCD-aware remap (bijective form): workgroups that share an XCD get consecutive logical
    // ids = consecutive entries of the tile table for one token chunk.  The table walks the
    // lower triangle in 4x8 super-tiles, so the 32 workgroups resident on an XCD touch ~12
    // distinct X panels per K-step instead of ~33: the rest are hits in that XCD's L2.
    const int nwg = gridDim.x, orig = blockIdx.x;
    int chunk, tile;
    if (p.map_mode == 2) {
        // all 8 XCDs walk the token chunks together, each over its own contiguous slice of the
        // tile table: a panel missed by one XCD's L2 is a MALL hit for the other seven
        const int xcd = orig & 7, pos = orig >> 3;
        chunk = pos / p.tiles_per_xcd;
        tile = xcd * p.tiles_per_xcd + (pos - chunk * p.tiles_per_xcd);
        if (tile >= p.n_tiles || chunk >= p.n_splits) return;
    } else if (p.map_mode == 1) {
        chunk = orig / p.n_tiles;
        tile = orig - chunk * p.n_tiles;
    } else {
        const int q8 = nwg >> 3, r8 = nwg & 7, xcd = orig & 7;
        const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
        chunk = logical / p.n_tiles;
        tile = logical - chunk * p.n_tiles;
    }
    const int tt_packed = p.tile_tab[tile];
    const int ti = tt_packed >> 16, tj = tt_packed & 0xFFFF;
    const bool diag = (ti == tj);

    const int base_cnt = p.n_tt / p.n_splits, rem = p.n_tt % p.n_splits;
    const int tt0 = chunk * base_cnt + (chunk < rem ? chunk : rem);
    const int cnt = base_cnt + (chunk < rem ? 1 : 0);

    // ---- staging geometry (per thread) ----
    const int row_lo = tid >> 4;                        // 0..31
    const int pch = tid & 15;                           // physical 16-B chunk in the 256-B row
    const int lch = pch ^ ((row_lo & 3) << 2);          // logical chunk (source-side swizzle)
    const int K = p.K;

    auto stage = [&](char* sbase, int tt) {
        const bool is_tail = p.has_tail && (tt == p.n_tt - 1);
        const __bf16* src = is_tail ? p.tail : p.X + (size_t)tt * BKT * (size_t)p.ldx;
        const size_t ld = is_tail ? (size_t)K : (size_t)p.ldx;
        const int nops = diag ? 1 : 2;
        for (int op = 0; op < nops; ++op) {
            const int c0 = (op == 0 ? ti : tj) * BT;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_lo + 32 * (r & 1);
                int col = c0 + (r >> 1) * 128 + lch * 8;
                col = col > K - 8 ? K - 8 : col;  // edge tiles: clamp (masked at the store)
                const __bf16* g = src + (size_t)row * ld + col;
                char* l = sbase + op * OP_BYTES + r * 8192 + wave * 1024;
                __builtin_amdgcn_global_load_lds((const QT_GLOBAL void*)g, (QT_LDS void*)l, 16, 0, 0);
            }
        }
    };

    // ---- fragment read geometry (per lane) ----
    const int g = lane >> 4, il = lane & 15, q = il >> 2, pp = il & 3;
    const int low = 2 * (g & 1) + (pp >> 1);
    const int rowpart = (8 * (g >> 1) + q) * 256 + 16 * low + 8 * (pp & 1);
    int aoff[4], boff[2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) aoff[mi] = wave_m * 16384 + rowpart + 64 * (mi ^ q);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
        boff[ni] = (wave_n >> 1) * 16384 + rowpart + 64 * ((((wave_n & 1) << 1) + ni) ^ q);

    f32x16 acc[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    // Software-pipelined fragment reads: the 12 transposing reads of k-step ks+1 are issued
    // BEFORE the 8 MFMAs of k-step ks (256 MFMA cycles cover the LDS latency) and waited for with
    // a counted lgkmcnt after them; sched_barrier pins that order against hipcc's scheduler.
    auto compute = [&](const char* abase) {
        const char* bbase = diag ? abase : abase + OP_BYTES;
        bf16x8 a[2][4], b[2][2];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) a[0][mi] = tr_load8(abase + aoff[mi]);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) b[0][ni] = tr_load8(bbase + boff[ni]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < 4) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) a[nxt][mi] = tr_load8(abase + aoff[mi] + (ks + 1) * 4096);
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) b[nxt][ni] = tr_load8(bbase + boff[ni] + (ks + 1) * 4096);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][mi], b[cur][ni], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (cnt > 0) {
        stage(stage0, tt0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int t = 0;
        for (; t + 1 < cnt; t += 2) {
            stage(stage1, tt0 + t + 1);
            compute(stage0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t + 2 < cnt) stage(stage0, tt0 + t + 2);
            compute(stage1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (t < cnt) compute(stage0);
    }

    // ---- epilogue: fp32 partial tile -> slab (row-major 256x256) ----
    float* slab = p.slabs + ((size_t)chunk * p.n_tiles + tile) * (size_t)(BT * BT);
    const int jl = lane & 31, ih = 4 * (lane >> 5);
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

// G[tile] += sum_s slab[s][tile]  (ascending s; one float4 per thread per step)
__global__ __launch_bounds__(256) void xtx_v1_reduce_kernel(const float* __restrict__ slabs, int n_tiles,
                                                         int n_splits, float* __restrict__ G, int K,
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
            const f32x4 v = *(const f32x4*)(slabs + ((size_t)sp * n_tiles + tile) * tile_elems +
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
    int n_tiles, n_tt, has_tail, n_splits;
    size_t slab_bytes, tail_bytes, tab_bytes;
};

// lower-triangular tiles in 4 (rows) x 8 (cols) super-tile order
void xtx_tile_order(int nt, std::vector<int>& tab) {
    tab.clear();
    for (int bi = 0; bi < nt; bi += 4)
        for (int bj = 0; bj <= bi + 3 && bj < nt; bj += 8)
            for (int ti = bi; ti < bi + 4 && ti < nt; ++ti)
                for (int tj = bj; tj < bj + 8 && tj <= ti; ++tj) tab.push_back((ti << 16) | tj);
}

XtxPlan xtx_plan(int64_t n_tokens, int K) {
    XtxPlan pl;
    const int nt = (K + BT - 1) / BT;
    pl.n_tiles = nt * (nt + 1) / 2;
    pl.n_tt = (int)((n_tokens + BKT - 1) / BKT);
    pl.has_tail = (n_tokens % BKT) != 0;
    // choose the token split S that best fills 256 CUs (1 workgroup per CU) in whole rounds
    int best = 1;
    double best_eff = -1.0;
    const size_t slab_cap = (size_t)3 << 30;
    for (int S = 1; S <= 64 && S <= (pl.n_tt > 0 ? pl.n_tt : 1); ++S) {
        if (S > 1 && pl.n_tt / S < 4) break;
        if ((size_t)S * pl.n_tiles * BT * BT * 4 > slab_cap) break;
        const long wgs = (long)pl.n_tiles * S;
        const long rounds = (wgs + NUM_CU - 1) / NUM_CU;
        const double eff = (double)wgs / (double)(rounds * NUM_CU);
        if (eff > best_eff + 0.02) {
            best_eff = eff;
            best = S;
        }
    }
    pl.n_splits = best;
    pl.slab_bytes = (size_t)pl.n_splits * pl.n_tiles * BT * BT * 4;
    pl.tail_bytes = pl.has_tail ? qt_align_up((size_t)BKT * K * 2, 256) : 0;
    pl.tab_bytes = qt_align_up((size_t)pl.n_tiles * 4, 256);
    return pl;
}

}  // namespace

size_t qt_xtx_v1_workspace_bytes(int64_t n_tokens, int K) {
    if (n_tokens <= 0 || K <= 0) return 0;
    XtxPlan pl = xtx_plan(n_tokens, K);
    return pl.slab_bytes + pl.tail_bytes + pl.tab_bytes + 256;
}

int qt_xtx_v1_accumulate(const void* X, int64_t n_tokens, int K, int64_t ldx, float* G,
                                 void* workspace, size_t workspace_bytes, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(K > 0 && K % 8 == 0, "qt_xtx_accumulate(v1): K=%d must be a positive multiple of 8", K);
    QT_CHECK_ARG(ldx >= K && ldx % 8 == 0, "qt_xtx_accumulate(v1): ldx=%lld must be >= K and a multiple of 8", (long long)ldx);
    QT_CHECK_ARG(n_tokens >= 0, "qt_xtx_accumulate(v1): n_tokens < 0");
    if (n_tokens == 0) return QT_OK;
    QT_CHECK_ARG(X && G, "qt_xtx_accumulate(v1): null pointer");
    QT_CHECK_ARG(((uintptr_t)X & 15) == 0 && ((uintptr_t)G & 15) == 0, "qt_xtx_accumulate(v1): X and G must be 16-byte aligned");
    XtxPlan pl = xtx_plan(n_tokens, K);
    const size_t need = pl.slab_bytes + pl.tail_bytes + pl.tab_bytes + 256;
    if (workspace_bytes < need || !workspace) {
        qt_set_error("qt_xtx_accumulate(v1): workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    char* ws = (char*)qt_align_up((size_t)workspace, 256);
    float* slabs = (float*)ws;
    __bf16* tail = pl.has_tail ? (__bf16*)(ws + pl.slab_bytes) : nullptr;
    int* tile_tab = (int*)(ws + pl.slab_bytes + pl.tail_bytes);
    {
        // host copy kept alive for the life of the process (the async copy reads it); std::map
        // nodes do not move, so the reference stays valid after the lock is dropped
        static std::mutex tabs_mutex;
        static std::map<int, std::vector<int>> tabs;
        std::vector<int>* tab_ptr;
        {
            std::lock_guard<std::mutex> lock(tabs_mutex);
            tab_ptr = &tabs[K];
            if (tab_ptr->empty()) xtx_tile_order((K + BT - 1) / BT, *tab_ptr);
        }
        std::vector<int>& tab = *tab_ptr;
        if ((int)tab.size() != pl.n_tiles) {
            qt_set_error("qt_xtx_accumulate(v1): internal tile table size mismatch");
            return QT_ERR_INVALID;
        }
        QT_HIP(hipMemcpyAsync(tile_tab, tab.data(), (size_t)pl.n_tiles * 4, hipMemcpyHostToDevice, stream));
    }
    if (pl.has_tail) {
        const int64_t full = n_tokens / BKT * BKT;
        const int64_t tail_rows = n_tokens - full;
        QT_HIP(hipMemsetAsync(tail, 0, (size_t)BKT * K * 2, stream));
        QT_HIP(hipMemcpy2DAsync(tail, (size_t)K * 2, (const char*)X + (size_t)full * ldx * 2, (size_t)ldx * 2,
                                (size_t)K * 2, (size_t)tail_rows, hipMemcpyDeviceToDevice, stream));
    }
    XtxParams p;
    p.X = (const __bf16*)X;
    p.tail = tail;
    p.ldx = ldx;
    p.K = K;
    p.n_tt = pl.n_tt;
    p.has_tail = pl.has_tail;
    p.n_tiles = pl.n_tiles;
    p.n_splits = pl.n_splits;
    p.slabs = slabs;
    p.tile_tab = tile_tab;
    static const int map_mode = [] {
        const char* e = getenv("QT_XTX_MAP");
        return e ? atoi(e) : 0;
    }();
    p.map_mode = map_mode;
    p.tiles_per_xcd = (pl.n_tiles + 7) / 8;
    const int grid = (map_mode == 2) ? 8 * p.tiles_per_xcd * pl.n_splits : pl.n_tiles * pl.n_splits;
    qt_prof_mark(QT_PROF_XTX, stream);
    hipLaunchKernelGGL(xtx_v1_kernel, dim3(grid), dim3(NTHREADS), 0, stream, p);
    qt_prof_mark(QT_PROF_XTX, stream);
    QT_LAUNCH_CHECK();
    hipLaunchKernelGGL(xtx_v1_reduce_kernel, dim3(pl.n_tiles, 16), dim3(256), 0, stream, slabs, pl.n_tiles,
                       pl.n_splits, G, K, (const int*)tile_tab);
    QT_LAUNCH_CHECK();
    return QT_OK;
}
