// a12 / a13 activation statistics: per-channel sum |x| over tokens
// (AWQ's x_mean numerator) and running per-channel min / max (SmoothQuant).  HBM-bound: one
// 16-byte load per lane (8 bf16 / fp16 channels), token chunks across workgroups, then an ordered
// reduction over chunks (deterministic; no atomics).
#include "common.h"

namespace {

constexpr int TPB = 256;
constexpr int CH_PER_WG = TPB * 8;

struct StatsPlan {
    int strips, n_chunks;
    int64_t tokens_per_chunk;
};

StatsPlan stats_plan(int64_t n_tokens, int K) {
    StatsPlan pl;
    pl.strips = (K + CH_PER_WG - 1) / CH_PER_WG;
    int64_t want = 2048 / pl.strips;
    if (want < 1) want = 1;
    int64_t max_chunks = (n_tokens + 15) / 16;
    if (want > max_chunks) want = max_chunks;
    if (want < 1) want = 1;
    pl.tokens_per_chunk = (n_tokens + want - 1) / want;
    pl.n_chunks = (int)((n_tokens + pl.tokens_per_chunk - 1) / pl.tokens_per_chunk);
    return pl;
}

// partial[chunk][stat][K], stat: 0 abs-sum, 1 min, 2 max
__global__ __launch_bounds__(TPB) void act_stats_partial_kernel(const unsigned short* __restrict__ X,
                                                                int64_t n_tokens, int K, int64_t ldx,
                                                                int64_t tokens_per_chunk,
                                                                float* __restrict__ partial, int dtype) {
    const int c0 = (blockIdx.x * TPB + threadIdx.x) * 8;
    const int chunk = blockIdx.y;
    if (c0 >= K) return;
    const int64_t t0 = (int64_t)chunk * tokens_per_chunk;
    int64_t t1 = t0 + tokens_per_chunk;
    if (t1 > n_tokens) t1 = n_tokens;
    float s[8], mn[8], mx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s[e] = 0.0f;
        mn[e] = INFINITY;
        mx[e] = -INFINITY;
    }
    for (int64_t t = t0; t < t1; ++t) {
        const s16x8 v = *(const s16x8*)(X + (size_t)t * ldx + c0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = qt_h16_to_f32((unsigned short)v[e], dtype);
            s[e] = s[e] + fabsf(f);
            mn[e] = fminf(mn[e], f);
            mx[e] = fmaxf(mx[e], f);
        }
    }
    float* p = partial + (size_t)chunk * 3 * K;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        p[c0 + e] = s[e];
        p[K + c0 + e] = mn[e];
        p[2 * K + c0 + e] = mx[e];
    }
}

__global__ __launch_bounds__(256) void act_stats_reduce_kernel(const float* __restrict__ partial, int n_chunks,
                                                               int K, float* __restrict__ abs_sum,
                                                               float* __restrict__ cmin, float* __restrict__ cmax) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float s = 0.0f, mn = INFINITY, mx = -INFINITY;
    for (int c = 0; c < n_chunks; ++c) {
        const float* p = partial + (size_t)c * 3 * K;
        s = s + p[k];
        mn = fminf(mn, p[K + k]);
        mx = fmaxf(mx, p[2 * K + k]);
    }
    if (abs_sum) abs_sum[k] = abs_sum[k] + s;
    if (cmin) cmin[k] = fminf(cmin[k], mn);
    if (cmax) cmax[k] = fmaxf(cmax[k], mx);
}

}  // namespace

extern "C" size_t qt_act_stats_workspace_bytes(int64_t n_tokens, int K) {
    if (n_tokens <= 0 || K <= 0) return 0;
    StatsPlan pl = stats_plan(n_tokens, K);
    return (size_t)pl.n_chunks * 3 * K * 4 + 256;
}

extern "C" int qt_act_stats_accumulate(const void* X, int x_dtype, int64_t n_tokens, int K, int64_t ldx, float* abs_sum,
                                       float* cmin, float* cmax, void* workspace, size_t workspace_bytes,
                                       qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(qt_dtype_is16(x_dtype), "qt_act_stats_accumulate: x_dtype %d must be QT_BF16 or QT_F16", x_dtype);
    QT_CHECK_ARG(K > 0 && K % 8 == 0 && ldx >= K && ldx % 8 == 0, "qt_act_stats_accumulate: K=%d ldx=%lld must be multiples of 8", K, (long long)ldx);
    if (n_tokens <= 0) return QT_OK;
    QT_CHECK_ARG(X && ((uintptr_t)X & 15) == 0, "qt_act_stats_accumulate: X must be non-null, 16-byte aligned");
    const size_t need = qt_act_stats_workspace_bytes(n_tokens, K);
    if (!workspace || workspace_bytes < need) {
        qt_set_error("qt_act_stats_accumulate: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    StatsPlan pl = stats_plan(n_tokens, K);
    float* partial = (float*)qt_align_up((size_t)workspace, 256);
    hipLaunchKernelGGL(act_stats_partial_kernel, dim3(pl.strips, pl.n_chunks), dim3(TPB), 0, stream,
                       (const unsigned short*)X, n_tokens, K, ldx, pl.tokens_per_chunk, partial, x_dtype);
    QT_LAUNCH_CHECK();
    hipLaunchKernelGGL(act_stats_reduce_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, (const float*)partial,
                       pl.n_chunks, K, abs_sum, cmin, cmax);
    QT_LAUNCH_CHECK();
    return QT_OK;
}
