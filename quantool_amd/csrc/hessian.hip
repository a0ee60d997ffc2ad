// a8/a9: Hessian post-processing ahead of the factorisation (upstream
// quantize_weight: H[dead,dead]=1, damp = percdamp*mean(diag H), H += damp I, optional
// activation-order permutation).  HBM-bound single passes.
#include "common.h"

namespace {

// One workgroup.  diag h[s] = c * G[p[s]][p[s]];  dead[s] = (h == 0);  mean over (dead ? 1 : h)
// carried in fp64;  damp = fp32(percdamp) * fp32(mean).  stats[0] = damp.
__global__ __launch_bounds__(1024) void diag_stats_kernel(const float* __restrict__ G, int K, float c,
                                                          float percdamp, const int32_t* __restrict__ perm,
                                                          uint8_t* __restrict__ dead, float* __restrict__ diag_out,
                                                          float* __restrict__ stats) {
    __shared__ double red[1024];
    double s = 0.0;
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const int o = perm ? perm[i] : i;
        const float h = G[(size_t)o * K + o] * c;
        const bool dd = (h == 0.0f);
        dead[i] = dd ? 1 : 0;
        s += dd ? 1.0 : (double)h;
    }
    if (diag_out) {
        for (int i = threadIdx.x; i < K; i += blockDim.x) diag_out[i] = G[(size_t)i * K + i] * c;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mean = (float)(red[0] / (double)K);
        stats[0] = percdamp * mean;
    }
}

// A[i][j] (j >= i) = Hd[a][b], a = K-1-i >= b = K-1-j;  Hd = P^T (c G) P, dead diag -> 1, + damp I.
__global__ __launch_bounds__(256) void build_flipped_kernel(const float* __restrict__ G, int K, float c,
                                                            const int32_t* __restrict__ perm,
                                                            const uint8_t* __restrict__ dead,
                                                            const float* __restrict__ stats, float* __restrict__ A) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= K || j < i) return;
    const int a = K - 1 - i, b = K - 1 - j;
    const int pa = perm ? perm[a] : a, pb = perm ? perm[b] : b;
    const int hi = pa > pb ? pa : pb, lo = pa > pb ? pb : pa;
    float v = G[(size_t)hi * K + lo] * c;
    if (a == b) {
        if (dead[a]) v = 1.0f;
        v = v + stats[0];
    }
    A[(size_t)i * K + j] = v;
}

// The same result in two coalesced passes (K >= 2048).  build_flipped_kernel reads G[max(pa, pb)][min(pa, pb)]: for
// pb > pa that walks DOWN a column of the lower triangle, one 64-byte sector per 4-byte element (K = 14336: 1.45 ms,
// ~0.6 TB/s).  A permuted symmetric gather cannot avoid strided accesses on a triangle, so pass 1 writes the full
// symmetric matrix S (64 x 64 tiles through LDS: the tile and its transpose, both coalesced), and pass 2 takes ONE
// row of S per output row into LDS and gathers the permuted columns from there.  Pure data movement: every value
// goes through the same G * c (and the same diagonal fix-up) as in the one-pass kernel, so A is bit-identical.
__global__ __launch_bounds__(256) void symmetrize_to_kernel(const float* __restrict__ G, int K, float* __restrict__ S) {
    __shared__ float t[64][65];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
    for (int r = ty; r < 64; r += 4) {
        const int i = bi * 64 + r, j = bj * 64 + tx;
        t[r][tx] = (i < K && j < K && j <= i) ? G[(size_t)i * K + j] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int i = bi * 64 + r, j = bj * 64 + tx;
        if (i < K && j < K) {
            // a diagonal tile holds its lower half only: element (i, j) with j > i is the mirror image t[tx][r]
            S[(size_t)i * K + j] = (bi == bj && j > i) ? t[tx][r] : t[r][tx];
        }
        if (bi != bj) {
            const int it = bj * 64 + r, jt = bi * 64 + tx;      // transposed tile: S[bj-rows][bi-cols]
            if (it < K && jt < K) S[(size_t)it * K + jt] = t[tx][r];
        }
    }
}

__global__ __launch_bounds__(256) void build_flipped_rows_kernel(const float* __restrict__ S, int K, float c,
                                                                 const int32_t* __restrict__ perm,
                                                                 const uint8_t* __restrict__ dead,
                                                                 const float* __restrict__ stats, float* __restrict__ A) {
    extern __shared__ __attribute__((aligned(16))) float row[];   // S[pa][0..K)
    const int i = blockIdx.x;
    const int a = K - 1 - i;
    const int pa = perm ? perm[a] : a;
    const float* src = S + (size_t)pa * K;
    for (int q = threadIdx.x * 4; q < K; q += 256 * 4) *(f32x4*)(row + q) = *(const f32x4*)(src + q);   // K % 4 == 0
    __syncthreads();
    for (int j = i + threadIdx.x; j < K; j += 256) {
        const int b = K - 1 - j;
        const int pb = perm ? perm[b] : b;
        float v = row[pb] * c;
        if (a == b) {
            if (dead[a]) v = 1.0f;
            v = v + stats[0];
        }
        A[(size_t)i * K + j] = v;
    }
}

__global__ __launch_bounds__(256) void diag_only_kernel(const float* __restrict__ G, int K, float c,
                                                        float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) out[i] = G[(size_t)i * K + i] * c;
}

// a9: perm = argsort(diag, descending), stable (ties keep ascending index), by rank counting:
// rank(i) = #{j : d[j] > d[i]  or  (d[j] == d[i] and j < i)};  perm[rank(i)] = i, inv[i] = rank(i).
// O(K^2) compares on K <= 32768 values staged through LDS in chunks: ~1 ms at K = 28672, exact and
// deterministic (no sort network, no atomics).  NaN (a non-finite activation reached diag H) orders as
// the largest value with the index tie-break, as torch.argsort(descending=True) places it, so perm
// is a permutation for ANY input and the gathers that index with it stay in bounds.
__global__ __launch_bounds__(256) void argsort_rank_kernel(const float* __restrict__ d, int K,
                                                           int32_t* __restrict__ perm, int32_t* __restrict__ inv) {
    __shared__ float chunk[1024];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float di = (i < K) ? d[i] : 0.0f;
    const bool nan_i = di != di;
    int rank = 0;
    for (int c0 = 0; c0 < K; c0 += 1024) {
        __syncthreads();
        for (int e = threadIdx.x; e < 1024; e += 256) chunk[e] = (c0 + e < K) ? d[c0 + e] : -INFINITY;
        __syncthreads();
        const int lim = (K - c0 < 1024) ? K - c0 : 1024;
        for (int e = 0; e < lim; ++e) {
            const float dj = chunk[e];
            const bool nan_j = dj != dj;
            const bool gt = nan_j ? !nan_i : (dj > di);
            const bool eq = nan_j ? nan_i : (dj == di);
            rank += gt || (eq && (c0 + e) < i);
        }
    }
    if (i < K) {
        perm[rank] = i;
        if (inv) inv[i] = rank;
    }
}

// The same rank counting spread over a 2-d grid: block (bx, by) counts, for its 256 values i, the j's of chunk by
// (1024 values) that sort before them, and adds the partial count into rank[i] (integer atomics: exact, so the
// result does not depend on the order of the adds).  K / 256 workgroups walking all K values one after the other
// (the kernel above: 56 workgroups x 14336 dependent steps at K = 14336) become K^2 / 2^18 short ones.
__global__ __launch_bounds__(256) void argsort_count_kernel(const float* __restrict__ d, int K,
                                                            int32_t* __restrict__ rank) {
    __shared__ float chunk[1024];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int c0 = blockIdx.y * 1024;
    const float di = (i < K) ? d[i] : 0.0f;
    const bool nan_i = di != di;
    for (int e = threadIdx.x; e < 1024; e += 256) chunk[e] = (c0 + e < K) ? d[c0 + e] : -INFINITY;
    __syncthreads();
    const int lim = (K - c0 < 1024) ? K - c0 : 1024;
    int cnt = 0;
    for (int e = 0; e < lim; ++e) {
        const float dj = chunk[e];
        const bool nan_j = dj != dj;
        const bool gt = nan_j ? !nan_i : (dj > di);
        const bool eq = nan_j ? nan_i : (dj == di);
        cnt += gt || (eq && (c0 + e) < i);
    }
    if (i < K && cnt) atomicAdd(rank + i, cnt);
}

__global__ __launch_bounds__(256) void argsort_scatter_kernel(const int32_t* __restrict__ rank, int K,
                                                              int32_t* __restrict__ perm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) perm[rank[i]] = i;
}

}  // namespace

extern "C" int qt_argsort_desc(const float* values, int K, int32_t* perm, int32_t* inv, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(values && perm && K > 0, "qt_argsort_desc: bad arguments");
    if (inv && K > 1024) {
        // inv[i] = rank(i) is the accumulator of the 2-d count
        QT_HIP(hipMemsetAsync(inv, 0, (size_t)K * sizeof(int32_t), stream));
        hipLaunchKernelGGL(argsort_count_kernel, dim3((K + 255) / 256, (K + 1023) / 1024), dim3(256), 0, stream, values, K,
                           inv);
        QT_LAUNCH_CHECK();
        hipLaunchKernelGGL(argsort_scatter_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, (const int32_t*)inv, K,
                           perm);
        QT_LAUNCH_CHECK();
        return QT_OK;
    }
    hipLaunchKernelGGL(argsort_rank_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, values, K, perm, inv);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

// two-pass form: K a multiple of 4, one row of S in LDS (<= 160 KiB); QT_PREPARE_TWO_PASS=0 keeps the one-pass kernel
static bool prepare_two_pass(int K) {
    const char* e = getenv("QT_PREPARE_TWO_PASS");
    if (e && atoi(e) == 0) return false;
    return K >= 2048 && K % 4 == 0 && (size_t)K * 4 <= 160 * 1024;
}

extern "C" size_t qt_hessian_prepare_workspace_bytes(int K) {
    if (K <= 0) return 0;
    return 512 + (prepare_two_pass(K) ? qt_align_up((size_t)K * K * 4, 256) + 256 : 0);
}

extern "C" int qt_hessian_prepare(const float* G, int K, int64_t n_samples, float percdamp, const int32_t* perm,
                                  float* A, uint8_t* dead, float* diag_out, void* workspace, size_t workspace_bytes,
                                  qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(G && A && dead && K > 0 && n_samples > 0, "qt_hessian_prepare: bad arguments");
    const size_t need = qt_hessian_prepare_workspace_bytes(K);
    if (!workspace || workspace_bytes < need) {
        qt_set_error("qt_hessian_prepare: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    float* stats = (float*)qt_align_up((size_t)workspace, 256);
    const float c = (float)(2.0 / (double)n_samples);
    hipLaunchKernelGGL(diag_stats_kernel, dim3(1), dim3(1024), 0, stream, G, K, c, percdamp, perm, dead, diag_out,
                       stats);
    QT_LAUNCH_CHECK();
    if (prepare_two_pass(K) && (((uintptr_t)G) & 15) == 0) {
        float* S = (float*)qt_align_up((size_t)(stats + 64), 256);
        const int nb = (K + 63) / 64;
        hipLaunchKernelGGL(symmetrize_to_kernel, dim3(nb, nb), dim3(256), 0, stream, G, K, S);
        QT_LAUNCH_CHECK();
        const size_t lds = (size_t)K * 4;
        static QtOncePerDevice lds_attr;
        QT_HIP(lds_attr.run([&] {
            return hipFuncSetAttribute((const void*)build_flipped_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
        }));
        hipLaunchKernelGGL(build_flipped_rows_kernel, dim3(K), dim3(256), lds, stream, (const float*)S, K, c, perm,
                           (const uint8_t*)dead, (const float*)stats, A);
        QT_LAUNCH_CHECK();
        return QT_OK;
    }
    hipLaunchKernelGGL(build_flipped_kernel, dim3((K + 255) / 256, K), dim3(256), 0, stream, G, K, c, perm,
                       (const uint8_t*)dead, (const float*)stats, A);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

extern "C" int qt_hessian_diag(const float* G, int K, int64_t n_samples, float* diag_out, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(G && diag_out && K > 0 && n_samples > 0, "qt_hessian_diag: bad arguments");
    const float c = (float)(2.0 / (double)n_samples);
    hipLaunchKernelGGL(diag_only_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, G, K, c, diag_out);
    QT_LAUNCH_CHECK();
    return QT_OK;
}
