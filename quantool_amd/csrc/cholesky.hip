// a8: U = chol((H + damp I)^-1, upper)   (upstream quantize_weight:
// cholesky -> cholesky_inverse -> cholesky(upper), reached through gptq.py:86 / base.py:161).
//
// One factorisation instead of three.  With A = flat-reverse(Hd) (A[i][j] = Hd[K-1-i][K-1-j],
// built by qt_hessian_prepare):   A = R^T R (R upper)  =>  Hd = (J R^T J)(J R J) = Ut Ut^T with
// Ut = J R^T J upper, so Hd^-1 = (Ut^-1)^T (Ut^-1) and, by uniqueness of the Cholesky factor,
//     U = Ut^-1 = J R^-T J = flat-reverse(Y),   Y = R^-T (lower).
// Cost 2/3 K^3 flops (vs 4/3 K^3) and every block recurrence below is a k-major "TN" product,
// i.e. one fp32-MFMA kernel (sgemm_tn) serves the whole chain:
//   potrf, block row j :  P = A[j, j:] - sum_{p<j} R[p, j]^T R[p, j:]        (sgemm SUB)
//                         R_jj = chol(P_jj), Dinv_j = R_jj^-1                 (panel kernel)
//                         R[j, j+1:] = Dinv_j^T P[:, nb:]                     (sgemm SET)
//   R^-T,  block row i :  T = sum_{p<i} R[p, i]^T Y[p, :i]   (Y lower => skip k < n0)
//                         Y[i, :i] = -Dinv_i^T T ;  Y[i, i] = Dinv_i^T
#include "common.h"
#include "sgemm_tn.h"

namespace {

constexpr int NB = 128;
constexpr int LDP = NB + 1;  // padded LDS leading dimension

// One workgroup: Cholesky (upper) of an n x n block (n <= 128) + its triangular inverse.
//   P      : input block, upper triangle read, row stride ldp
//   Rout   : receives R (upper triangle written; strict lower untouched), stride ldr
//   Dinv   : receives R^-1 as a dense 128x128 row-major block (zeros outside the triangle / n)
//   Ydiag  : receives (R^-1)^T as a dense n x n block (zeros above the diagonal), stride ldy
//   info   : first non-positive pivot (1-based global index) if any; col0 = global offset
__global__ __launch_bounds__(256) void potf2_inv_kernel(const float* __restrict__ P, int64_t ldp, int n,
                                                        float* __restrict__ Rout, int64_t ldr,
                                                        float* __restrict__ Dinv, float* __restrict__ Ydiag,
                                                        int64_t ldy, int32_t* info, int col0) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* a = sm;             // [NB][LDP]  factor
    float* x = sm + NB * LDP;  // [NB][LDP]  inverse
    const int tid = threadIdx.x;

    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        float v = (i == j) ? 1.0f : 0.0f;
        if (i < n && j < n && j >= i) v = P[(size_t)i * ldp + j];
        a[i * LDP + j] = v;
        x[i * LDP + j] = 0.0f;
    }
    __syncthreads();

    const int jcol = tid & (NB - 1), half = tid >> 7;
    for (int c = 0; c < n; ++c) {
        float piv = a[c * LDP + c];
        if (!(piv > 0.0f)) {  // uniform branch (all threads read the same pivot)
            if (tid == 0) atomicCAS(info, 0, col0 + c + 1);
            piv = 1.0f;
        }
        const float d = sqrtf(piv);
        __syncthreads();
        if (tid < NB) {
            if (tid == c) a[c * LDP + c] = d;
            else if (tid > c) a[c * LDP + tid] = a[c * LDP + tid] / d;
        }
        __syncthreads();
        // trailing update of the upper triangle: a[i][j] -= a[c][i] * a[c][j], c < i <= j
        if (jcol > c) {
            const float rcj = a[c * LDP + jcol];
            for (int i = c + 1 + half; i <= jcol; i += 2) a[i * LDP + jcol] -= a[c * LDP + i] * rcj;
        }
        // the loop-top barrier of the next step orders these writes before the pivot read
        __syncthreads();
    }

    // X = R^-1 (upper) by back substitution, one thread per column j.
    if (tid < NB) {
        const int j = tid;
        x[j * LDP + j] = 1.0f / a[j * LDP + j];
        for (int i = NB - 2; i >= 0; --i) {
            if (i < j) {
                float s = 0.0f;
                for (int pidx = i + 1; pidx <= j; ++pidx) s = fmaf(a[i * LDP + pidx], x[pidx * LDP + j], s);
                x[i * LDP + j] = -s / a[i * LDP + i];
            }
        }
    }
    __syncthreads();

    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        const bool in = (i < n && j < n);
        if (in && j >= i) Rout[(size_t)i * ldr + j] = a[i * LDP + j];
        Dinv[e] = (in && j >= i) ? x[i * LDP + j] : 0.0f;
        if (in) Ydiag[(size_t)i * ldy + j] = (j <= i) ? x[j * LDP + i] : 0.0f;
    }
}

// In-place flat reversal of the lower-triangular Y into the upper-triangular U:
// U[i][j] = Y[K-1-i][K-1-j] for j >= i, strict lower triangle of U = 0.
__global__ __launch_bounds__(256) void flat_reverse_lower_to_upper_kernel(float* __restrict__ U, int K) {
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
                                                                 const int32_t* __restrict__ info) {
    if (*info == 0) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < K) U[(size_t)i * K + j] = (i == j) ? 1.0f : 0.0f;
}

}  // namespace

extern "C" size_t qt_cholesky_inverse_upper_workspace_bytes(int K) {
    if (K <= 0) return 0;
    const size_t nb = (K + NB - 1) / NB;
    // P panel [128, K] + T panel [128, K] + Dinv [nb][128*128]
    return 2 * (size_t)NB * K * 4 + nb * NB * NB * 4 + 256;
}

extern "C" int qt_cholesky_inverse_upper(float* A, int K, float* U, int32_t* info, void* workspace,
                                         size_t workspace_bytes, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(K > 0 && A && U && info, "qt_cholesky_inverse_upper: bad arguments");
    const size_t need = qt_cholesky_inverse_upper_workspace_bytes(K);
    if (!workspace || workspace_bytes < need) {
        qt_set_error("qt_cholesky_inverse_upper: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    char* ws = (char*)qt_align_up((size_t)workspace, 256);
    float* P = (float*)ws;
    float* T = P + (size_t)NB * K;
    float* Dinv = T + (size_t)NB * K;
    float* Y = U;
    const int nblk = (K + NB - 1) / NB;
    const size_t panel_lds = 2 * NB * LDP * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        QT_HIP(hipFuncSetAttribute((const void*)potf2_inv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)panel_lds));
        attr_set = true;
    }
    QT_HIP(hipMemsetAsync(info, 0, sizeof(int32_t), stream));

    for (int j = 0; j < nblk; ++j) {
        const int j0 = j * NB, nbj = (K - j0 < NB) ? K - j0 : NB;
        SgemmArgs g;
        g.A = A + j0; g.lda = K;
        g.B = A + j0; g.ldb = K;
        g.Cin = A + (size_t)j0 * K + j0; g.ldcin = K;
        g.Cout = P; g.ldcout = K;
        g.M = nbj; g.N = K - j0; g.kdim = j0; g.k_mode = SG_K_FULL; g.mode = SG_MODE_SUB;
        int rc = qt_sgemm_tn(g, stream);
        if (rc) return rc;
        hipLaunchKernelGGL(potf2_inv_kernel, dim3(1), dim3(256), panel_lds, stream, (const float*)P, (int64_t)K,
                           nbj, A + (size_t)j0 * K + j0, (int64_t)K, Dinv + (size_t)j * NB * NB,
                           Y + (size_t)j0 * K + j0, (int64_t)K, info, j0);
        QT_LAUNCH_CHECK();
        if (j0 + nbj < K) {
            SgemmArgs t;
            t.A = Dinv + (size_t)j * NB * NB; t.lda = NB;
            t.B = P + nbj; t.ldb = K;
            t.Cin = nullptr; t.ldcin = 0;
            t.Cout = A + (size_t)j0 * K + j0 + nbj; t.ldcout = K;
            t.M = nbj; t.N = K - j0 - nbj; t.kdim = nbj; t.k_mode = SG_K_FULL; t.mode = SG_MODE_SET;
            rc = qt_sgemm_tn(t, stream);
            if (rc) return rc;
        }
    }
    for (int i = 1; i < nblk; ++i) {
        const int i0 = i * NB, nbi = (K - i0 < NB) ? K - i0 : NB;
        SgemmArgs g;
        g.A = A + i0; g.lda = K;
        g.B = Y; g.ldb = K;
        g.Cin = nullptr; g.ldcin = 0;
        g.Cout = T; g.ldcout = K;
        g.M = nbi; g.N = i0; g.kdim = i0; g.k_mode = SG_K_FROM_N0; g.mode = SG_MODE_SET;
        int rc = qt_sgemm_tn(g, stream);
        if (rc) return rc;
        SgemmArgs t;
        t.A = Dinv + (size_t)i * NB * NB; t.lda = NB;
        t.B = T; t.ldb = K;
        t.Cin = nullptr; t.ldcin = 0;
        t.Cout = Y + (size_t)i0 * K; t.ldcout = K;
        t.M = nbi; t.N = i0; t.kdim = nbi; t.k_mode = SG_K_FULL; t.mode = SG_MODE_NEG;
        rc = qt_sgemm_tn(t, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(flat_reverse_lower_to_upper_kernel, dim3((K + 255) / 256, K), dim3(256), 0, stream, U, K);
    QT_LAUNCH_CHECK();
    hipLaunchKernelGGL(identity_if_failed_kernel, dim3((K + 255) / 256, K), dim3(256), 0, stream, U, K,
                       (const int32_t*)info);
    QT_LAUNCH_CHECK();
    return QT_OK;
}
