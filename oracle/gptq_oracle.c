/*
 * oracle/gptq_oracle.c -- CPU restatement (TEST INFRASTRUCTURE, not product code)
 *
 * PARITY UNPINNED: the arithmetic of this path lives in the un-vendored third-party
 * packages llmcompressor>=0.8.1 / compressed-tensors (reference pyproject.toml:49-51),
 * which are neither in /root/reference nor installed here, and no reference test pins a
 * numeric result.  Every function below restates the published
 * upstream algorithm as recalled in SURVEY.md Appendix A; the reference call sites that
 * delegate to it are src/quantool/methods/llm_compressor/base.py:161 (oneshot) and
 * gptq/gptq.py:86 (GPTQModifier).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Arithmetic contract (what "bit-exact" means for the HIP path):
 *   - every elementwise op is a single IEEE-754 binary32 operation, no contraction
 *     (compile with -ffp-contract=off);
 *   - where upstream calls a BLAS matmul with reduction length k (order unspecified by
 *     torch), the oracle fixes the order as an ascending-k fmaf chain starting from 0;
 *   - round() is round-half-to-even (torch.round), division is IEEE division.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

static inline float clampf(float x, float lo, float hi) {
    /* torch.clamp(x, lo, hi) == min(max(x, lo), hi); NaN propagates */
    if (x != x) return x;
    x = x < lo ? lo : x;
    x = x > hi ? hi : x;
    return x;
}

/* ---- a7: accumulate_hessian [SURVEY A.2] -------------------------------------------
 * G += X^T X over n_tokens rows of bf16 X[n_tokens, K] (row stride ldx elements).
 * Products of two bf16 values are exact in binary32; the sum is carried in binary64 and
 * rounded once, so G is the correctly rounded Gram matrix up to ~1e-16 relative -- the
 * "truth" the fp32 reference matmul (order unspecified) and the HIP kernel are both
 * compared to with a relative tolerance.  Gd is the caller-owned binary64 accumulator
 * [K*K]; only the lower triangle (i >= j) is touched.
 */
void orc_xtx_accumulate_f64(const uint16_t* X, int64_t n_tokens, int K, int64_t ldx, double* Gd) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = 0; i < K; ++i) {
        for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            for (int64_t t = 0; t < n_tokens; ++t) {
                float a = bf16_to_f32(X[t * ldx + i]);
                float b = bf16_to_f32(X[t * ldx + j]);
                s += (double)(a * b); /* exact product */
            }
            Gd[(int64_t)i * K + j] += s;
        }
    }
}

/* ---- a10: minmax observer -> calculate_qparams [SURVEY A.2] ------------------------
 * W[R,K] fp32 row-major; group_size divides K.  symmetric: scale = max(|min|,|max|) /
 * ((qmax-qmin)/2), zp = 0.  asymmetric: scale = (max-min)/(qmax-qmin),
 * zp = clamp(round(qmin - min/scale), qmin, qmax).  scale clamped below at FLT_EPSILON.
 */
void orc_minmax_qparams(const float* W, int R, int K, int group_size, int symmetric, float qmin,
                        float qmax, float* scale, float* zp) {
    int G = K / group_size;
#pragma omp parallel for
    for (int r = 0; r < R; ++r) {
        for (int g = 0; g < G; ++g) {
            const float* w = W + (int64_t)r * K + (int64_t)g * group_size;
            float mn = w[0], mx = w[0];
            for (int c = 1; c < group_size; ++c) {
                mn = w[c] < mn ? w[c] : mn;
                mx = w[c] > mx ? w[c] : mx;
            }
            mn = mn < 0.0f ? mn : 0.0f;
            mx = mx > 0.0f ? mx : 0.0f;
            float s, z;
            if (symmetric) {
                float amax = fabsf(mn) > fabsf(mx) ? fabsf(mn) : fabsf(mx);
                float bit_range = qmax - qmin; /* 15 for int4 */
                s = amax / (bit_range / 2.0f);
                s = s < 1.1920928955078125e-07f ? 1.1920928955078125e-07f : s;
                z = 0.0f;
            } else {
                s = (mx - mn) / (qmax - qmin);
                s = s < 1.1920928955078125e-07f ? 1.1920928955078125e-07f : s;
                z = qmin - mn / s;
                z = clampf(rintf(z), qmin, qmax);
            }
            scale[(int64_t)r * G + g] = s;
            zp[(int64_t)r * G + g] = z;
        }
    }
}

/* fake_quantize [SURVEY A.2]: q = round_half_even(clamp(x/s + zp, qmin, qmax)); dq=(q-zp)*s */
static inline float fake_quant(float w, float s, float z, float qmin, float qmax, float* qint) {
    float x = w / s;
    x = x + z;
    x = clampf(x, qmin, qmax);
    float q = rintf(x);
    *qint = q;
    return (q - z) * s;
}

/* ---- a11: column sweep of quantize_weight [SURVEY A.2] -----------------------------
 * W[R,K] fp32 (modified in place: on return holds the dequantised weights), U[K,K] fp32
 * upper Cholesky factor of H^-1 (only j >= i read), scale/zp[R,G] fp32, g_idx[K] int32
 * (group of each *sweep position*), blocksize (128 upstream).
 * Outputs: Q[R,K] int8 integer levels, loss[R] per-row sum of (w-q)^2/d^2 / 2.
 * Rows are independent, so the row loop is the parallel one.
 */
void orc_gptq_sweep(float* W, int R, int K, const float* U, const float* scale, const float* zp,
                    int G, const int32_t* g_idx, int blocksize, float qmin, float qmax, int8_t* Q,
                    float* loss) {
#pragma omp parallel
    {
        float* err = (float*)malloc(sizeof(float) * (size_t)blocksize);
        float* pbuf = (float*)malloc(sizeof(float) * (size_t)K);
#pragma omp for schedule(dynamic, 4)
        for (int r = 0; r < R; ++r) {
            float* w = W + (int64_t)r * K;
            float row_loss = 0.0f;
            for (int i1 = 0; i1 < K; i1 += blocksize) {
                int i2 = i1 + blocksize < K ? i1 + blocksize : K;
                int cnt = i2 - i1;
                float blk_loss = 0.0f;
                for (int i = 0; i < cnt; ++i) {
                    int c = i1 + i;
                    float d = U[(int64_t)c * K + c];
                    int g = g_idx[c];
                    float qi;
                    float wv = w[c];
                    float q = fake_quant(wv, scale[(int64_t)r * G + g], zp[(int64_t)r * G + g],
                                         qmin, qmax, &qi);
                    Q[(int64_t)r * K + c] = (int8_t)qi;
                    float diff = wv - q;
                    float d2 = d * d;
                    float l = (diff * diff) / d2;
                    blk_loss = blk_loss + l;
                    float e = diff / d;
                    err[i] = e;
                    w[c] = q;
                    /* W1[:, i:] -= err (x) Hinv1[i, i:]  -- K=1 matmul (one rounding) then
                     * subtraction (second rounding) */
                    const float* urow = U + (int64_t)c * K;
                    for (int j = c + 1; j < i2; ++j) {
                        float p = e * urow[j];
                        w[j] = w[j] - p;
                    }
                }
                row_loss = row_loss + blk_loss / 2.0f;
                /* W[:, i2:] -= Err1 @ Hinv[i1:i2, i2:]  -- ascending-k fmaf chain from 0 */
                /* (loop order i-outer keeps each element's chain ascending in i and streams
                 * rows of U) */
                if (i2 < K) {
                    for (int j = i2; j < K; ++j) pbuf[j] = 0.0f;
                    for (int i = 0; i < cnt; ++i) {
                        const float* urow = U + (int64_t)(i1 + i) * K;
                        float e = err[i];
                        for (int j = i2; j < K; ++j) pbuf[j] = fmaf(e, urow[j], pbuf[j]);
                    }
                    for (int j = i2; j < K; ++j) w[j] = w[j] - pbuf[j];
                }
            }
            loss[r] = row_loss;
        }
        free(err);
        free(pbuf);
    }
}

/* ---- a14: pack_to_int32 [SURVEY A.5] -----------------------------------------------
 * u = q + 8 (4-bit offset binary); element j of each group of 8 at bits 4j..4j+3; the
 * *unsigned* array is zero-padded along K to a multiple of 8 (pad nibble = 0, not 8).
 */
void orc_pack_int4(const int8_t* Q, int R, int K, int32_t* packed) {
    int Kw = (K + 7) / 8;
#pragma omp parallel for
    for (int r = 0; r < R; ++r) {
        for (int wv = 0; wv < Kw; ++wv) {
            uint32_t acc = 0;
            for (int j = 0; j < 8; ++j) {
                int c = wv * 8 + j;
                uint32_t u = 0;
                if (c < K) u = (uint32_t)((int)Q[(int64_t)r * K + c] + 8) & 0xFu;
                acc |= u << (4 * j);
            }
            packed[(int64_t)r * Kw + wv] = (int32_t)acc;
        }
    }
}

/* ---- reference-order Cholesky pieces in plain fp32 (small-K cross-check of the LAPACK
 * calls oracle/reference_path.py makes through scipy).  Unblocked, lower, in place. */
int orc_potrf_lower_f32(float* A, int K) {
    for (int j = 0; j < K; ++j) {
        float s = A[(int64_t)j * K + j];
        for (int k = 0; k < j; ++k) s = fmaf(-A[(int64_t)j * K + k], A[(int64_t)j * K + k], s);
        if (!(s > 0.0f)) return j + 1;
        float d = sqrtf(s);
        A[(int64_t)j * K + j] = d;
        for (int i = j + 1; i < K; ++i) {
            float t = A[(int64_t)i * K + j];
            for (int k = 0; k < j; ++k) t = fmaf(-A[(int64_t)i * K + k], A[(int64_t)j * K + k], t);
            A[(int64_t)i * K + j] = t / d;
        }
    }
    return 0;
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
