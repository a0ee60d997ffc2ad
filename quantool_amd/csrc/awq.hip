// a12 / a13: AWQ per-channel scale search and SmoothQuant scales (SURVEY.md 8a rows a12, a13;
// upstream AWQModifier._compute_best_scale / _pseudo_quantize_tensor / _compute_layer_means and
// SmoothQuantModifier._calculate_smoothing_scales, reached through awq.py:81 / smoothquant.py:77).
//
// The search loss  mean((X W^T - X Wq^T)^2)  is evaluated through the Gram matrix the GPTQ path
// already builds:  sum_r d_r (X^T X) d_r^T / (N R) = <X^T X, D^T D>_F / (N R)  with
// D = W - pseudo_quant(W s)/s.  Exact algebra, and both factors are Gram matrices: D^T D comes from
// the same bf16-MFMA xtx_kernel (D rounded to bf16: the loss moves by < 5e-4 relative, two orders
// below the spacing of the grid points), lower triangle only, followed by one HBM pass for the
// Frobenius product -- R K^2 bf16 flops per grid point instead of 2 N R K.
#include "common.h"
#include "sgemm_tn.h"

#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

constexpr int WM_ROWS = 64;  // rows per workgroup in the weight-mean pass

// partial[chunk][k] = sum over the chunk's rows of |w[r][k]| / (group amax + 1e-6); one wave per
// (group, row chunk), each lane owns gs/64 columns of the group (gs <= 512).
__global__ __launch_bounds__(64) void awq_wmean_partial_kernel(const void* __restrict__ W, int dtype, int R, int K,
                                                               int64_t ldw, int gs, float* __restrict__ partial) {
    const int g = blockIdx.x, chunk = blockIdx.y, lane = threadIdx.x;
    const int per = gs / 64;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
    const int r0 = chunk * WM_ROWS, r1 = (r0 + WM_ROWS < R) ? r0 + WM_ROWS : R;
    for (int r = r0; r < r1; ++r) {
        float v[8];
        float m = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = 0.0f;
            if (e < per) {
                v[e] = fabsf(qt_load_w(W, dtype, (size_t)r * ldw + (size_t)g * gs + e * 64 + lane));
                m = fmaxf(m, v[e]);
            }
        }
        m = wave_max(m) + 1e-6f;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < per) acc[e] = acc[e] + v[e] / m;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (e < per) partial[(size_t)chunk * K + (size_t)g * gs + e * 64 + lane] = acc[e];
}

// Long groups (group_size > 512, channel-wise included): the per-lane register arrays above do not
// scale, so the group maxima come first (one wave per (row, group), lanes striding the group) and the
// column sums are then taken one thread per column over a chunk of rows.
__global__ __launch_bounds__(64) void row_group_absmax_kernel(const void* __restrict__ W, int dtype, int R, int K,
                                                              int64_t ldw, int gs, float* __restrict__ amax) {
    const int g = blockIdx.x, r = blockIdx.y, lane = threadIdx.x;
    float m = 0.0f;
    for (int c = lane; c < gs; c += 64) m = fmaxf(m, fabsf(qt_load_w(W, dtype, (size_t)r * ldw + (size_t)g * gs + c)));
    m = wave_max(m);
    if (lane == 0) amax[(size_t)r * (K / gs) + g] = m;
}

__global__ __launch_bounds__(256) void awq_wmean_cols_kernel(const void* __restrict__ W, int dtype, int R, int K,
                                                             int64_t ldw, int gs, const float* __restrict__ amax,
                                                             float* __restrict__ partial) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int chunk = blockIdx.y;
    if (k >= K) return;
    const int G = K / gs, g = k / gs;
    const int r0 = chunk * WM_ROWS, r1 = (r0 + WM_ROWS < R) ? r0 + WM_ROWS : R;
    float acc = 0.0f;
    for (int r = r0; r < r1; ++r) {
        const float m = amax[(size_t)r * G + g] + 1e-6f;
        acc = acc + fabsf(qt_load_w(W, dtype, (size_t)r * ldw + k)) / m;
    }
    partial[(size_t)chunk * K + k] = acc;
}

__global__ __launch_bounds__(256) void chunk_sum_kernel(const float* __restrict__ partial, int n_chunks, int K,
                                                        float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float s = 0.0f;
    for (int c = 0; c < n_chunks; ++c) s = s + partial[(size_t)c * K + k];
    out[k] = out[k] + s;
}

// scales[g][k] for the n_grid ratios g/n_grid (one workgroup per ratio: the ratios are independent, and one workgroup
// walking all twenty took 0.37 ms at K = 14336):
//   s = clamp(x_mean^r / (w_mean^(1-r) + 1e-4), 1e-4);  s /= sqrt(max s * min s);  nan/inf -> 1
__global__ __launch_bounds__(1024) void awq_scales_kernel(const float* __restrict__ x_sum, float inv_tokens,
                                                          const float* __restrict__ w_sum, float inv_rows, int K,
                                                          int n_grid, int duo, float* __restrict__ scales) {
    __shared__ float smax[16], smin[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const int gi = blockIdx.x;
        const float r = (float)gi / (float)n_grid;
        float mx = -INFINITY, mn = INFINITY;
        for (int k = tid; k < K; k += 1024) {
            const float xm = x_sum[k] * inv_tokens, wm = w_sum[k] * inv_rows;
            float s = duo ? powf(xm, r) / (powf(wm, 1.0f - r) + 1e-4f) : powf(xm, r);
            s = fmaxf(s, 1e-4f);
            scales[(size_t)gi * K + k] = s;
            mx = fmaxf(mx, s);
            mn = fminf(mn, s);
        }
        mx = wave_max(mx);
        mn = wave_min(mn);
        if (lane == 0) {
            smax[wave] = mx;
            smin[wave] = mn;
        }
        __syncthreads();
        mx = smax[0];
        mn = smin[0];
        for (int w = 1; w < 16; ++w) {
            mx = fmaxf(mx, smax[w]);
            mn = fminf(mn, smin[w]);
        }
        const float nrm = sqrtf(mx * mn);
        for (int k = tid; k < K; k += 1024) {
            float s = scales[(size_t)gi * K + k] / nrm;
            if (!(fabsf(s) <= 3.4028234e38f)) s = 1.0f;  // inf or nan
            scales[(size_t)gi * K + k] = s;
        }
    }
}

// One wave per (row, group).  DIFF: Db[r][k] = bf16(w - pseudo_quant(w * s_k) / s_k), row-major
// with leading dimension K (the layout qt_xtx_accumulate reads: rows play the role of tokens).
// !DIFF: out[r][k] = pseudo_quant(w * s_k) / s_k in the weight's own dtype, leading dimension ldo
// (the trial weights of one grid point when the search loss needs a forward of the parent module).
template <int DIFF>   // 0: trial weights in the weight dtype; 1: D as bf16; 2: D as fp32 (exact re-scoring)
__global__ __launch_bounds__(64) void awq_pseudo_quant_kernel(const void* __restrict__ W, int dtype, int R, int K,
                                                              int64_t ldw, const float* __restrict__ s, int gs,
                                                              int symmetric, int num_bits, void* __restrict__ out,
                                                              int64_t ldo) {
    const int g = blockIdx.x, r = blockIdx.y, lane = threadIdx.x;
    const int per = (gs + 63) / 64;
    if (per > 8 || (gs & 63)) {
        // long or odd-sized group (channel-wise: the whole row): two passes over the group, the second one out of L2
        float mx = -INFINITY, mn = INFINITY, amax = 0.0f;
        for (int c = lane; c < gs; c += 64) {
            const int k = g * gs + c;
            const float v = qt_load_w(W, dtype, (size_t)r * ldw + k) * s[k];
            mx = fmaxf(mx, v);
            mn = fminf(mn, v);
            amax = fmaxf(amax, fabsf(v));
        }
        mx = wave_max(mx);
        mn = wave_min(mn);
        amax = wave_max(amax);
        for (int c = lane; c < gs; c += 64) {
            const int k = g * gs + c;
            const float wv = qt_load_w(W, dtype, (size_t)r * ldw + k);
            const float v = wv * s[k];
            float q;
            if (symmetric) {
                const float max_int = (float)((1 << (num_bits - 1)) - 1), min_int = -(float)(1 << (num_bits - 1));
                const float sc = fmaxf(amax, 1e-5f) / max_int;
                q = fminf(fmaxf(rintf(v / sc), min_int), max_int) * sc;
            } else {
                const float max_int = (float)((1 << num_bits) - 1);
                const float sc = fmaxf(mx - mn, 1e-5f) / max_int;
                const float z = fminf(fmaxf(-rintf(mn / sc), 0.0f), max_int);
                q = (fminf(fmaxf(rintf(v / sc) + z, 0.0f), max_int) - z) * sc;
            }
            if (DIFF == 1) ((__bf16*)out)[(size_t)r * K + k] = (__bf16)(wv - q / s[k]);
            else if (DIFF == 2) ((float*)out)[(size_t)r * K + k] = wv - q / s[k];
            else qt_store_w(out, dtype, (size_t)r * ldo + k, q / s[k]);
        }
        return;
    }
    float w[8], ws[8];
    float mx = -INFINITY, mn = INFINITY, amax = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        w[e] = ws[e] = 0.0f;
        if (e < per) {
            const int k = g * gs + e * 64 + lane;
            w[e] = qt_load_w(W, dtype, (size_t)r * ldw + k);
            ws[e] = w[e] * s[k];
            mx = fmaxf(mx, ws[e]);
            mn = fminf(mn, ws[e]);
            amax = fmaxf(amax, fabsf(ws[e]));
        }
    }
    mx = wave_max(mx);
    mn = wave_min(mn);
    amax = wave_max(amax);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (e < per) {
            const int k = g * gs + e * 64 + lane;
            float q;
            if (symmetric) {
                const float max_int = (float)((1 << (num_bits - 1)) - 1), min_int = -(float)(1 << (num_bits - 1));
                const float sc = fmaxf(amax, 1e-5f) / max_int;
                q = fminf(fmaxf(rintf(ws[e] / sc), min_int), max_int) * sc;
            } else {
                const float max_int = (float)((1 << num_bits) - 1);
                const float sc = fmaxf(mx - mn, 1e-5f) / max_int;
                const float z = fminf(fmaxf(-rintf(mn / sc), 0.0f), max_int);
                q = (fminf(fmaxf(rintf(ws[e] / sc) + z, 0.0f), max_int) - z) * sc;
            }
            if (DIFF == 1) ((__bf16*)out)[(size_t)r * K + k] = (__bf16)(w[e] - q / s[k]);
            else if (DIFF == 2) ((float*)out)[(size_t)r * K + k] = w[e] - q / s[k];
            else qt_store_w(out, dtype, (size_t)r * ldo + k, q / s[k]);
        }
    }
}

// The common case -- 16-bit weights, group_size 128 -- at 16 bytes per lane: 16 lanes cover one (row, group),
// a 256-thread block covers 16 of them; min / max / amax are reduced with xor-shuffles inside the 16-lane
// group (order-independent), every element goes through exactly the operations of awq_pseudo_quant_kernel.
// The wave-per-group kernel above moves 2 bytes per lane in 128-byte wave accesses (measured 1.1 TB/s over
// the 20 grid points of a Llama-3-8B layer).
template <int DIFF>
__global__ __launch_bounds__(256) void awq_pseudo_quant_g128_kernel(const unsigned short* __restrict__ W, int dtype,
                                                                    int R, int K, int64_t ldw,
                                                                    const float* __restrict__ s, int symmetric,
                                                                    int num_bits, void* __restrict__ out, int64_t ldo,
                                                                    int64_t s_stride, int64_t out_stride) {
    // blockIdx.y: grid point (its scales s + y * s_stride, its output out + y * out_stride elements)
    s += (size_t)blockIdx.y * (size_t)s_stride;
    const size_t out_off = (size_t)blockIdx.y * (size_t)out_stride;
    const int gpr = K >> 7;                                    // groups per row
    const long grp = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    const bool live = grp < (long)R * gpr;                      // whole 16-lane groups are live or not
    const long grpc = live ? grp : 0;
    const int r = (int)(grpc / gpr), g = (int)(grpc - (long)r * gpr);
    const int k0 = g * 128 + sub * 8;
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
    const u16x8 raw = *(const u16x8*)(W + (size_t)r * ldw + k0);
    const f32x4 s0 = *(const f32x4*)(s + k0), s1 = *(const f32x4*)(s + k0 + 4);
    float w[8], ws[8], sk[8];
    float mx = -INFINITY, mn = INFINITY, amax = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sk[e] = e < 4 ? s0[e] : s1[e - 4];
        w[e] = qt_h16_to_f32(raw[e], dtype);
        ws[e] = w[e] * sk[e];
        mx = fmaxf(mx, ws[e]);
        mn = fminf(mn, ws[e]);
        amax = fmaxf(amax, fabsf(ws[e]));
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off));
        mn = fminf(mn, __shfl_xor(mn, off));
        amax = fmaxf(amax, __shfl_xor(amax, off));
    }
    if (!live) return;
    float res[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float q;
        if (symmetric) {
            const float max_int = (float)((1 << (num_bits - 1)) - 1), min_int = -(float)(1 << (num_bits - 1));
            const float sc = fmaxf(amax, 1e-5f) / max_int;
            q = fminf(fmaxf(rintf(ws[e] / sc), min_int), max_int) * sc;
        } else {
            const float max_int = (float)((1 << num_bits) - 1);
            const float sc = fmaxf(mx - mn, 1e-5f) / max_int;
            const float z = fminf(fmaxf(-rintf(mn / sc), 0.0f), max_int);
            q = (fminf(fmaxf(rintf(ws[e] / sc) + z, 0.0f), max_int) - z) * sc;
        }
        res[e] = (DIFF == 0) ? q / sk[e] : w[e] - q / sk[e];
    }
    if (DIFF == 2) {
        float* dst = (float*)out + out_off + (size_t)r * K + k0;
        *(f32x4*)dst = f32x4{res[0], res[1], res[2], res[3]};
        *(f32x4*)(dst + 4) = f32x4{res[4], res[5], res[6], res[7]};
    } else {
        u16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (DIFF == 1 || dtype == QT_BF16) o[e] = __builtin_bit_cast(unsigned short, (__bf16)res[e]);
            else o[e] = __builtin_bit_cast(unsigned short, (_Float16)res[e]);
        }
        unsigned short* dst = (unsigned short*)out + out_off + (DIFF == 1 ? (size_t)r * K : (size_t)r * ldo) + k0;
        *(u16x8*)dst = o;
    }
}

// DIFF as awq_pseudo_quant_kernel; picks the 16-byte-per-lane kernel when the layout allows it
template <int DIFF>
static int launch_pseudo_quant(const void* W, int w_dtype, int R, int K, int64_t ldw, const float* s, int gs,
                               int symmetric, int num_bits, void* out, int64_t ldo, hipStream_t stream) {
    const bool fast = gs == 128 && qt_dtype_is16(w_dtype) && (K & 127) == 0 && (ldw & 7) == 0 &&
                      (((uintptr_t)W | (uintptr_t)out | (uintptr_t)s) & 15) == 0 && (DIFF != 0 || (ldo & 7) == 0);
    if (fast) {
        const long lanes = (long)R * (K >> 7) * 16;
        hipLaunchKernelGGL(awq_pseudo_quant_g128_kernel<DIFF>, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream,
                           (const unsigned short*)W, w_dtype, R, K, ldw, s, symmetric, num_bits, out, ldo, (int64_t)0,
                           (int64_t)0);
        QT_LAUNCH_CHECK();
        return QT_OK;
    }
    const size_t esz = qt_dtype_size(w_dtype);
    const size_t osz = DIFF == 1 ? 2 : (DIFF == 2 ? 4 : esz);
    const int64_t opitch = DIFF == 0 ? ldo : K;
    for (int row0 = 0; row0 < R; row0 += 32768) {   // gridDim.y limit
        const int rows = (R - row0 < 32768) ? R - row0 : 32768;
        hipLaunchKernelGGL(awq_pseudo_quant_kernel<DIFF>, dim3(K / gs, rows), dim3(64), 0, stream,
                           (const void*)((const char*)W + (size_t)row0 * ldw * esz), w_dtype, rows, K, ldw, s, gs,
                           symmetric, num_bits, (void*)((char*)out + (size_t)row0 * opitch * osz), ldo);
        QT_LAUNCH_CHECK();
    }
    return QT_OK;
}

// partial[i] = sum_{j <= i} G[i][j] * C[i][j] * (j < i ? 2 : 1)   (both lower triangles valid)
__global__ __launch_bounds__(256) void sym_dot_rows_kernel(const float* __restrict__ G, const float* __restrict__ C,
                                                           int K, float* __restrict__ partial) {
    __shared__ double red[256];
    const int i = blockIdx.x;
    double acc = 0.0;
    for (int j = threadIdx.x; j <= i; j += 256) {
        const double v = (double)G[(size_t)i * K + j] * (double)C[(size_t)i * K + j];
        acc += (j < i) ? 2.0 * v : v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[i] = (float)red[0];
}

__global__ __launch_bounds__(256) void symmetrize_kernel(float* __restrict__ G, int K) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < K && j > i) G[(size_t)i * K + j] = G[(size_t)j * K + i];
}

__global__ __launch_bounds__(256) void partial_sum_f64_kernel(const float* __restrict__ partial, int n, double scale,
                                                              float* __restrict__ out, int accumulate) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.0f) + (float)(red[0] * scale);
}

// index of the first minimum of n <= 1024 values (the grid search's argmin: upstream keeps the first
// ratio whose loss is strictly smaller, i.e. the first index among ties); NaN never wins
__global__ __launch_bounds__(64) void argmin_first_kernel(const float* __restrict__ v, int n, int32_t* __restrict__ out) {
    float best = INFINITY;
    int bi = 0;
    for (int i = threadIdx.x; i < n; i += 64) {
        const float x = v[i];
        if (x < best) {
            best = x;
            bi = i;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_down(best, off);
        const int oi = __shfl_down(bi, off);
        if (ob < best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    if (threadIdx.x == 0) out[0] = bi;
}

// out[r][k] = W[r][k] * s[k]  (fp32 product rounded to the output dtype, as `weight * scales`)
__global__ __launch_bounds__(256) void scale_columns_kernel(const void* __restrict__ W, int dtype, int R, int K,
                                                            int64_t ldw, const float* __restrict__ s, int divide,
                                                            void* __restrict__ out, int64_t ldo) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (k >= K) return;
    const float w = qt_load_w(W, dtype, (size_t)r * ldw + k);
    const float v = divide ? w / s[k] : w * s[k];
    qt_store_w(out, dtype, (size_t)r * ldo + k, v);
}

// cmax[k] = max(cmax[k], max_r |W[r][k]|): thread per column, rows strided over blockIdx.y chunks
__global__ __launch_bounds__(256) void col_absmax_partial_kernel(const void* __restrict__ W, int dtype, int R, int K,
                                                                 int64_t ldw, int rows_per, float* __restrict__ part) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int r0 = blockIdx.y * rows_per, r1 = (r0 + rows_per < R) ? r0 + rows_per : R;
    float m = 0.0f;
    for (int r = r0; r < r1; ++r) m = fmaxf(m, fabsf(qt_load_w(W, dtype, (size_t)r * ldw + k)));
    part[(size_t)blockIdx.y * K + k] = m;
}
__global__ __launch_bounds__(256) void chunk_max_kernel(const float* __restrict__ part, int n_chunks, int K,
                                                        float* __restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    float m = out[k];
    for (int c = 0; c < n_chunks; ++c) m = fmaxf(m, part[(size_t)c * K + k]);
    out[k] = m;
}

// s = a^alpha / w^(1-alpha), a = cmax - cmin;  where w == 0: s = a
__global__ __launch_bounds__(256) void smooth_scales_kernel(const float* __restrict__ cmin,
                                                            const float* __restrict__ cmax,
                                                            const float* __restrict__ wmax, int K, float alpha,
                                                            float* __restrict__ s) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const float a = cmax[k] - cmin[k], w = wmax[k];
    s[k] = (w > 0.0f) ? powf(a, alpha) / powf(w, 1.0f - alpha) : a;
}

// Qt[k][r] = level of W[r][k] under plain round-to-nearest with the given group parameters
__global__ __launch_bounds__(256) void rtn_kernel(const void* __restrict__ W, int dtype, int R, int K, int64_t ldw,
                                                  const float* __restrict__ scale, const float* __restrict__ zp,
                                                  int G, int gs, float qmin, float qmax, int8_t* __restrict__ Qt) {
    __shared__ int8_t tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int cl = e & 63, rl = e >> 6;
        const int r = r0 + rl, c = c0 + cl;
        int8_t q = 0;
        if (r < R && c < K) {
            const int g = c / gs;
            const float sc = scale[(size_t)r * G + g], z = zp[(size_t)r * G + g];
            float x = qt_load_w(W, dtype, (size_t)r * ldw + c) / sc;
            x = x + z;
            x = fminf(fmaxf(x, qmin), qmax);
            q = (int8_t)rintf(x);
        }
        tile[rl][cl] = q;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
        const int rl = e & 63, cl = e >> 6;
        const int r = r0 + rl, c = c0 + cl;
        if (r < R && c < K) Qt[(size_t)c * R + r] = tile[rl][cl];
    }
}

}  // namespace

extern "C" size_t qt_awq_weight_mean_workspace_bytes(int R, int K) {
    if (R <= 0 || K <= 0) return 0;
    // per-chunk partial sums + (long groups only) one maximum per (row, group); K / 64 bounds the groups
    return (size_t)((R + WM_ROWS - 1) / WM_ROWS) * K * 4 + qt_align_up((size_t)R * ((K + 63) / 64) * 4, 256) + 512;
}

extern "C" int qt_awq_weight_mean_accumulate(const void* W, int w_dtype, int R, int K, int64_t ldw, int group_size,
                                             float* w_sum, void* workspace, size_t workspace_bytes,
                                             qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(W && w_sum && R > 0 && K > 0, "qt_awq_weight_mean_accumulate: bad arguments");
    QT_CHECK_ARG(qt_dtype_ok(w_dtype), "qt_awq_weight_mean_accumulate: dtype");
    const int gs = group_size <= 0 ? K : group_size;
    if (K % gs != 0) {
        qt_set_error("qt_awq_weight_mean_accumulate: group_size %d does not divide K", gs);
        return QT_ERR_UNSUPPORTED;
    }
    const size_t need = qt_awq_weight_mean_workspace_bytes(R, K);
    if (!workspace || workspace_bytes < need) {
        qt_set_error("qt_awq_weight_mean_accumulate: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    float* partial = (float*)qt_align_up((size_t)workspace, 256);
    const int chunks = (R + WM_ROWS - 1) / WM_ROWS;
    if (gs <= 512 && gs % 64 == 0) {
        hipLaunchKernelGGL(awq_wmean_partial_kernel, dim3(K / gs, chunks), dim3(64), 0, stream, W, w_dtype, R, K, ldw,
                           gs, partial);
        QT_LAUNCH_CHECK();
    } else {
        QT_CHECK_ARG(R <= 65535, "qt_awq_weight_mean_accumulate: R=%d > 65535 rows per call with long groups", R);
        float* amax = partial + (size_t)chunks * K;
        hipLaunchKernelGGL(row_group_absmax_kernel, dim3(K / gs, R), dim3(64), 0, stream, W, w_dtype, R, K, ldw, gs,
                           amax);
        QT_LAUNCH_CHECK();
        hipLaunchKernelGGL(awq_wmean_cols_kernel, dim3((K + 255) / 256, chunks), dim3(256), 0, stream, W, w_dtype, R,
                           K, ldw, gs, (const float*)amax, partial);
        QT_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(chunk_sum_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, (const float*)partial, chunks, K,
                       w_sum);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

extern "C" int qt_awq_scales(const float* x_abs_sum, int64_t n_tokens, const float* w_sum, int64_t n_rows, int K,
                             int n_grid, int duo_scaling, float* scales, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(x_abs_sum && w_sum && scales && K > 0 && n_grid > 0 && n_tokens > 0 && n_rows > 0,
                 "qt_awq_scales: bad arguments");
    hipLaunchKernelGGL(awq_scales_kernel, dim3(n_grid), dim3(1024), 0, stream, x_abs_sum, (float)(1.0 / (double)n_tokens),
                       w_sum, (float)(1.0 / (double)n_rows), K, n_grid, duo_scaling, scales);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

extern "C" int qt_symmetrize_lower(float* G, int K, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(G && K > 0, "qt_symmetrize_lower: bad arguments");
    hipLaunchKernelGGL(symmetrize_kernel, dim3((K + 255) / 256, K), dim3(256), 0, stream, G, K);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

extern "C" size_t qt_awq_loss_workspace_bytes(int R, int K) {
    if (R <= 0 || K <= 0) return 0;
    // D [R,K] (bf16 for the fast form, fp32 for the exact one: sized for fp32) + C [K,K] fp32 + per-row
    // partials + the Gram kernel's own workspace
    return qt_align_up((size_t)R * K * 4, 256) + (size_t)K * K * 4 + qt_align_up((size_t)K * 4, 256) +
           qt_xtx_workspace_bytes(R, K) + 1024;
}

extern "C" int qt_awq_loss(const void* W, int w_dtype, int R, int K, int64_t ldw, const float* s, int group_size,
                           int symmetric, int num_bits, const float* Gfull, int64_t n_tokens, int exact,
                           float weight, int accumulate, float* loss_out, void* workspace, size_t workspace_bytes,
                           qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(W && s && Gfull && loss_out && R > 0 && K > 0 && n_tokens > 0, "qt_awq_loss: bad arguments");
    QT_CHECK_ARG(K % 8 == 0, "qt_awq_loss: K=%d must be a multiple of 8", K);
    QT_CHECK_ARG(qt_dtype_ok(w_dtype), "qt_awq_loss: dtype %d unsupported", w_dtype);
    const int gs = group_size <= 0 ? K : group_size;
    if (K % gs != 0) {
        qt_set_error("qt_awq_loss: group_size %d unsupported", gs);
        return QT_ERR_UNSUPPORTED;
    }
    QT_CHECK_ARG(R <= 65535, "qt_awq_loss: R=%d > 65535 rows per call (split the balance layer)", R);
    const size_t need = qt_awq_loss_workspace_bytes(R, K);
    if (!workspace || workspace_bytes < need) {
        qt_set_error("qt_awq_loss: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    char* ws = (char*)qt_align_up((size_t)workspace, 256);
    void* D = (void*)ws;
    float* C = (float*)(ws + qt_align_up((size_t)R * K * 4, 256));
    float* partial = C + (size_t)K * K;
    char* xws = (char*)partial + qt_align_up((size_t)K * 4, 256);
    const size_t xws_bytes = workspace_bytes - (size_t)(xws - (char*)workspace);
    if (!exact) {
        int rcq = launch_pseudo_quant<1>(W, w_dtype, R, K, ldw, s, gs, symmetric, num_bits, D, (int64_t)K, stream);
        if (rcq) return rcq;
        // <X^T X, D^T D> straight from the Gram kernel's accumulators (no D^T D in memory, no second pass);
        // QT_AWQ_FUSED_DOT=0 or an unsuitable layout: materialise D^T D and multiply in a second pass
        static const bool fused = [] { const char* e = getenv("QT_AWQ_FUSED_DOT"); return !(e && atoi(e) == 0); }();
        if (fused && qt_xtx_frobenius_workspace_bytes(R, K, 1) <= xws_bytes) {
            const int rcf = qt_xtx_frobenius(D, QT_BF16, R, K, K, Gfull, (double)weight / ((double)n_tokens * (double)R),
                                             loss_out, accumulate, xws, xws_bytes, stream, 1, 0);
            if (rcf == QT_OK) return QT_OK;
            if (rcf != QT_ERR_UNSUPPORTED) return rcf;
        }
        QT_HIP(hipMemsetAsync(C, 0, (size_t)K * K * 4, stream));
        const int rc = qt_xtx_accumulate(D, QT_BF16, R, K, K, C, xws, xws_bytes, stream_);
        if (rc) return rc;
    } else {
        // D in fp32 and D^T D on the f32 MFMA: one ascending-k fmaf chain per entry, no bf16 rounding of
        // D.  16x the MFMA time of the fast form: used only to break near-ties of the 20-point search.
        int rcq = launch_pseudo_quant<2>(W, w_dtype, R, K, ldw, s, gs, symmetric, num_bits, D, (int64_t)K, stream);
        if (rcq) return rcq;
        SgemmArgs g;
        g.A = (const float*)D; g.lda = K;
        g.B = (const float*)D; g.ldb = K;
        g.Cin = nullptr; g.ldcin = 0;
        g.Cout = C; g.ldcout = K;
        g.M = K; g.N = K; g.kdim = R; g.k_mode = SG_K_FULL; g.mode = SG_MODE_SET;
        const int rc = qt_sgemm_tn(g, stream);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(sym_dot_rows_kernel, dim3(K), dim3(256), 0, stream, Gfull, (const float*)C, K, partial);
    QT_LAUNCH_CHECK();
    hipLaunchKernelGGL(partial_sum_f64_kernel, dim3(1), dim3(256), 0, stream, (const float*)partial, K,
                       (double)weight / ((double)n_tokens * (double)R), loss_out, accumulate);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

// All n_grid fast search losses of one balance Linear in three launches: D for every grid point (one launch,
// [n_grid][R][K] bf16), one Gram launch over all of them with the Frobenius product against X^T X in its
// epilogue (n_grid times the work items of a single D^T D: short matrices -- R rows play the tokens -- fill
// the chip only together), one sum per grid point.  losses[g] (+)= weight * loss_g.  Falls back to n_grid
// qt_awq_loss calls where the layout does not allow it (group size != 128, R % 64, fp32 weights ...).
extern "C" size_t qt_awq_losses_workspace_bytes(int R, int K, int n_grid) {
    if (R <= 0 || K <= 0 || n_grid <= 0) return 0;
    const size_t batched = qt_align_up((size_t)n_grid * R * K * 2, 256) + qt_xtx_frobenius_workspace_bytes(R, K, n_grid) + 1024;
    const size_t single = qt_awq_loss_workspace_bytes(R, K);
    return batched > single ? batched : single;
}

extern "C" int qt_awq_losses(const void* W, int w_dtype, int R, int K, int64_t ldw, const float* scales, int n_grid,
                             int group_size, int symmetric, int num_bits, const float* Gfull, int64_t n_tokens,
                             float weight, int accumulate, float* losses, void* workspace, size_t workspace_bytes,
                             qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(W && scales && Gfull && losses && R > 0 && K > 0 && n_tokens > 0 && n_grid > 0 && n_grid <= 65535,
                 "qt_awq_losses: bad arguments");
    QT_CHECK_ARG(qt_dtype_ok(w_dtype), "qt_awq_losses: dtype %d unsupported", w_dtype);
    const size_t need = qt_awq_losses_workspace_bytes(R, K, n_grid);
    if (!workspace || workspace_bytes < need) {
        qt_set_error("qt_awq_losses: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    const int gs = group_size <= 0 ? K : group_size;
    static const bool fused = [] { const char* e = getenv("QT_AWQ_FUSED_DOT"); return !(e && atoi(e) == 0); }();
    const bool fast = fused && gs == 128 && qt_dtype_is16(w_dtype) && (K & 127) == 0 && (ldw & 7) == 0 && R % 64 == 0 &&
                      (((uintptr_t)W | (uintptr_t)scales) & 15) == 0;
    if (fast) {
        char* ws = (char*)qt_align_up((size_t)workspace, 256);
        unsigned short* D = (unsigned short*)ws;
        char* xws = ws + qt_align_up((size_t)n_grid * R * K * 2, 256);
        const size_t xws_bytes = workspace_bytes - (size_t)(xws - (char*)workspace);
        const long lanes = (long)R * (K >> 7) * 16;
        hipLaunchKernelGGL(awq_pseudo_quant_g128_kernel<1>, dim3((unsigned)((lanes + 255) / 256), n_grid), dim3(256), 0,
                           stream, (const unsigned short*)W, w_dtype, R, K, ldw, scales, symmetric, num_bits, (void*)D,
                           (int64_t)K, (int64_t)K, (int64_t)R * K);
        QT_LAUNCH_CHECK();
        const int rc = qt_xtx_frobenius(D, QT_BF16, R, K, K, Gfull, (double)weight / ((double)n_tokens * (double)R), losses,
                                        accumulate, xws, xws_bytes, stream, n_grid, (int64_t)R * K);
        if (rc != QT_ERR_UNSUPPORTED) return rc;
    }
    for (int g = 0; g < n_grid; ++g) {
        const int rc = qt_awq_loss(W, w_dtype, R, K, ldw, scales + (size_t)g * K, group_size, symmetric, num_bits, Gfull,
                                   n_tokens, 0, weight, accumulate, losses + g, workspace, workspace_bytes, stream_);
        if (rc) return rc;
    }
    return QT_OK;
}

extern "C" int qt_argmin_f32(const float* values, int n, int32_t* index_out, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(values && index_out && n > 0 && n <= 1024, "qt_argmin_f32: bad arguments (n <= 1024)");
    hipLaunchKernelGGL(argmin_first_kernel, dim3(1), dim3(64), 0, stream, values, n, index_out);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

extern "C" int qt_awq_pseudo_quantize(const void* W, int w_dtype, int R, int K, int64_t ldw, const float* s,
                                       int group_size, int symmetric, int num_bits, void* out, int64_t ldo,
                                       qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(W && s && out && R > 0 && K > 0, "qt_awq_pseudo_quantize: bad arguments");
    QT_CHECK_ARG(qt_dtype_ok(w_dtype), "qt_awq_pseudo_quantize: dtype %d unsupported", w_dtype);
    QT_CHECK_ARG(num_bits >= 2 && num_bits <= 8, "qt_awq_pseudo_quantize: num_bits=%d", num_bits);
    const int gs = group_size <= 0 ? K : group_size;
    if (K % gs != 0) {
        qt_set_error("qt_awq_pseudo_quantize: group_size %d unsupported", gs);
        return QT_ERR_UNSUPPORTED;
    }
    return launch_pseudo_quant<0>(W, w_dtype, R, K, ldw, s, gs, symmetric, num_bits, out, ldo, stream);
}

extern "C" int qt_scale_columns(const void* W, int w_dtype, int R, int K, int64_t ldw, const float* s, int divide,
                                void* out, int64_t ldo, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(W && s && out && R > 0 && K > 0, "qt_scale_columns: bad arguments");
    QT_CHECK_ARG(qt_dtype_ok(w_dtype), "qt_scale_columns: dtype");
    const size_t esz = qt_dtype_size(w_dtype);
    for (int row0 = 0; row0 < R; row0 += 32768) {  // gridDim.y <= 65535
        const int rows = (R - row0 < 32768) ? R - row0 : 32768;
        hipLaunchKernelGGL(scale_columns_kernel, dim3((K + 255) / 256, rows), dim3(256), 0, stream,
                           (const void*)((const char*)W + (size_t)row0 * ldw * esz), w_dtype, rows, K, ldw, s, divide,
                           (void*)((char*)out + (size_t)row0 * ldo * esz), ldo);
        QT_LAUNCH_CHECK();
    }
    return QT_OK;
}

extern "C" size_t qt_col_absmax_workspace_bytes(int R, int K) {
    if (R <= 0 || K <= 0) return 0;
    return (size_t)((R + 127) / 128) * K * 4 + 256;
}

extern "C" int qt_col_absmax_accumulate(const void* W, int w_dtype, int R, int K, int64_t ldw, float* wmax,
                                        void* workspace, size_t workspace_bytes, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(W && wmax && R > 0 && K > 0, "qt_col_absmax_accumulate: bad arguments");
    const size_t need = qt_col_absmax_workspace_bytes(R, K);
    if (!workspace || workspace_bytes < need) {
        qt_set_error("qt_col_absmax_accumulate: workspace %zu < required %zu", workspace_bytes, need);
        return QT_ERR_WORKSPACE;
    }
    float* part = (float*)qt_align_up((size_t)workspace, 256);
    const int chunks = (R + 127) / 128;
    hipLaunchKernelGGL(col_absmax_partial_kernel, dim3((K + 255) / 256, chunks), dim3(256), 0, stream, W, w_dtype, R, K,
                       ldw, 128, part);
    QT_LAUNCH_CHECK();
    hipLaunchKernelGGL(chunk_max_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, (const float*)part, chunks, K,
                       wmax);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

extern "C" int qt_smoothquant_scales(const float* cmin, const float* cmax, const float* wmax, int K, float alpha,
                                     float* s, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(cmin && cmax && wmax && s && K > 0, "qt_smoothquant_scales: bad arguments");
    hipLaunchKernelGGL(smooth_scales_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, cmin, cmax, wmax, K, alpha,
                       s);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

extern "C" int qt_rtn_quantize(const void* W, int w_dtype, int R, int K, int64_t ldw, const float* scale,
                               const float* zp, int G, int group_size, int num_bits, int8_t* Qt,
                               qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(W && scale && zp && Qt && R > 0 && K > 0 && G > 0, "qt_rtn_quantize: bad arguments");
    const int gs = group_size <= 0 ? K : group_size;
    QT_CHECK_ARG(K % gs == 0 && K / gs == G, "qt_rtn_quantize: G != K / group_size");
    const float qmin = -(float)(1 << (num_bits - 1)), qmax = (float)((1 << (num_bits - 1)) - 1);
    hipLaunchKernelGGL(rtn_kernel, dim3((K + 63) / 64, (R + 63) / 64), dim3(256), 0, stream, W, w_dtype, R, K, ldw,
                       scale, zp, G, gs, qmin, qmax, Qt);
    QT_LAUNCH_CHECK();
    return QT_OK;
}
