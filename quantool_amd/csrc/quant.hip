// a10 / a14 and weight plumbing kernels: minmax observer ->
// calculate_qparams, column gather for activation ordering, pack_to_int32, dequantise.
// All HBM-bound single passes; one wave per (row, group) with wavefront shuffles for the
// group-wise min/max reduction.
#include "common.h"

namespace {


__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// grid (G, R), one wave per (row, group)
__global__ __launch_bounds__(64) void qparams_kernel(const void* __restrict__ W, int dtype, int R, int K,
                                                     int64_t ldw, int gs, int symmetric, float qmin, float qmax,
                                                     float* __restrict__ scale, float* __restrict__ zp,
                                                     float* __restrict__ scale_t, float* __restrict__ zp_t) {
    const int g = blockIdx.x, r = blockIdx.y, lane = threadIdx.x;
    const int G = K / gs;
    const size_t base = (size_t)r * ldw + (size_t)g * gs;
    float mn = INFINITY, mx = -INFINITY;
    for (int c = lane; c < gs; c += 64) {
        const float w = qt_load_w(W, dtype, base + c);
        mn = fminf(mn, w);
        mx = fmaxf(mx, w);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane == 0) {
        mn = fminf(mn, 0.0f);
        mx = fmaxf(mx, 0.0f);
        float s, z;
        const float eps = 1.1920928955078125e-07f;
        if (symmetric) {
            const float amax = fmaxf(fabsf(mn), fabsf(mx));
            s = amax / ((qmax - qmin) / 2.0f);
            s = fmaxf(s, eps);
            z = 0.0f;
        } else {
            s = (mx - mn) / (qmax - qmin);
            s = fmaxf(s, eps);
            z = qmin - mn / s;
            z = fminf(fmaxf(rintf(z), qmin), qmax);
        }
        scale[(size_t)r * G + g] = s;
        zp[(size_t)r * G + g] = z;
        if (scale_t) scale_t[(size_t)g * R + r] = s;
        if (zp_t) zp_t[(size_t)g * R + r] = z;
    }
}

// W_f32[r][s] = dead[s] ? 0 : float(W[r][perm[s]])
__global__ __launch_bounds__(256) void gather_f32_kernel(const void* __restrict__ W, int dtype, int R, int K,
                                                         int64_t ldw, const int32_t* __restrict__ perm,
                                                         const uint8_t* __restrict__ dead, float* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = blockIdx.y;
    if (s >= K) return;
    const int src = perm ? perm[s] : s;
    float v = qt_load_w(W, dtype, (size_t)r * ldw + src);
    if (dead && dead[s]) v = 0.0f;
    out[(size_t)r * K + s] = v;
}

// One read of W for the observer AND the sweep's working copy (static / no activation ordering: qparams on the
// ORIGINAL columns, working copy in sweep order): a workgroup takes one row into LDS (in its own dtype), the four waves
// reduce its groups (wave per group, shuffles), then the row goes out permuted, widened to fp32, dead positions zeroed.
// Same values as qparams_kernel + gather_f32_kernel (min / max are order-free, the scale arithmetic is the same code).
template <typename T>
__global__ __launch_bounds__(256) void gather_qparams_kernel(const T* __restrict__ W, int dtype, int R, int K, int64_t ldw,
                                                             const int32_t* __restrict__ perm,
                                                             const uint8_t* __restrict__ dead, int gs, int symmetric,
                                                             float qmin, float qmax, float* __restrict__ out,
                                                             float* __restrict__ scale, float* __restrict__ zp,
                                                             float* __restrict__ scale_t, float* __restrict__ zp_t,
                                                             int64_t ld_t) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* row = (T*)smem;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* src = W + (size_t)r * ldw;
    constexpr int V = 16 / sizeof(T);                    // elements per 16-byte load
    if ((((uintptr_t)src) & 15) == 0 && K % V == 0) {
        for (int c = tid * V; c < K; c += 256 * V) *(f32x4*)(row + c) = *(const f32x4*)(src + c);
    } else {
        for (int c = tid; c < K; c += 256) row[c] = src[c];
    }
    __syncthreads();
    auto val = [&](int c) -> float {
        if constexpr (sizeof(T) == 4) return (float)row[c];
        else return qt_h16_to_f32((unsigned short)row[c], dtype);
    };
    const int G = K / gs;
    for (int g = wave; g < G; g += 4) {
        float mn = INFINITY, mx = -INFINITY;
        for (int c = lane; c < gs; c += 64) {
            const float w = val(g * gs + c);
            mn = fminf(mn, w);
            mx = fmaxf(mx, w);
        }
        mn = wave_min(mn);
        mx = wave_max(mx);
        if (lane == 0) {
            mn = fminf(mn, 0.0f);
            mx = fmaxf(mx, 0.0f);
            float s, z;
            const float eps = 1.1920928955078125e-07f;
            if (symmetric) {
                const float amax = fmaxf(fabsf(mn), fabsf(mx));
                s = amax / ((qmax - qmin) / 2.0f);
                s = fmaxf(s, eps);
                z = 0.0f;
            } else {
                s = (mx - mn) / (qmax - qmin);
                s = fmaxf(s, eps);
                z = qmin - mn / s;
                z = fminf(fmaxf(rintf(z), qmin), qmax);
            }
            scale[(size_t)r * G + g] = s;
            zp[(size_t)r * G + g] = z;
            if (scale_t) scale_t[(size_t)g * ld_t + r] = s;
            if (zp_t) zp_t[(size_t)g * ld_t + r] = z;
        }
    }
    float* dst = out + (size_t)r * K;
    if (K % 4 == 0) {
        for (int s4 = tid * 4; s4 < K; s4 += 1024) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int sp = s4 + e;
                const float x = val(perm ? perm[sp] : sp);
                v[e] = (dead && dead[sp]) ? 0.0f : x;
            }
            *(f32x4*)(dst + s4) = v;
        }
    } else {
        for (int sp = tid; sp < K; sp += 256) {
            const float x = val(perm ? perm[sp] : sp);
            dst[sp] = (dead && dead[sp]) ? 0.0f : x;
        }
    }
}

// packed[r][w] from Qt[K][R]; workgroup = 64 rows x 32 words, transposed through LDS so that
// reads run along rows (contiguous in Qt) and writes along words (contiguous in packed).
__global__ __launch_bounds__(256) void pack_int4_kernel(const int8_t* __restrict__ Qt, int R, int K,
                                                        const int32_t* __restrict__ col_src,
                                                        int32_t* __restrict__ packed) {
    __shared__ uint32_t tile[64][33];
    const int Kw = (K + 7) / 8;
    const int r0 = blockIdx.y * 64, w0 = blockIdx.x * 32;
    const int tid = threadIdx.x;
    for (int e = tid; e < 64 * 32; e += 256) {
        const int rl = e & 63, wl = e >> 6;
        const int r = r0 + rl, w = w0 + wl;
        uint32_t acc = 0;
        if (r < R && w < Kw) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = w * 8 + j;
                if (c < K) {
                    const int sc = col_src ? col_src[c] : c;
                    const uint32_t u = ((uint32_t)((int)Qt[(size_t)sc * R + r] + 8)) & 0xFu;
                    acc |= u << (4 * j);
                }
            }
        }
        tile[rl][wl] = acc;
    }
    __syncthreads();
    for (int e = tid; e < 64 * 32; e += 256) {
        const int wl = e & 31, rl = e >> 5;
        const int r = r0 + rl, w = w0 + wl;
        if (r < R && w < Kw) packed[(size_t)r * Kw + w] = (int32_t)tile[rl][wl];
    }
}

// The same with FOUR rows per lane on the read side (R % 4 == 0, Qt 4-byte aligned): a lane's load is one 32-bit word =
// the levels of rows 4 rq .. 4 rq + 3 at one sweep position, so a wave's load instruction moves 256 contiguous bytes
// instead of 64.  Workgroup = 256 rows x 32 words.
__global__ __launch_bounds__(256) void pack_int4_rows4_kernel(const int8_t* __restrict__ Qt, int R, int K,
                                                              const int32_t* __restrict__ col_src,
                                                              int32_t* __restrict__ packed) {
    __shared__ uint32_t tile[256][33];
    const int Kw = (K + 7) / 8;
    const int r0 = blockIdx.y * 256, w0 = blockIdx.x * 32;
    const int tid = threadIdx.x;
    for (int e = tid; e < 64 * 32; e += 256) {
        const int rq = e & 63, wl = e >> 6;
        const int r = r0 + 4 * rq, w = w0 + wl;
        uint32_t acc[4] = {0, 0, 0, 0};
        if (r < R && w < Kw) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = w * 8 + j;
                if (c < K) {
                    const int sc = col_src ? col_src[c] : c;
                    const uint32_t q4 = *(const uint32_t*)(Qt + (size_t)sc * R + r);   // rows r .. r + 3
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t u = (uint32_t)((int)(int8_t)(q4 >> (8 * i)) + 8) & 0xFu;
                        acc[i] |= u << (4 * j);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) tile[4 * rq + i][wl] = acc[i];
    }
    __syncthreads();
    for (int e = tid; e < 256 * 32; e += 256) {
        const int wl = e & 31, rl = e >> 5;
        const int r = r0 + rl, w = w0 + wl;
        if (r < R && w < Kw) packed[(size_t)r * Kw + w] = (int32_t)tile[rl][wl];
    }
}

// out[r][c] = (q - zp[r][g(c)]) * scale[r][g(c)]; 64x64 tiles transposed through LDS
__global__ __launch_bounds__(256) void dequant_kernel(const int8_t* __restrict__ Qt, int R, int K,
                                                      const int32_t* __restrict__ col_src,
                                                      const float* __restrict__ scale, const float* __restrict__ zp,
                                                      int G, const int32_t* __restrict__ g_of_col,
                                                      void* __restrict__ out, int out_dtype, int64_t ldo) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x;
    for (int e = tid; e < 64 * 64; e += 256) {
        const int rl = e & 63, cl = e >> 6;
        const int r = r0 + rl, c = c0 + cl;
        float v = 0.0f;
        if (r < R && c < K) {
            const int sc = col_src ? col_src[c] : c;
            v = (float)Qt[(size_t)sc * R + r];
        }
        tile[rl][cl] = v;
    }
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
        const int cl = e & 63, rl = e >> 6;
        const int r = r0 + rl, c = c0 + cl;
        if (r < R && c < K) {
            const int g = g_of_col[c];
            const float q = tile[rl][cl];
            const float z = zp[(size_t)r * G + g];
            const float v = (q - z) * scale[(size_t)r * G + g];
            qt_store_w(out, out_dtype, (size_t)r * ldo + c, v);
        }
    }
}

}  // namespace

static inline void qt_range(int num_bits, float* qmin, float* qmax) {
    *qmin = -(float)(1 << (num_bits - 1));
    *qmax = (float)((1 << (num_bits - 1)) - 1);
}

extern "C" int qt_group_minmax_qparams(const void* W, int w_dtype, int R, int K, int64_t ldw, int group_size,
                                       int symmetric, int num_bits, float* scale, float* zp, float* scale_t,
                                       float* zp_t, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(W && scale && zp && R > 0 && K > 0, "qt_group_minmax_qparams: bad arguments");
    QT_CHECK_ARG(qt_dtype_ok(w_dtype), "qt_group_minmax_qparams: dtype %d unsupported", w_dtype);
    QT_CHECK_ARG(num_bits >= 2 && num_bits <= 8, "qt_group_minmax_qparams: num_bits=%d", num_bits);
    const int gs = group_size <= 0 ? K : group_size;
    QT_CHECK_ARG(K % gs == 0, "qt_group_minmax_qparams: K=%d not divisible by group_size=%d", K, gs);
    float qmin, qmax;
    qt_range(num_bits, &qmin, &qmax);
    // gridDim.y is limited to 65535: rows are processed in chunks (row0 offsets every pointer)
    const size_t esz = qt_dtype_size(w_dtype);
    const int G = K / gs;
    for (int row0 = 0; row0 < R; row0 += 32768) {
        const int rows = (R - row0 < 32768) ? R - row0 : 32768;
        hipLaunchKernelGGL(qparams_kernel, dim3(G, rows), dim3(64), 0, stream,
                           (const void*)((const char*)W + (size_t)row0 * ldw * esz), w_dtype, R, K, ldw, gs, symmetric,
                           qmin, qmax, scale + (size_t)row0 * G, zp + (size_t)row0 * G,
                           scale_t ? scale_t + row0 : nullptr, zp_t ? zp_t + row0 : nullptr);
        QT_LAUNCH_CHECK();
    }
    return QT_OK;
}

extern "C" int qt_weight_gather_f32(const void* W, int w_dtype, int R, int K, int64_t ldw, const int32_t* perm,
                                    const uint8_t* dead, float* W_f32, qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(W && W_f32 && R > 0 && K > 0, "qt_weight_gather_f32: bad arguments");
    QT_CHECK_ARG(qt_dtype_ok(w_dtype), "qt_weight_gather_f32: dtype %d unsupported", w_dtype);
    const size_t esz = qt_dtype_size(w_dtype);
    for (int row0 = 0; row0 < R; row0 += 32768) {
        const int rows = (R - row0 < 32768) ? R - row0 : 32768;
        hipLaunchKernelGGL(gather_f32_kernel, dim3((K + 255) / 256, rows), dim3(256), 0, stream,
                           (const void*)((const char*)W + (size_t)row0 * ldw * esz), w_dtype, rows, K, ldw, perm, dead,
                           W_f32 + (size_t)row0 * K);
        QT_LAUNCH_CHECK();
    }
    return QT_OK;
}

extern "C" int qt_weight_gather_qparams(const void* W, int w_dtype, int R, int K, int64_t ldw, const int32_t* perm,
                                        const uint8_t* dead, int group_size, int symmetric, int num_bits, float* W_f32,
                                        float* scale, float* zp, float* scale_t, float* zp_t, int64_t ld_t,
                                        qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(W && W_f32 && scale && zp && R > 0 && K > 0, "qt_weight_gather_qparams: bad arguments");
    QT_CHECK_ARG(qt_dtype_ok(w_dtype), "qt_weight_gather_qparams: dtype %d unsupported", w_dtype);
    QT_CHECK_ARG(num_bits >= 2 && num_bits <= 8, "qt_weight_gather_qparams: num_bits=%d", num_bits);
    const int gs = group_size <= 0 ? K : group_size;
    QT_CHECK_ARG(K % gs == 0, "qt_weight_gather_qparams: K=%d not divisible by group_size=%d", K, gs);
    QT_CHECK_ARG((!scale_t && !zp_t) || ld_t >= R, "qt_weight_gather_qparams: ld_t=%lld < R=%d", (long long)ld_t, R);
    const size_t esz = qt_dtype_size(w_dtype);
    const size_t lds = qt_align_up((size_t)K * esz, 16);
    QT_CHECK_ARG(lds <= 160 * 1024, "qt_weight_gather_qparams: a row of %d elements does not fit the LDS", K);
    QT_CHECK_ARG(((uintptr_t)W_f32 & 15) == 0, "qt_weight_gather_qparams: W_f32 must be 16-byte aligned");
    float qmin, qmax;
    qt_range(num_bits, &qmin, &qmax);
    static QtOncePerDevice attr16, attr32;
    if (esz == 2) {
        QT_HIP(attr16.run([&] {
            return hipFuncSetAttribute((const void*)gather_qparams_kernel<unsigned short>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }));
        hipLaunchKernelGGL(gather_qparams_kernel<unsigned short>, dim3(R), dim3(256), lds, stream, (const unsigned short*)W,
                           w_dtype, R, K, ldw, perm, dead, gs, symmetric, qmin, qmax, W_f32, scale, zp, scale_t, zp_t, ld_t);
    } else {
        QT_HIP(attr32.run([&] {
            return hipFuncSetAttribute((const void*)gather_qparams_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
        }));
        hipLaunchKernelGGL(gather_qparams_kernel<float>, dim3(R), dim3(256), lds, stream, (const float*)W, w_dtype, R, K, ldw,
                           perm, dead, gs, symmetric, qmin, qmax, W_f32, scale, zp, scale_t, zp_t, ld_t);
    }
    QT_LAUNCH_CHECK();
    return QT_OK;
}

extern "C" int qt_pack_int4(const int8_t* Qt, int R, int K, const int32_t* col_src, int32_t* packed,
                            qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(Qt && packed && R > 0 && K > 0, "qt_pack_int4: bad arguments");
    const int Kw = (K + 7) / 8;
    if (R % 4 == 0 && (((uintptr_t)Qt) & 3) == 0 && (R + 255) / 256 <= 65535)
        hipLaunchKernelGGL(pack_int4_rows4_kernel, dim3((Kw + 31) / 32, (R + 255) / 256), dim3(256), 0, stream, Qt, R, K,
                           col_src, packed);
    else
        hipLaunchKernelGGL(pack_int4_kernel, dim3((Kw + 31) / 32, (R + 63) / 64), dim3(256), 0, stream, Qt, R, K, col_src,
                           packed);
    QT_LAUNCH_CHECK();
    return QT_OK;
}

extern "C" int qt_dequantize(const int8_t* Qt, int R, int K, const int32_t* col_src, const float* scale,
                             const float* zp, int G, const int32_t* g_of_col, void* out, int out_dtype, int64_t ldo,
                             qt_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    QT_CHECK_ARG(Qt && scale && zp && g_of_col && out && R > 0 && K > 0 && G > 0, "qt_dequantize: bad arguments");
    QT_CHECK_ARG(qt_dtype_ok(out_dtype), "qt_dequantize: dtype %d unsupported", out_dtype);
    hipLaunchKernelGGL(dequant_kernel, dim3((K + 63) / 64, (R + 63) / 64), dim3(256), 0, stream, Qt, R, K, col_src,
                       scale, zp, G, g_of_col, out, out_dtype, ldo);
    QT_LAUNCH_CHECK();
    return QT_OK;
}
