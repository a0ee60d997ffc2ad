// Shared helpers for the gfx950 kernels and their C-ABI wrappers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>

#include "../../include/quantool_amd.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define QT_LDS __attribute__((address_space(3)))
#define QT_GLOBAL __attribute__((address_space(1)))

void qt_set_error(const char* fmt, ...);
void qt_prof_mark(int kernel_id, hipStream_t stream);  // no-op unless qt_profile_enable(1)

#define QT_CHECK_ARG(cond, ...)         \
    do {                                \
        if (!(cond)) {                  \
            qt_set_error(__VA_ARGS__);  \
            return QT_ERR_INVALID;      \
        }                               \
    } while (0)

#define QT_HIP(expr)                                                              \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            qt_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                         __FILE__, __LINE__);                                     \
            return QT_ERR_HIP;                                                    \
        }                                                                         \
    } while (0)

#define QT_LAUNCH_CHECK()                                                         \
    do {                                                                          \
        hipError_t _e = hipGetLastError();                                        \
        if (_e != hipSuccess) {                                                   \
            qt_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                     \
            return QT_ERR_HIP;                                                    \
        }                                                                         \
    } while (0)

static inline size_t qt_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// xtx.hip, internal: *loss_out (+)= scale * <H, X^T X>_F with the product never stored (the Gram kernel's
// epilogue multiplies its tile with H's).  QT_ERR_UNSUPPORTED when the layout needs the two-pass form.
size_t qt_xtx_frobenius_workspace_bytes(int64_t n_tokens, int K, int n_batch);
// n_batch problems in one launch: X_b = X + b * x_batch_stride elements (same n_tokens, K, H); loss_out[b]
int qt_xtx_frobenius(const void* X, int x_dtype, int64_t n_tokens, int K, int64_t ldx, const float* H, double scale,
                     float* loss_out, int accumulate, void* workspace, size_t workspace_bytes, hipStream_t stream,
                     int n_batch, int64_t x_batch_stride);

// Wave priority of the latency-bound chain kernels (sweep block, panel factor, panel solve): their
// sparse dependent instruction streams lose issue arbitration to co-resident MFMA-dense waves that run
// their clusters at s_setprio 1; raised, a chain step runs closer to its stand-alone time while the
// dense kernels lose almost no issue slots.  QT_CHAIN_PRIO=0..3 (read per call, for A/B runs).
int qt_chain_prio();
__device__ __forceinline__ void qt_set_chain_prio(int prio) {
    if (prio >= 3) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1) __builtin_amdgcn_s_setprio(1);
}

// Per-device, thread-safe "do once": kernel attributes (dynamic LDS size) belong to a device, and
// the entry points may be called from several host threads (one per stream).  Usage:
//   static QtOncePerDevice once;  QT_HIP(once.run([&] { return hipFuncSetAttribute(...); }));
struct QtOncePerDevice {
    std::mutex m;
    uint64_t done = 0;
    template <class F>
    hipError_t run(F&& f) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const uint64_t bit = 1ull << (dev & 63);
        std::lock_guard<std::mutex> lock(m);
        if (done & bit) return hipSuccess;
        e = f();
        if (e == hipSuccess) done |= bit;
        return e;
    }
};

__device__ __forceinline__ float qt_bf16_to_f32(unsigned short h) {
    return __uint_as_float(((unsigned)h) << 16);
}
__device__ __forceinline__ float qt_f16_to_f32(unsigned short h) {
    return (float)__builtin_bit_cast(_Float16, h);   // exact
}
// 16-bit element (bf16 or IEEE half by dtype) -> fp32, exact
__device__ __forceinline__ float qt_h16_to_f32(unsigned short h, int dtype) {
    return dtype == QT_F16 ? qt_f16_to_f32(h) : qt_bf16_to_f32(h);
}
// element idx of a weight / activation matrix stored as fp32, bf16 or fp16 -> fp32 (upstream: .float())
__device__ __forceinline__ float qt_load_w(const void* W, int dtype, size_t idx) {
    if (dtype == QT_F32) return ((const float*)W)[idx];
    return qt_h16_to_f32(((const unsigned short*)W)[idx], dtype);
}
// fp32 -> the tensor's dtype, round to nearest even (upstream: .to(dtype))
__device__ __forceinline__ void qt_store_w(void* out, int dtype, size_t idx, float v) {
    if (dtype == QT_F32) ((float*)out)[idx] = v;
    else if (dtype == QT_F16) ((_Float16*)out)[idx] = (_Float16)v;
    else ((__bf16*)out)[idx] = (__bf16)v;
}
static inline bool qt_dtype_ok(int d) { return d == QT_F32 || d == QT_BF16 || d == QT_F16; }
static inline bool qt_dtype_is16(int d) { return d == QT_BF16 || d == QT_F16; }
static inline size_t qt_dtype_size(int d) { return d == QT_F32 ? 4 : 2; }
