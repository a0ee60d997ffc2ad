// Shared helpers for the gfx950 kernels and their C-ABI wrappers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>

#include "../../include/quantool_amd.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
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
