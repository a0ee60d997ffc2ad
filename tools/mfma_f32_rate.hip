// Probe: what rate does v_mfma_f32_32x32x2_f32 sustain per SIMD, by waves per SIMD, accumulators per wave, and with
// the operands re-read from LDS in front of every MFMA pair (the f32 update GEMM's inner loop)?
// Build:  hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_rate.hip -o tools/_build/mfma_f32_rate      Run on a GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC, bool LDS>
__global__ __launch_bounds__(1024) void rate(float* out, int iters, float seed) {
    __shared__ float sm[16 * 256];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 256; i += blockDim.x) sm[i] = seed + (float)(i % 17) * 1e-3f;
    __syncthreads();
    f32x16 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float a = seed + lane * 1e-3f, b = seed - lane * 1e-3f;
    const float* As = sm + (lane & 31) + (lane >> 5) * 128;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            if (LDS) {
                a = As[kk * 256];
                b = As[kk * 256 + 32];
            }
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b + (float)j, acc[j], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(int waves_per_simd, float* out) {
    const int threads = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;
    const int blocks_per_cu = (64 * 4 * waves_per_simd) / threads;
    const int grid = 256 * blocks_per_cu;
    const int iters = 4000 / NACC;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rate<NACC, LDS>), dim3(grid), dim3(threads), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)grid * (threads / 64) * iters * 8 * NACC;
    const double tf = mfmas * 4096.0 / (ms * 1e-3) / 1e12;
    printf("waves/SIMD %d  acc/wave %d  lds %d: %8.3f ms  %7.1f TFLOP/s  (%.1f cycles per MFMA per SIMD at 2.4 GHz)\n",
           waves_per_simd, NACC, (int)LDS, ms, tf, 2.4e9 * ms * 1e-3 / (mfmas / 1024.0));
}

int main() {
    float* out;
    hipMalloc(&out, 4096);
    for (int w : {1, 2, 4}) {
        run<1, false>(w, out);
        run<2, false>(w, out);
        run<4, false>(w, out);
        run<2, true>(w, out);
        run<4, true>(w, out);
    }
    return 0;
}
