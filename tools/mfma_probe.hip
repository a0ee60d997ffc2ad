// Probe: how does v_mfma_f32_32x32x16_bf16 accumulate its 16 products?  Compares one instruction's result on
// random (wide-exponent) inputs with host models:
//   seq     : acc = c; for k = 0..15: acc = fl32(acc + a_k * b_k)           (products exact in fp32)
//   seq_rev : the same, k = 15..0
//   halves  : two sequential chains k = 0..7 and 8..15 (the two 8-element lane groups), then summed with c
//   quads   : four chains of 4, summed pairwise
//   exact   : c + sum_k a_k b_k in long double, rounded once
// Build:  hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/_build/mfma_probe      Run on a GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(const uint16_t* A, const uint16_t* B, const float* C, float* D) {
    // A[i][k] (32 x 16), B[k][j] (16 x 32), row-major; lane l: i or j = l % 32, k = 8 * (l / 32) + 0..7
    const int lane = threadIdx.x, ij = lane & 31, kb = 8 * (lane >> 5);
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = __builtin_bit_cast(__bf16, A[ij * 16 + kb + e]);
        b[e] = __builtin_bit_cast(__bf16, B[(kb + e) * 32 + ij]);
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + ij];
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + ij] = d[r];
}

static float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t tobf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }

int main() {
    const int trials = 200;
    int ok[5] = {0, 0, 0, 0, 0}, total = 0;
    uint16_t hA[32 * 16], hB[16 * 32];
    float hC[32 * 32], hD[32 * 32];
    uint16_t *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, sizeof hC); hipMalloc(&dD, sizeof hD);
    srand(1);
    for (int t = 0; t < trials; ++t) {
        const int spread = (t % 4) * 6;     // exponent spread 0, 6, 12, 18 bits
        for (int i = 0; i < 32 * 16; ++i) {
            hA[i] = tobf(((rand() / (float)RAND_MAX) * 2 - 1) * ldexpf(1.0f, spread ? rand() % spread : 0));
            hB[i] = tobf(((rand() / (float)RAND_MAX) * 2 - 1) * ldexpf(1.0f, spread ? rand() % spread : 0));
        }
        for (int i = 0; i < 32 * 32; ++i) hC[i] = (t & 1) ? 0.0f : ((rand() / (float)RAND_MAX) * 2 - 1) * 8.0f;
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        hipMemcpy(dC, hC, sizeof hC, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float p[16];
                for (int k = 0; k < 16; ++k) p[k] = bf(hA[i * 16 + k]) * bf(hB[k * 32 + j]);   // exact in fp32
                const float c = hC[i * 32 + j];
                volatile float s;
                float m[5];
                s = c; for (int k = 0; k < 16; ++k) s = s + p[k]; m[0] = s;
                s = c; for (int k = 15; k >= 0; --k) s = s + p[k]; m[1] = s;
                { volatile float h0 = 0, h1 = 0; for (int k = 0; k < 8; ++k) { h0 = h0 + p[k]; h1 = h1 + p[8 + k]; } s = h0 + h1; s = s + c; m[2] = s; }
                { volatile float q[4] = {0, 0, 0, 0}; for (int g = 0; g < 4; ++g) for (int k = 0; k < 4; ++k) q[g] = q[g] + p[4 * g + k];
                  volatile float u = q[0] + q[1], v = q[2] + q[3]; s = u + v; s = s + c; m[3] = s; }
                { long double e = c; for (int k = 0; k < 16; ++k) e += (long double)p[k]; m[4] = (float)e; }
                for (int x = 0; x < 5; ++x) ok[x] += (memcmp(&m[x], &hD[i * 32 + j], 4) == 0);
                ++total;
            }
    }
    const char* names[5] = {"seq", "seq_rev", "halves", "quads", "exact"};
    for (int x = 0; x < 5; ++x) printf("%-8s matches %d of %d outputs (%.2f %%)\n", names[x], ok[x], total, 100.0 * ok[x] / total);
    return 0;
}
