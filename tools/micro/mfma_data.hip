// Does the fp32 MFMA rate depend on the DATA?  mfma_peak.hip issues back-to-back v_mfma_f32_16x16x4_f32 on ONE constant operand pair (155 TFLOP/s); every
// real fp32 kernel of this library (Winograd, GEMM 64/128 tiles, 16x16x4 or 32x32x2, blocked attention) levels off near 0.6 of that whatever its tiling.
// Here the same loop rotates through 16 operand pairs held in registers -- no instruction between two MFMAs in either mode -- filled with (0) one constant,
// (1) small integers, (2) uniform random mantissas, (3) random values AND signs; each mode runs ~1.5 s so that a power / clock response shows, and prints
// the rate of every ~100 ms slice.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const float* __restrict__ ops, float* out, int iters) {
    f32x4 acc[16];
    float a[16], b[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        acc[i] = f32x4{0, 0, 0, 0};
        a[i] = ops[(i * 2 + 0) * 256 + threadIdx.x];
        b[i] = ops[(i * 2 + 1) * 256 + threadIdx.x];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *out, *ops;
    hipMalloc(&out, 4096 * 256 * 4);
    hipMalloc(&ops, 32 * 256 * 4);
    float h[32 * 256];
    const char* names[] = {"one constant pair", "small integers", "random mantissas, positive", "random values and signs"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512, iters = 20000;                       // 2 waves per SIMD, ~8.5 ms per launch at peak
    for (int mode = 0; mode < 4; ++mode) {
        srand(1);
        for (int i = 0; i < 32 * 256; ++i) {
            const float u = (float)rand() / RAND_MAX;
            h[i] = mode == 0 ? 1.0f : mode == 1 ? (float)(rand() % 7 - 3) : mode == 2 ? 1.0f + u : (u - 0.5f) * 4.0f * (1.0f + (rand() % 1000) * 1e-3f);
        }
        hipMemcpy(ops, h, sizeof h, hipMemcpyHostToDevice);
        printf("%-28s:", names[mode]);
        for (int slice = 0; slice < 12; ++slice) {
            hipEventRecord(e0);
            for (int r = 0; r < 12; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, (const float*)ops, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf(" %.0f", (double)blocks * 4 * iters * 16 * 2048.0 * 12 / ms / 1e9);
        }
        printf("  TFLOP/s per ~100 ms slice\n");
    }
    return 0;
}
