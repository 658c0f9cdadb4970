// Practical fp32 MFMA ceiling on this box: back-to-back v_mfma_f32_16x16x4_f32, operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {256, 512, 1024, 2048}) {
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 4000;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)blocks * 4 * iters * 7 * 2048.0;
            if (rep) printf("blocks %4d (%.1f waves/SIMD): %.3f ms  %.1f TFLOP/s\n", blocks, blocks * 4 / 1024.0, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
