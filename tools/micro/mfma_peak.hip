// fp32 MFMA rate of this box: back-to-back v_mfma_f32_16x16x4_f32 on register operands.  Round 1's version timed ONE 0.4 ms launch
// with 7 accumulators and read 111-135 TFLOP/s -- short enough to sit on the clock ramp.  This one runs each configuration for
// ~100 ms (25 launches of ~4 ms after a warm-up), with 4 / 8 / 12 independent accumulators per wave and 1-4 waves per SIMD, and
// reports the best launch of each.  The guide's figure for the same instruction is 155 TFLOP/s measured (157.3 nominal).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(float* out, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 160000 / NACC;                              // ~4 ms per launch at one wave per SIMD
    float best = 1e30f;
    for (int rep = 0; rep < 30; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 5 && ms < best) best = ms;
    }
    const double flops = (double)blocks * 4 * iters * NACC * 2048.0;
    printf("acc %2d  blocks %4d (%.0f waves/SIMD): best %.3f ms  %.1f TFLOP/s\n", NACC, blocks, blocks * 4 / 1024.0, best, flops / best / 1e9);
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    for (int blocks : {256, 512, 1024}) { run<4>(out, blocks); run<8>(out, blocks); run<12>(out, blocks); }
    return 0;
}
