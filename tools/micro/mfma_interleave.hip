// Does a wave's own vector-ALU / LDS work hide behind its MFMAs?  12 independent accumulators (the pure loop reaches the 155 TFLOP/s
// peak, mfma_peak.hip); behind every MFMA sit NV independent v_add_f32 and NL ds_read_b32 (waited for once per 12 MFMAs), pinned
// in place by sched_barrier.  One and two waves per SIMD.  Reported: cycles per MFMA at 2.4 GHz (32 = matrix pipe never idle).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV, int NL>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    f32x4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    float x[4] = {a, b, a + b, a - b}, r[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) r[i] = 0.f;
    const float* lp = lds + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[v & 3]) : "v"(b));
            if (NL && (i % (NL == 1 ? 1 : 2)) == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[i]) : "v"((unsigned)(size_t)lp), "n"(0));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NL) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); 
#pragma unroll
            for (int i = 0; i < 12; ++i) x[0] += r[i] * 0.f; }
    }
    float s = x[0] + x[1] + x[2] + x[3];
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV, int NL>
void run(float* out, int blocks) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 8000;
    float best = 1e30f;
    for (int rep = 0; rep < 12; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, NL>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3 && ms < best) best = ms;
    }
    const double mfma_per_simd = (double)(blocks / 256) * iters * 12;
    printf("valu/mfma %d  lds/mfma %s  waves/SIMD %d: %.3f ms  %.1f cycles per MFMA per SIMD (2.4 GHz)\n", NV, NL == 0 ? "0" : NL == 1 ? "1" : "1/2", blocks / 256, best,
           best * 1e-3 * 2.4e9 / mfma_per_simd);
}
int main() {
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    for (int blocks : {256, 512}) {
        run<0, 0>(out, blocks); run<1, 0>(out, blocks); run<2, 0>(out, blocks); run<4, 0>(out, blocks); run<6, 0>(out, blocks);
        run<0, 1>(out, blocks); run<0, 2>(out, blocks); run<2, 1>(out, blocks);
    }
    return 0;
}
