// The product's fp32 GEMM launcher (csrc/gru_kernels.hip, compiled in here) alone on random matrices: the shapes of the temporal branch at 10 000 frames,
// then two of them behind 9 ms of a nearly idle device (64 waiting workgroups, the GRU recurrence's footprint), behind 9 ms of GEMMs, and behind an empty queue.
// Results of round 6: profiles/r06_fp32_gemm_micro.txt.
#include "../../video-based-gait-analysis-for-dementia_amd/csrc/gru_kernels.hip"
#include <cstdio>
#include <cstdlib>
thread_local grk::GraphRecorder* grk::g_recorder = nullptr;
__global__ void spin_kernel(long long cycles, float* out) {                 // 64 workgroups that only wait: the recurrence's footprint on the device
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = 1.f;
}
int main() {
    const int M = 10000;
    float *A, *B, *C, *bias;
    const size_t na = (size_t)M * 3200, nb = (size_t)3072 * 3200, nc = (size_t)M * 3072;
    (void)hipMalloc(&A, na * 4); (void)hipMalloc(&B, nb * 4); (void)hipMalloc(&C, nc * 4); (void)hipMalloc(&bias, 4096 * 4);
    float* h = (float*)malloc(na * 4);
    srand(3);
    for (size_t i = 0; i < na; ++i) h[i] = ((float)(rand() & 0xffff) / 65536.0f - 0.5f) * 2.0f;
    (void)hipMemcpy(A, h, na * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, h + 777, nb * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(bias, h, 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int shapes[][2] = {{3000, 3072}, {3000, 3200}, {3072, 1536}, {3072, 1000}, {900, 3072}, {900, 600}};
    for (auto& sh : shapes) {
        const int N = sh[0], K = sh[1];
        float best = 1e30f;
        for (int rep = 0; rep < 10; ++rep) {
            (void)hipEventRecord(e0);
            for (int r = 0; r < 3; ++r) (void)grk::launch_gemm_nt_bias(A, B, bias, C, M, N, K, N, 0);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2 && ms < best) best = ms;
        }
        printf("M %d N %4d K %4d   %.3f ms  %6.1f TFLOP/s\n", M, N, K, best / 3, 2.0 * M * N * K * 3 / best / 1e9);
    }
    // the same GEMM behind ~9 ms of a nearly idle device (as behind the GRU recurrence: 64 workgroups), and behind 9 ms of other GEMMs
    for (int lead = 0; lead < 3; ++lead) {
        float tot = 0;
        for (int rep = 0; rep < 6; ++rep) {
            if (lead == 0) hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(512), 0, 0, 900000LL, C);      // wall_clock64: 100 MHz -> 9 ms
            if (lead == 1) for (int r = 0; r < 6; ++r) (void)grk::launch_gemm_nt_bias(A, B, bias, C, M, 3000, 3072, 3000, 0);
            if (lead == 2) { (void)hipDeviceSynchronize(); }
            (void)hipEventRecord(e0);
            (void)grk::launch_gemm_nt_bias(A, B, bias, C, M, 3000, 3072, 3000, 0);
            (void)grk::launch_gemm_nt_bias(A, B, bias, C, M, 3000, 3200, 3000, 0);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 1) tot += ms;
        }
        printf("two big GEMMs behind %-28s %.3f ms\n", lead == 0 ? "9 ms of 64 waiting workgroups:" : lead == 1 ? "9 ms of GEMMs:" : "an empty queue:", tot / 5);
    }
    return 0;
}
