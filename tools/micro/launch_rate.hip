// How fast can dependent kernels be dispatched?  k streams x 300 launches of an (almost) empty kernel with the grid of a
// small convolution (448 workgroups x 512 threads, 45 KB of LDS), launched from one host thread round-robin.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/micro/launch_rate.hip -o /tmp/launch_rate && /tmp/launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void empty_kernel(float* p) {
    extern __shared__ float sm[];
    if (threadIdx.x == 0 && p == nullptr) sm[0] = 1.f;
}
int main() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(empty_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    float* buf;
    hipMalloc(&buf, 1024);
    for (int ns : {1, 2, 4, 8}) {
        hipStream_t st[8];
        for (int i = 0; i < ns; ++i) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        for (int lds : {0, 45 * 1024}) {
            for (int wg : {16, 448}) {
                for (int i = 0; i < ns; ++i) hipLaunchKernelGGL(empty_kernel, dim3(wg), dim3(512), lds, st[i], buf);
                hipDeviceSynchronize();
                const int reps = 300;
                auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; ++r)
                    for (int i = 0; i < ns; ++i) hipLaunchKernelGGL(empty_kernel, dim3(wg), dim3(512), lds, st[i], buf);
                auto t1 = std::chrono::steady_clock::now();
                hipDeviceSynchronize();
                auto t2 = std::chrono::steady_clock::now();
                const double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * ns);
                const double all_us = std::chrono::duration<double, std::micro>(t2 - t0).count() / (reps * ns);
                printf("streams %d  grid %3d x 512  lds %2d KB: host %.2f us/launch, end-to-end %.2f us/launch (system-wide)\n", ns, wg, lds / 1024, host_us, all_us);
            }
        }
        for (int i = 0; i < ns; ++i) hipStreamDestroy(st[i]);
    }
    return 0;
}
