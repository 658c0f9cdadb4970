// Micro-benchmark: what does the traffic of layer1's 64 -> 256 1x1 convolution with residual (bf16 NHWC, 256 frames: read 103 MB + 411 MB, write 411 MB)
// cost as PLAIN streaming kernels of different shapes?  The convolution kernels run it at 3.6-3.9 TB/s; torch's add (2 reads + 1 write) runs at 6.0 TB/s.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/stream_mix.hip -o /tmp/stream_mix && /tmp/stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ u32x4 mix(u32x4 r, u32x4 i) { return u32x4{r[0] + i[0], r[1] ^ i[1], r[2] + i[2], r[3] ^ i[3]}; }

// one thread = 16 bytes of the output (8 of a pixel's 256 channels): its residual 16 bytes + the pixel's input 16-byte piece (part % 8 of 128 bytes)
__global__ __launch_bounds__(256) void k_oneshot(const u32x4* __restrict__ in, const u32x4* __restrict__ res, u32x4* __restrict__ out, long units) {
    const long u = (long)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const long px = u >> 5;
    out[u] = mix(res[u], in[px * 8 + (u & 7)]);
}
template <int UNROLL>
__global__ __launch_bounds__(256) void k_persist(const u32x4* __restrict__ in, const u32x4* __restrict__ res, u32x4* __restrict__ out, long units) {
    const long stride = (long)gridDim.x * 256 * UNROLL;
    for (long base = (long)blockIdx.x * 256 * UNROLL + threadIdx.x; base < units; base += stride) {
        u32x4 r[UNROLL], i[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) { const long u = base + k * 256; r[k] = res[u]; i[k] = in[(u >> 5) * 8 + (u & 7)]; }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) out[base + k * 256] = mix(r[k], i[k]);
    }
}
// phased like the convolution: a 512-thread workgroup takes 112 pixels: input (14 KB) + residual (56 KB) -> LDS, barrier, LDS -> output
__global__ __launch_bounds__(512) void k_tile(const u32x4* __restrict__ in, const u32x4* __restrict__ res, u32x4* __restrict__ out, long pixels) {
    extern __shared__ u32x4 lds[];
    u32x4* lin = lds;             // 112 * 8
    u32x4* lres = lds + 112 * 8;  // 112 * 32
    const long p0 = (long)blockIdx.x * 112;
    const int tid = threadIdx.x;
    for (int u = tid; u < 112 * 8; u += 512) lin[u] = in[p0 * 8 + u];
    for (int u = tid; u < 112 * 32; u += 512) lres[u] = res[p0 * 32 + u];
    __syncthreads();
    for (int u = tid; u < 112 * 32; u += 512) out[p0 * 32 + u] = mix(lres[u], lin[(u >> 5) * 8 + (u & 7)]);
}
// the same with the loads as LDS-DMA (what the convolution kernels use)
__global__ __launch_bounds__(512) void k_tile_dma(const u32x4* __restrict__ in, const u32x4* __restrict__ res, u32x4* __restrict__ out, long pixels) {
    extern __shared__ u32x4 lds[];
    u32x4* lin = lds;
    u32x4* lres = lds + 112 * 8 + 128;
    const long p0 = (long)blockIdx.x * 112;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int ub = wave * 64; ub < 112 * 8; ub += 512)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(in + p0 * 8 + ub + lane), (__attribute__((address_space(3))) void*)(lin + ub), 16, 0, 0);
    for (int ub = wave * 64; ub < 112 * 32; ub += 512)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(res + p0 * 32 + ub + lane), (__attribute__((address_space(3))) void*)(lres + ub), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int u = tid; u < 112 * 32; u += 512) out[p0 * 32 + u] = mix(lres[u], lin[(u >> 5) * 8 + (u & 7)]);
}

int main() {
    const long N = 256, HW = 3136, pixels = N * HW, units = pixels * 32;
    u32x4 *in, *res, *out;
    CK(hipMalloc(&in, pixels * 128)); CK(hipMalloc(&res, units * 16)); CK(hipMalloc(&out, units * 16));
    CK(hipMemset(in, 1, pixels * 128)); CK(hipMemset(res, 2, units * 16));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double bytes = pixels * 128.0 + 2.0 * units * 16;
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
        printf("%-28s %8.1f us  %.2f TB/s\n", name, ms * 1e3, bytes / ms * 1e-9);
    };
    run("oneshot 16B/thread", [&] { k_oneshot<<<dim3((unsigned)((units + 255) / 256)), 256>>>(in, res, out, units); });
    for (int g : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        char nm[64];
        snprintf(nm, sizeof nm, "persistent x1 grid %d", g); run(nm, [&] { k_persist<1><<<g, 256>>>(in, res, out, units); });
        snprintf(nm, sizeof nm, "persistent x4 grid %d", g); run(nm, [&] { k_persist<4><<<g, 256>>>(in, res, out, units); });
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tile_dma), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    for (int lds : {72 * 1024, 80 * 1024}) {     // 72 KB: two workgroups per CU; 80 KB: two as well (160 KB) -- 112*(8+32)*16 = 71 680 B needed
        char nm[64];
        snprintf(nm, sizeof nm, "tile 112 px, lds %d", lds); run(nm, [&] { k_tile<<<dim3((unsigned)(pixels / 112)), 512, lds>>>(in, res, out, pixels); });
        snprintf(nm, sizeof nm, "tile 112 px DMA, lds %d", lds); run(nm, [&] { k_tile_dma<<<dim3((unsigned)(pixels / 112)), 512, lds>>>(in, res, out, pixels); });
    }
    return 0;
}
