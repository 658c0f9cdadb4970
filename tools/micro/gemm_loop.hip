// Where does the fp32 GEMM loop lose its matrix-pipe time?  gemm_nt_bias_f32_128 (csrc/gru_kernels.hip) runs 10 000 x 3 000 x 3 072 at 92 TFLOP/s = 0.59 of the
// fp32 MFMA rate with 64x64 or 128x128 tiles, v_mfma 16x16x4 or 32x32x2, guarded or unguarded loads alike.  This is its chunk loop taken apart, 512 workgroups
// (2 per CU, as the kernel's 70 KB of LDS give), 4 waves each, 128 MFMAs per wave and chunk:
//   mode 0  operand reads from LDS + MFMAs only            mode 1  + one barrier per chunk
//   mode 2  + the 16 ds_write_b64 of the staging           mode 3  + the 8 global_load_dwordx4 of the next chunk (a real 10 112 x 3 072 matrix pair)
//   mode 4  mode 3 with 3 workgroups per CU (single LDS stage would allow it)      mode 5  mode 0 with the MFMAs alone (operands read once)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int LD = 34, STAGE = 2 * 128 * LD;
template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ A, const float* __restrict__ B, float* out, int K, int tiles_n) {
    extern __shared__ __align__(16) float gsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
    int tile = blockIdx.x;
    if (MODE == 7) { const int total = gridDim.x, per = (total + 7) / 8; tile = (tile & 7) * per + (tile >> 3); if (tile >= total) return; }
    const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
    for (int i = tid; i < (MODE == 4 ? 1 : 2) * STAGE; i += 256) gsm[i] = (float)((i * 7) % 13) * 0.25f - 1.5f;
    __syncthreads();
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lr = tid >> 3, lk = (tid & 7) * 4;
    f32x4 ra[4], rb[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) { ra[h] = f32x4{1.f, 2.f, 3.f, 4.f}; rb[h] = f32x4{.5f, .25f, .125f, 1.f}; }
    const unsigned oa = (unsigned)(m0 + lr) * (unsigned)K + lk, ob = (unsigned)(n0 + lr) * (unsigned)K + lk;
    int cur = 0;
    float a5[4], b5[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a5[i] = gsm[(wm * 64 + l15 + 16 * i) * LD + lq]; b5[i] = gsm[128 * LD + (wn * 64 + l15 + 16 * i) * LD + lq]; }
    for (int k0 = 0; k0 < K; k0 += 32, cur ^= (MODE == 4 ? 0 : 1)) {
        if (MODE >= 3 && MODE != 5 && k0 + 32 < K) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                ra[h] = *reinterpret_cast<const f32x4*>(A + (size_t)(k0 + 32) + (size_t)h * 32 * K + oa);
                rb[h] = *reinterpret_cast<const f32x4*>(B + (size_t)(k0 + 32) + (size_t)h * 32 * K + ob);
            }
        }
        const float* As = gsm + cur * STAGE + (wm * 64 + l15) * LD + lq;
        const float* Bs = gsm + cur * STAGE + 128 * LD + (wn * 64 + l15) * LD + lq;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = MODE == 5 ? a5[i] : As[i * 16 * LD + kk * 4];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = MODE == 5 ? b5[j] : Bs[j * 16 * LD + kk * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (MODE >= 2 && MODE != 5) {
            float* st = MODE == 4 ? gsm : gsm + (cur ^ 1) * STAGE;
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                f32x2* pa = reinterpret_cast<f32x2*>(st + (lr + 32 * h) * LD + lk);
                f32x2* pb = reinterpret_cast<f32x2*>(st + 128 * LD + (lr + 32 * h) * LD + lk);
                pa[0] = f32x2{ra[h][0], ra[h][1]}; pa[1] = f32x2{ra[h][2], ra[h][3]};
                pb[0] = f32x2{rb[h][0], rb[h][1]}; pb[1] = f32x2{rb[h][2], rb[h][3]};
            }
        }
        if (MODE >= 1 && MODE != 5) __syncthreads();
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[blockIdx.x * 256 + tid] = s;
}
template <int MODE>
void run(const float* A, const float* B, float* out, const char* what) {
    const int K = 3072, tiles_n = 24, wgs = (MODE == 4 ? 768 : MODE >= 6 ? 1896 : 512);
    const size_t lds = MODE == 4 ? 50 * 1024 : 2 * STAGE * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * STAGE * sizeof(float)));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 12; ++rep) {
        (void)hipEventRecord(e0);
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), MODE == 4 ? 2 * STAGE * sizeof(float) * 0 + lds : lds, 0, A, B, out, K, tiles_n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 3 && ms < best) best = ms;
    }
    printf("mode %d  %-62s %7.3f ms per launch  %6.1f TFLOP/s\n", MODE, what, best / 4, (double)wgs * 4 * (K / 32) * 128 * 2048.0 * 4 / best / 1e9);
}
int main() {
    float *A, *B, *out;
    const size_t na = (size_t)10240 * 3072, nb = (size_t)3072 * 3072;
    (void)hipMalloc(&A, na * 4); (void)hipMalloc(&B, nb * 4); (void)hipMalloc(&out, 1024 * 256 * 4);
    (void)hipMemset(A, 0, na * 4); (void)hipMemset(B, 0, nb * 4);
    run<5>(A, B, out, "MFMAs alone (operands read once)");
    run<0>(A, B, out, "operand reads from LDS + MFMAs");
    run<1>(A, B, out, "+ one barrier per chunk");
    run<2>(A, B, out, "+ staging stores (16 ds_write_b64)");
    run<3>(A, B, out, "+ next chunk's global loads (the kernel's loop)");
    run<4>(A, B, out, "the same, 3 workgroups per CU");
    run<6>(A, B, out, "mode 3 on the GEMM's 79 x 24 = 1 896 tiles (3.7 rounds)");
    run<7>(A, B, out, "... dealt to the XCDs in row bands, as the kernel does");
    {   // the same two with random matrices instead of zeros
        float* h = (float*)malloc(na * 4);
        srand(3);
        for (size_t i = 0; i < na; ++i) h[i] = ((float)(rand() & 0xffff) / 65536.0f - 0.5f) * 2.0f;
        (void)hipMemcpy(A, h, na * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(B, h + 12345, nb * 4, hipMemcpyHostToDevice);
        free(h);
    }
    run<3>(A, B, out, "mode 3, random matrices");
    run<6>(A, B, out, "mode 6, random matrices");
    return 0;
}
