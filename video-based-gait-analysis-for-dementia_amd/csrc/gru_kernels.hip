// GRU gait-feature encoder (reference: BidirectionalModel.forward, use_pareFeat=True, eval;
// lib/models/layers/gait_feat_encoder.py:79-104; nn.GRU equations, gate order r,z,n, h0 = 0).
//   1. xin = x + cparam_mpl(cparams)           per-joint 3->128 locally connected (line 85)
//   2. gi  = xin . W_ih^T + b_ih               fp32 MFMA GEMM batched over all b*T rows, per direction
//   3. recurrence, one workgroup per (sequence, direction): W_hh is stored transposed [k][g] so the
//      900 gate rows are read coalesced from L2 each step; h lives in LDS
//   4. layer 1 on concat(fwd,bwd) of layer 0 (steps 2-3 again)
//   5. heads: speed/step MLPs on the final hidden states, phase MLP + tanh on the layer-1 outputs
#include "kernels.h"

namespace grk {

#define GRK_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// C[M][N] = A[M][K] . B[N][K]^T + bias[N]   (row-major, K contiguous, K % 4 == 0, 16-byte aligned rows)
// 64x64 tile, K-chunk 32, 4 waves as 2x2, each 2x2 MFMA 16x16x4 tiles.  LDS row stride 34 floats:
// the 32 lanes of a half-wave (16 rows x 2 k) fall on 32 distinct banks (2*row + k).
constexpr int kGemmLd = 34;
// blockIdx.z = K slice (split-K for GEMMs with few rows: slice z covers [z*kc, (z+1)*kc) and writes its partial sums to C + z*zstride;
// gemm_splitk_reduce adds the slices in a fixed order and the bias).  kc is a multiple of 32; one slice = the plain GEMM.
__global__ __launch_bounds__(256) void gemm_nt_bias_f32(const float* __restrict__ A, const float* __restrict__ B,
                                                          const float* __restrict__ bias, float* __restrict__ C, int M, int N, int K,
                                                          int ldc, int kc, size_t zstride) {
    __shared__ __align__(16) float As[64 * kGemmLd];
    __shared__ __align__(16) float Bs[64 * kGemmLd];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lr = tid >> 3, lk = (tid & 7) * 4;
    const int kbeg = blockIdx.z * kc, kend = (kbeg + kc < K) ? kbeg + kc : K;
    C += (size_t)blockIdx.z * zstride;
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = lr + 32 * h;
            f32x4 va = f32x4{0.f, 0.f, 0.f, 0.f}, vb = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k0 + lk < kend) {
                if (m0 + r < M) va = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + r) * K + k0 + lk);
                if (n0 + r < N) vb = *reinterpret_cast<const f32x4*>(B + (size_t)(n0 + r) * K + k0 + lk);
            }
            f32x2* pa = reinterpret_cast<f32x2*>(As + r * kGemmLd + lk);
            f32x2* pb = reinterpret_cast<f32x2*>(Bs + r * kGemmLd + lk);
            pa[0] = f32x2{va[0], va[1]}; pa[1] = f32x2{va[2], va[3]};
            pb[0] = f32x2{vb[0], vb[1]}; pb[1] = f32x2{vb[2], vb[3]};
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = As[(wm * 32 + i * 16 + l15) * kGemmLd + kk * 4 + lq];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = Bs[(wn * 32 + j * 16 + l15) * kGemmLd + kk * 4 + lq];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 32 + j * 16 + l15;
            if (n >= N) continue;
            const float bv = bias ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + i * 16 + lq * 4 + r;
                if (m < M) C[(size_t)m * ldc + n] = acc[i][j][r] + bv;
            }
        }
}

// C = bias + sum_z P[z]  (fixed order: deterministic)
__global__ __launch_bounds__(256) void gemm_splitk_reduce(const float* __restrict__ P, const float* __restrict__ bias, float* __restrict__ C, int M, int N,
                                                            int ldc, int splits) {
    const long total = (long)M * N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i - (long)m * N);
        float v = bias ? bias[n] : 0.f;
        for (int z = 0; z < splits; ++z) v += P[(size_t)z * total + i];
        C[(size_t)m * ldc + n] = v;
    }
}

hipError_t launch_gemm_nt_bias(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int ldc, hipStream_t s) {
    if (K % 4 != 0) return hipErrorInvalidValue;
    const int blocks = ((N + 63) / 64) * ((M + 63) / 64);
    // few rows (a clip of 16 frames is ONE row block): 47 workgroups would stream a 37 MB weight matrix; split K so that >= ~512 do
    int splits = 1;
    if (blocks < 128 && K >= 512) {                          // (at 188 blocks -- 8 clips x 32 frames -- splitting already costs more than it saves)
        splits = (512 + blocks - 1) / blocks;
        if (splits > 16) splits = 16;
        if (splits > K / 128) splits = K / 128;
    }
    if (splits <= 1) {
        GRK_TRY(launch_k(gemm_nt_bias_f32, dim3((N + 63) / 64, (M + 63) / 64, 1), dim3(256), 0, s, A, B, bias, C, M, N, K, ldc, K, (size_t)0));
        return hipGetLastError();
    }
    const int kc = ((K + splits - 1) / splits + 31) / 32 * 32;
    splits = (K + kc - 1) / kc;
    float* part = nullptr;                                  // per call: these GEMMs are not on the per-frame hot path and are not graph-captured
    GRK_TRY(hipMallocAsync(reinterpret_cast<void**>(&part), (size_t)splits * M * N * sizeof(float), s));
    hipError_t e = launch_k(gemm_nt_bias_f32, dim3((N + 63) / 64, (M + 63) / 64, splits), dim3(256), 0, s, A, B, static_cast<const float*>(nullptr), part, M, N, K, N, kc,
                            (size_t)M * N);
    if (e == hipSuccess) {
        const long total = (long)M * N;
        e = launch_k(gemm_splitk_reduce, dim3((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), dim3(256), 0, s, part, bias, C, M, N, ldc, splits);
    }
    hipFreeAsync(part, s);
    return e != hipSuccess ? e : hipGetLastError();
}

// xc[r, c*24+j] = sum_f cp[r,f] * wc[c,f,j];  xin = x + xc   (dropout is the identity in eval)
__global__ __launch_bounds__(256) void gru_prep_kernel(const float* __restrict__ x, const float* __restrict__ cp,
                                                         const float* __restrict__ wc, float* __restrict__ xc, float* __restrict__ xin,
                                                         long rows) {
    const long total = rows * 3072;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / 3072;
        const int e = (int)(i % 3072), c = e / 24, j = e % 24;
        const float v = cp[r * 3] * wc[(c * 3 + 0) * 24 + j] + cp[r * 3 + 1] * wc[(c * 3 + 1) * 24 + j] +
                        cp[r * 3 + 2] * wc[(c * 3 + 2) * 24 + j];
        xc[i] = v;
        xin[i] = x[i] + v;
    }
}

// One (sequence, direction) per workgroup of 1024 threads.  gi: (2, b*T, 900); w_hhT: [dir](300, 900).
constexpr int kH = 300;
__global__ __launch_bounds__(1024) void gru_recurrent_kernel(const float* __restrict__ gi, const float* __restrict__ w_hhT_f,
                                                               const float* __restrict__ w_hhT_b, const float* __restrict__ b_hh_f,
                                                               const float* __restrict__ b_hh_b, float* __restrict__ out,
                                                               float* __restrict__ hfin, int hfin_off, int b, int T) {
    __shared__ float h[kH];
    __shared__ float gh[3 * kH];
    const int seq = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const float* wT = dir ? w_hhT_b : w_hhT_f;
    const float* bh = dir ? b_hh_b : b_hh_f;
    const float* gid = gi + (size_t)dir * b * T * 900;
    if (tid < kH) h[tid] = 0.f;
    __syncthreads();
    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        if (tid < 3 * kH) {
            float acc = bh[tid];
#pragma unroll 10
            for (int k = 0; k < kH; ++k) acc += h[k] * wT[(size_t)k * 900 + tid];
            gh[tid] = acc;
        }
        __syncthreads();
        if (tid < kH) {
            const float* g = gid + ((size_t)seq * T + t) * 900;
            const float r = 1.f / (1.f + expf(-(g[tid] + gh[tid])));
            const float z = 1.f / (1.f + expf(-(g[kH + tid] + gh[kH + tid])));
            const float nn = tanhf(g[2 * kH + tid] + r * gh[2 * kH + tid]);
            const float hn = (1.f - z) * nn + z * h[tid];
            h[tid] = hn;
            out[((size_t)seq * T + t) * (2 * kH) + dir * kH + tid] = hn;
        }
        __syncthreads();
    }
    if (tid < kH) hfin[(size_t)seq * (4 * kH) + hfin_off + dir * kH + tid] = h[tid];
}

// hidden (rows,100) -> LeakyReLU(0.05) -> Linear(100 -> nout) [-> tanh]
__global__ __launch_bounds__(64) void mlp_out_kernel(const float* __restrict__ hidden, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, float* __restrict__ out, int nout, int ld_out,
                                                       int col_off, int do_tanh) {
    const long r = blockIdx.x;
    const int lane = threadIdx.x;
    float hv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int k = lane + 64 * i;
        float v = k < 100 ? hidden[r * 100 + k] : 0.f;
        hv[i] = v > 0.f ? v : 0.05f * v;
    }
    for (int o = 0; o < nout; ++o) {
        float acc = hv[0] * w2[o * 100 + lane] + (lane + 64 < 100 ? hv[1] * w2[o * 100 + lane + 64] : 0.f);
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
        if (lane == 0) {
            acc += b2[o];
            out[r * ld_out + col_off + o] = do_tanh ? tanhf(acc) : acc;
        }
    }
}

hipError_t launch_gru(const float* x, const float* cparams, GruWeights w, GruWorkspace ws, float* y, float* phase, float* xc,
                      int b, int T, hipStream_t s) {
    const long rows = (long)b * T;
    hipError_t e;
    int blocks = (int)((rows * 3072 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    GRK_TRY(launch_k(gru_prep_kernel, dim3(blocks), dim3(256), 0, s, x, cparams, w.cparam_w, xc, ws.xin, rows));
    const float* layer_in = ws.xin;
    int in_size = 3072;
    float* layer_out[2] = {ws.l0, ws.l1};
    for (int layer = 0; layer < 2; ++layer) {
        for (int d = 0; d < 2; ++d) {
            e = launch_gemm_nt_bias(layer_in, w.w_ih[layer][d], w.b_ih[layer][d], ws.gi + (size_t)d * rows * 900, (int)rows, 900, in_size, 900, s);
            if (e != hipSuccess) return e;
        }
        GRK_TRY(launch_k(gru_recurrent_kernel, dim3(b, 2), dim3(1024), 0, s, ws.gi, w.w_hh[layer][0], w.w_hh[layer][1],
                           w.b_hh[layer][0], w.b_hh[layer][1], layer_out[layer], ws.hfin, layer * 600, b, T));
        layer_in = layer_out[layer];
        in_size = 600;
    }
    // heads: hidden activations reuse the gi workspace
    float* hid = ws.gi;
    if ((e = launch_gemm_nt_bias(ws.hfin, w.speed_w0, w.speed_b0, hid, b, 100, 1200, 100, s)) != hipSuccess) return e;
    GRK_TRY(launch_k(mlp_out_kernel, dim3(b), dim3(64), 0, s, hid, w.speed_w2, w.speed_b2, y, 1, 3, 0, 0));
    float* hid2 = hid + (size_t)b * 100;
    if ((e = launch_gemm_nt_bias(ws.hfin, w.step_w0, w.step_b0, hid2, b, 100, 1200, 100, s)) != hipSuccess) return e;
    GRK_TRY(launch_k(mlp_out_kernel, dim3(b), dim3(64), 0, s, hid2, w.step_w2, w.step_b2, y, 2, 3, 1, 0));
    float* hid3 = hid2 + (size_t)b * 100;
    if ((e = launch_gemm_nt_bias(ws.l1, w.phase_w0, w.phase_b0, hid3, (int)rows, 100, 600, 100, s)) != hipSuccess) return e;
    GRK_TRY(launch_k(mlp_out_kernel, dim3((unsigned)rows), dim3(64), 0, s, hid3, w.phase_w2, w.phase_b2, phase, 4, 4, 0, 1));
    return hipGetLastError();
}

}  // namespace grk
