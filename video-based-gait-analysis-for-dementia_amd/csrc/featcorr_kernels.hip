// Pose-feature corrector around the GRU gait encoder and the attention block.
// Reference: FeatCorrector.forward, eval -- lib/models/layers/feature_correction.py:104-157 -- in the configuration GRNet builds
// (grnet.py:69-79, configs/config_grnet.yaml: one layer, 4 heads, h_size 1024 -> 1000, use_jwff), and the camera parameters of the
// use_gait_feat branch of GRNet.forward (grnet.py:156-160).  The class reads names that are defined nowhere; they are
// bound as DESIGN.md records (use_leff = leff_smpl_feats = False, N = n); tests/golden/featcorr.npz holds the outputs of the
// reference's own code run with exactly those bindings.
//   cparams  = [bs*s, (bbox_xy - cimg)/scale/112 + t]                                      grnet.py:157-160
//   avg, phase, . = featnet(x, cparams)                                                     gru_kernels.hip
//   raw      = [avg (3) | phase[:2]/|phase[:2]| | phase[2:]/|phase[2:]|]                    :117-127
//   g_t      = W_t3 . lrelu(W_t0 . raw + b) + b   (7 -> 1536 -> 3072)                       :128, 66-73
//   g_s      = W_s3 . lrelu(W_s0 . raw + b) + b   (7 -> 64 -> 128)                          :129, 75-82
//   y        = BN1d(x + g_t)            y_s = BN1d_s([x | g_s])       (running statistics)   :130-139
//   y        = TSAttnBlock(y as (128,24), y_s as (128,25))                                   :143-145   tsattn_kernels.hip
//   out      = y + x                                                                         :150
// Everything is fp32 byte work except the 1536 -> 3072 layer, which runs on the fp32 matrix cores (gemm_nt_bias_f32).
#include "kernels.h"

namespace grk {

#define GRK_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

// cam (M,3) [s,tx,ty], bbox (M,4) [cx,cy,w,h], cimg (M,2) -> cparams (M,3)
__global__ __launch_bounds__(256) void gait_cparams_kernel(const float* __restrict__ cam, int cam_ld, const float* __restrict__ bbox,
                                                             const float* __restrict__ cimg, float* __restrict__ cp, int M) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float bs = bbox[m * 4 + 2] / 224.0f;
    const float scale = bs * cam[(size_t)m * cam_ld];
    cp[m * 3 + 0] = scale;
    cp[m * 3 + 1] = (bbox[m * 4 + 0] - cimg[m * 2 + 0]) / scale / 112.0f + cam[(size_t)m * cam_ld + 1];
    cp[m * 3 + 2] = (bbox[m * 4 + 1] - cimg[m * 2 + 1]) / scale / 112.0f + cam[(size_t)m * cam_ld + 2];
}

// one workgroup per row m = (clip, frame): raw gait features, both first layers, the whole small MLP
__global__ __launch_bounds__(256) void gfeat_hidden_kernel(const float* __restrict__ avg, const float* __restrict__ phase, FeatCorrWeights w,
                                                             float* __restrict__ hid_t, float* __restrict__ g_s, int n) {
    __shared__ float raw[7];
    __shared__ float hs[64];
    const int m = blockIdx.x, tid = threadIdx.x, clip = m / n;
    if (tid < 3) raw[tid] = avg[clip * 3 + tid];
    if (tid >= 32 && tid < 36) {
        const int k = tid - 32;
        const float* p = phase + (size_t)m * 4;
        const float a = p[k & 2], b = p[(k & 2) + 1];
        raw[3 + k] = p[k] / sqrtf(a * a + b * b);
    }
    __syncthreads();
    for (int o = tid; o < 1536; o += 256) {
        float acc = w.t0_b[o];
#pragma unroll
        for (int k = 0; k < 7; ++k) acc += w.t0_w[o * 7 + k] * raw[k];
        hid_t[(size_t)m * 1536 + o] = acc > 0.f ? acc : 0.05f * acc;
    }
    if (tid < 64) {
        float acc = w.s0_b[tid];
#pragma unroll
        for (int k = 0; k < 7; ++k) acc += w.s0_w[tid * 7 + k] * raw[k];
        hs[tid] = acc > 0.f ? acc : 0.05f * acc;
    }
    __syncthreads();
    if (tid < 128) {
        float acc = w.s3_b[tid];
        for (int k = 0; k < 64; ++k) acc += w.s3_w[tid * 64 + k] * hs[k];
        g_s[(size_t)m * 128 + tid] = acc;
    }
}

// y = BN(x + g_t) (M,3072); y_s = BN_s([x | g_s]) (M,3200); BN folded to scale / shift at load
__global__ __launch_bounds__(256) void featcorr_bn_kernel(const float* __restrict__ x, const float* __restrict__ g_t, const float* __restrict__ g_s,
                                                            FeatCorrWeights w, float* __restrict__ y, float* __restrict__ ys, long M) {
    const long total = M * 3200;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / 3200;
        const int e = (int)(i - m * 3200);
        if (e < 3072) {
            const float xv = x[m * 3072 + e];
            y[m * 3072 + e] = (xv + g_t[m * 3072 + e]) * w.bn_scale[e] + w.bn_shift[e];
            ys[i] = xv * w.bns_scale[e] + w.bns_shift[e];
        } else {
            ys[i] = g_s[m * 128 + e - 3072] * w.bns_scale[e] + w.bns_shift[e];
        }
    }
}

__global__ __launch_bounds__(256) void residual_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) out[i] = a[i] + b[i];
}

size_t featcorr_ws_floats(int b, int n) {
    const size_t M = (size_t)b * n;
    return M * (1536 + 3072 + 128 + 3072 + 3200 + 3072) + tsattn_ws_floats(b, n) + 256;
}

hipError_t launch_gait_cparams(const float* cam, int cam_ld, const float* bbox, const float* cimg, float* cparams, int M, hipStream_t s) {
    GRK_TRY(launch_k(gait_cparams_kernel, dim3((M + 255) / 256), dim3(256), 0, s, cam, cam_ld, bbox, cimg, cparams, M));
    return hipGetLastError();
}

// x (b,n,3072), avg (b,3), phase (b,n,4) [the GRU's outputs for x] -> out (b*n,128,24) = corrected pose features
hipError_t launch_featcorr(const float* x, const float* avg, const float* phase, const FeatCorrWeights& w, const TsAttnWeights& tw, float* ws, float* out,
                           int b, int n, hipStream_t s) {
    const size_t M = (size_t)b * n;
    float* hid_t = ws;
    float* g_t = hid_t + M * 1536;
    float* g_s = g_t + M * 3072;
    float* y = g_s + M * 128;
    float* ys = y + M * 3072;
    float* att = ys + M * 3200;
    float* tws = att + M * 3072;
    GRK_TRY(launch_k(gfeat_hidden_kernel, dim3((unsigned)M), dim3(256), 0, s, avg, phase, w, hid_t, g_s, n));
    GRK_TRY(launch_gemm_nt_bias(hid_t, w.t3_w, w.t3_b, g_t, (int)M, 3072, 1536, 3072, s));
    int blocks = (int)((M * 3200 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    GRK_TRY(launch_k(featcorr_bn_kernel, dim3(blocks), dim3(256), 0, s, x, (const float*)g_t, (const float*)g_s, w, y, ys, (long)M));
    GRK_TRY(launch_tsattn(y, ys, tw, tws, att, b, n, s));
    GRK_TRY(launch_k(residual_add_kernel, dim3(blocks), dim3(256), 0, s, (const float*)att, x, out, (long)(M * 3072)));
    return hipGetLastError();
}

}  // namespace grk
