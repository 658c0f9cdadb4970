// The first convolution of the stem (hrnet.py:470: Conv2d(3, 64, 3, stride 2, padding 1) + BN + ReLU) on the fp32 matrix cores with the
// reduction flattened: K = (channel, tap) = 27 values + 1 zero = 7 k-steps of 4.  The tap-major layout of the generic kernel pads the
// 3 input channels to 8 per tap (18 k-steps, 27 of 72 K elements real) and stages stride-2 rows through the LDS; this layer's input is
// 0.6 MB per frame and its output 3.2 MB, so it is a byte-moving kernel: nothing here goes through the LDS and there is no barrier.
//   wave  = one output row (frame n, row y): Wo / 16 tiles of 16 pixels x all 64 output channels;
//   A[row = pixel l15][k = lq] of k-step s = in[n][c][2y + ky - 1][2x + kx - 1], k = 4s + lq = 9c + 3ky + kx -- one dword per lane and
//     k-step straight from global memory (L1/L2 hits: a frame's three planes are read 2.25 times), the next tile's seven requested
//     under the current tile's MFMAs; positions outside the image are a select to zero;
//   B[k = lq][col = l15] = folded weight of channel 16 nt + l15: 28 registers per lane, loaded once per wave (pack_stem_weights);
//   D[row = 4 lq + r][col = l15]: four consecutive pixels of one output channel per lane -> one 16-byte store per channel block.
#include "kernels.h"

namespace grk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__global__ __launch_bounds__(256) void conv_stem_f32(const ConvArgs a) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.N * a.Ho) return;
    const int n = row / a.Ho, y = row - n * a.Ho;
    const float* w = a.w;
    float bw[7][4];
#pragma unroll
    for (int s = 0; s < 7; ++s)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) bw[s][nt] = w[(s * 4 + nt) * 64 + lane];
    float bias[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) bias[nt] = a.bias[nt * 16 + l15];
    // per k-step: the input row this lane reads (nullptr-like flag: row outside the image) and its kx
    const float* rowp[7];
    int kxm1[7];
    bool yok[7];
    const float* inb = a.in + ((size_t)n * a.in_ctot + a.in_coff) * a.H * a.W;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const int k = 4 * s + lq, kk = k < 27 ? k : 26, c = kk / 9, t = kk - 9 * c, ky = t / 3, kx = t - 3 * ky;
        const int yin = 2 * y + ky - 1;
        yok[s] = yin >= 0 && yin < a.H;
        rowp[s] = inb + ((size_t)c * a.H + (yok[s] ? yin : 0)) * a.W;
        kxm1[s] = kx - 1;
    }
    auto fetch = [&](int x0, float (&v)[7]) {
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            const int xi = 2 * (x0 + l15) + kxm1[s];
            const bool ok = yok[s] && xi >= 0 && xi < a.W;
            const float t = rowp[s][ok ? xi : 0];
            v[s] = ok ? t : 0.f;
        }
    };
    float cur[7], nxt[7];
    fetch(0, cur);
    float* ob = a.out + (((size_t)n * a.out_ctot + a.out_coff + l15) * a.Ho + y) * a.Wo + 4 * lq;
    const size_t cstride = (size_t)16 * a.Ho * a.Wo;
    for (int x0 = 0; x0 < a.Wo; x0 += 16) {
        fetch(x0 + 16 < a.Wo ? x0 + 16 : x0, nxt);          // the last tile re-requests itself
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[s], bw[s][nt], acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            f32x4 v = acc[nt] + bias[nt];
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            *reinterpret_cast<f32x4*>(ob + nt * cstride + x0) = v;
        }
#pragma unroll
        for (int s = 0; s < 7; ++s) cur[s] = nxt[s];
    }
}

}  // namespace

bool conv_stem_eligible(int cin, int cout, int ks, int stride, int h, int w, int n_add) {
    return cin == 3 && cout == 64 && ks == 3 && stride == 2 && n_add == 0 && h % 2 == 0 && w % 32 == 0;
}

// out[(s * 4 + nt) * 64 + lq * 16 + l15] = w[cout = 16 nt + l15][k = 4 s + lq], k = 9 c + 3 ky + kx (0 for k = 27)
void pack_stem_weights(const double* w_folded /* (64,3,3,3) */, float* out /* 7*4*64 */) {
    for (int s = 0; s < 7; ++s)
        for (int nt = 0; nt < 4; ++nt)
            for (int lq = 0; lq < 4; ++lq)
                for (int l15 = 0; l15 < 16; ++l15) {
                    const int k = 4 * s + lq, co = 16 * nt + l15;
                    out[(s * 4 + nt) * 64 + lq * 16 + l15] = k < 27 ? (float)w_folded[(size_t)co * 27 + k] : 0.f;
                }
}

hipError_t launch_conv_stem(ConvArgs a, hipStream_t s) {
    if (!conv_stem_eligible(a.Cin, a.Cout, a.ks, a.stride, a.H, a.W, a.n_add) || a.Ho * 2 != a.H || a.Wo * 2 != a.W) return hipErrorInvalidValue;
    const int rows = a.N * a.Ho;
    return launch_k(conv_stem_f32, dim3((rows + 3) / 4), dim3(256), 0, s, a);
}

}  // namespace grk
