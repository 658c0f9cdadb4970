// The 64 -> 256 1x1 convolutions of layer1's bottlenecks (hrnet.py:80-100: expansion + BN + residual + ReLU, and the shortcut of the first
// block) on the fp32 matrix cores with both operands straight from global memory.  These layers move 115 MB per launch for 1.6 GFLOP:
// they are byte-moving kernels, and the generic whole-K kernel (LDS-staged 16-channel chunks, a barrier per chunk, 112-pixel tiles)
// runs them at 3.2-3.7 TB/s (31 us); this one at 23 us.  (The 256 -> 64 reductions stay on the generic kernel: with K = 256 a wave's
// weights are 64 KB against 32 KB of input -- 16 channels per wave read the input four times (61 us), 64 channels with the weights
// re-loaded per chunk of 64 channels 36 us, against 28; and 64 -> 64 is 224 waves in all (21 against 11 us).)  Here:
//   wave  = a run of 14 tiles of 16 consecutive pixels of one frame x NT blocks of 16 output channels; its weights (K / 4 k-steps x NT
//           fragments = 64 registers for 64 -> 64 channels) are loaded once and stay in registers;
//   A[row = pixel l15][k = lq] of k-step s = in[channel 4 s + lq][pixel]: NCHW keeps a channel's pixels contiguous, so a k-step's
//           operand is one dword per lane (16 lanes = 64 contiguous bytes per channel), the next tile's requested under this tile's MFMAs;
//   B[k = lq][col = l15] = folded weight of output channel 16 nt + l15;
//   D[row = 4 lq + r][col = l15]: four consecutive pixels of one output channel per lane -> bias, residual and ReLU on 16-byte
//           vectors, one 16-byte store per channel block.  No LDS, no barrier.
#include "kernels.h"

namespace grk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kPwRun = 14;     // tiles per wave: 224 pixels = 4 rows of a 56-wide map

template <int KC, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_pw_f32(const ConvArgs a) {
    constexpr int KS = KC / 4;
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
    const int HW = a.H * a.W, runs = HW / (16 * kPwRun), groups = a.Cout / (16 * NT);
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= a.N * runs * groups) return;
    // consecutive waves share the pixels and differ in the channel group: the input run is read from HBM once and from L2 after that
    const int g = w % groups, t = w / groups, n = t / runs, px0 = (t - n * runs) * 16 * kPwRun, co0 = g * 16 * NT;
    float bw[KS][NT];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bw[s][nt] = a.w[(size_t)(4 * s + lq) * a.CoutPad + co0 + nt * 16 + l15];
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = a.bias[co0 + nt * 16 + l15];
    const float* inp = a.in + ((size_t)n * a.in_ctot + a.in_coff + lq) * HW + px0 + l15;            // channel lq, this lane's pixel of tile 0
    float* outp = a.out + ((size_t)n * a.out_ctot + a.out_coff + co0 + l15) * HW + px0 + 4 * lq;       // channel co0 + l15, pixels 4 lq ..
    const float* addp = a.n_add ? a.add[0] + ((size_t)n * a.add_ctot[0] + a.add_coff[0] + co0 + l15) * HW + px0 + 4 * lq : nullptr;
    const size_t cstr = (size_t)16 * HW;
    float cur[KS], nxt[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) cur[s] = inp[(size_t)(4 * s) * HW];
    for (int tile = 0; tile < kPwRun; ++tile) {
        const int tn = tile + 1 < kPwRun ? tile + 1 : tile;     // the last tile re-requests itself
        f32x4 res[NT];
        if (addp) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) res[nt] = *reinterpret_cast<const f32x4*>(addp + nt * cstr + 16 * tile);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) nxt[s] = inp[(size_t)(4 * s) * HW + 16 * tn];
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[s], bw[s][nt], acc[nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 v = acc[nt] + bias[nt];
            if (addp) v += res[nt];
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            *reinterpret_cast<f32x4*>(outp + nt * cstr + 16 * tile) = v;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
    }
}

}  // namespace

bool conv_pw_eligible(int cin, int cout, int ks, int stride, int h, int w, int n_add) {
    return ks == 1 && stride == 1 && n_add <= 1 && (h * w) % (16 * kPwRun) == 0 && cin == 64 && cout % 64 == 0;
}

// a.w: the direct kernels' packing [CinPad][CoutPad] (one tap)
hipError_t launch_conv_pw(ConvArgs a, hipStream_t s) {
    if (!conv_pw_eligible(a.Cin, a.Cout, a.ks, a.stride, a.H, a.W, a.n_add) || (a.n_add == 1 && a.add_shift[0] != 0) || a.CinPad < a.Cin || a.CoutPad < a.Cout)
        return hipErrorInvalidValue;
    const int runs = a.H * a.W / (16 * kPwRun);
    return launch_k(conv_pw_f32<64, 4>, dim3((a.N * runs * (a.Cout / 64) + 3) / 4), dim3(256), 0, s, a);
}

}  // namespace grk
