// 1x1 convolutions on 56x56 maps whose weights fit a wave's registers -- layer1's 64 -> 256 expansions and shortcut (hrnet.py:80-100,
// + BN + residual + ReLU) and the PARE head's 128 -> 25 heat-map layer (pare.py:388-397) -- on the fp32 matrix cores with both operands
// straight from global memory.  The expansions move 115 MB per launch for 1.6 GFLOP: byte-moving kernels, which the generic whole-K
// kernel (LDS-staged 16-channel chunks, a barrier per chunk, 112-pixel tiles) runs at 3.2-3.7 TB/s (31 us); this one takes 23-25 us,
// and the heat-map layer 11.5 against 20.  (The 256 -> 64 reductions stay on the generic kernel: with K = 256 a wave's weights are
// 64 KB against 32 KB of input -- 16 channels per wave read the input four times (61 us), 64 channels with the weights re-loaded per
// chunk of 64 channels 36 us, against 28.  64 -> 64 and 128 -> 64 measure within 1 us of the generic kernel and stay on it.)  Here:
//   wave  = a run of 14 (or 2: launches of few waves) tiles of 16 consecutive pixels of one frame x NT blocks of 16 output channels; its
//           weights (K / 4 k-steps x NT fragments = 64 registers) are loaded once and stay in registers;
//   A[row = pixel l15][k = lq] of k-step s = in[channel 4 s + lq][pixel]: NCHW keeps a channel's pixels contiguous, so a k-step's
//           operand is one dword per lane (16 lanes = 64 contiguous bytes per channel), the next tile's requested under this tile's MFMAs;
//   B[k = lq][col = l15] = folded weight of output channel 16 nt + l15;
//   D[row = 4 lq + r][col = l15]: four consecutive pixels of one output channel per lane -> bias, residual and ReLU on 16-byte
//           vectors, one 16-byte store per channel block.  No LDS, no barrier.
#include "kernels.h"

namespace grk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {


template <int KC, int NT, int RUN /* tiles per wave */, bool MASK /* the last channel block is partial */>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_pw_f32(const ConvArgs a) {
    constexpr int run = RUN;
    constexpr int KS = KC / 4;
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
    const int HW = a.H * a.W, runs = HW / (16 * run), groups = (a.Cout + 16 * NT - 1) / (16 * NT);
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= a.N * runs * groups) return;
    // consecutive waves share the pixels and differ in the channel group: the input run is read from HBM once and from L2 after that
    const int g = w % groups, t = w / groups, n = t / runs, px0 = (t - n * runs) * 16 * run, co0 = g * 16 * NT;
    float bw[KS][NT];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bw[s][nt] = a.w[(size_t)(4 * s + lq) * a.CoutPad + co0 + nt * 16 + l15];
    float bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = a.bias[co0 + nt * 16 + l15];      // [CoutPad]: zero beyond Cout
    const float* inp = a.in + ((size_t)n * a.in_ctot + a.in_coff + lq) * HW + px0 + l15;            // channel lq, this lane's pixel of tile 0
    float* outp = a.out + ((size_t)n * a.out_ctot + a.out_coff + co0 + l15) * HW + px0 + 4 * lq;       // channel co0 + l15, pixels 4 lq ..
    const float* addp = a.n_add ? a.add[0] + ((size_t)n * a.add_ctot[0] + a.add_coff[0] + co0 + l15) * HW + px0 + 4 * lq : nullptr;
    const size_t cstr = (size_t)16 * HW;
    float cur[KS], nxt[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) cur[s] = inp[(size_t)(4 * s) * HW];
    for (int tile = 0; tile < run; ++tile) {
        const int tn = tile + 1 < run ? tile + 1 : tile;     // the last tile re-requests itself
        f32x4 res[NT];
        if (addp) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                if (!MASK || co0 + nt * 16 + l15 < a.Cout) res[nt] = *reinterpret_cast<const f32x4*>(addp + nt * cstr + 16 * tile);
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
            if (!MASK || co0 + nt * 16 + l15 < a.Cout) *reinterpret_cast<f32x4*>(outp + nt * cstr + 16 * tile) = v;
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
    }
}

}  // namespace

bool conv_pw_eligible(int cin, int cout, int ks, int stride, int h, int w, int n_add) {
    return ks == 1 && stride == 1 && n_add <= 1 && (h * w) % (16 * 14) == 0 && ((cin == 64 && cout >= 17) || (cin == 128 && cout >= 17 && cout <= 32));      // 128 -> 64 would need 128 weight registers + operands: spills
}

// a.w: the direct kernels' packing [CinPad][CoutPad] (one tap)
hipError_t launch_conv_pw(ConvArgs a, hipStream_t s) {
    if (!conv_pw_eligible(a.Cin, a.Cout, a.ks, a.stride, a.H, a.W, a.n_add) || (a.n_add == 1 && a.add_shift[0] != 0) || a.CinPad < a.Cin)
        return hipErrorInvalidValue;
    const int nt = a.Cout > 32 ? 4 : 2, groups = (a.Cout + 16 * nt - 1) / (16 * nt), tiles = a.H * a.W / 16;
    if (a.CoutPad < groups * 16 * nt) return hipErrorInvalidValue;
    // tiles per wave: 14 (224 pixels) where that still leaves about a wave per SIMD, else 2 (a wave's weights are 16-32 KB: short runs pay
    // for them again and again, long runs of a 224-wave launch leave the chip empty).  Compile-time: the tile loop is software-pipelined
    // by the compiler only with a constant trip count (64 -> 256 at 16 frames: 23 us against 29 with a run-time bound).
    if (tiles % 14 != 0 || tiles % 2 != 0) return hipErrorInvalidValue;
    const bool lng = (long)a.N * (tiles / 14) * groups >= 800;
    const dim3 grid((a.N * (tiles / (lng ? 14 : 2)) * groups + 3) / 4);
    const bool mask = a.Cout % (16 * nt) != 0;
#define GRK_PW2(KC, NT, RUN) (mask ? launch_k(conv_pw_f32<KC, NT, RUN, true>, grid, dim3(256), 0, s, a) : launch_k(conv_pw_f32<KC, NT, RUN, false>, grid, dim3(256), 0, s, a))
#define GRK_PW(KC, NT) (lng ? GRK_PW2(KC, NT, 14) : GRK_PW2(KC, NT, 2))
    if (a.Cin == 64) return nt == 4 ? GRK_PW(64, 4) : GRK_PW(64, 2);
    return GRK_PW(128, 2);
#undef GRK_PW2
#undef GRK_PW
}

}  // namespace grk
