// bf16 path, round 5: a whole CHAIN of 3x3 stride-1 BasicBlocks of one HR branch (hrnet.py:30-59: conv-BN-ReLU, conv-BN, + x, ReLU;
// 4 blocks = 8 convolutions per branch and module, hrnet.py:141-187) in ONE launch with the frame resident in LDS.
//
// Why: at 256 frames the 64 ch @28x28 / 128 ch @14x14 / 256 ch @7x7 branch convolutions ran 30-31 us per launch at 0.19 of the bf16 matrix
// peak (MFMA-busy 0.24-0.27, SQ_WAIT_ANY 0.45-0.69: profiles/r04_bf16_n256_*): every launch pays a workgroup prologue, an HBM round trip
// for ~2 us of MFMAs per workgroup, an LDS epilogue and a launch boundary, 144 times per step.  All four branches of a module execute the
// SAME 57.8 MFLOP per frame, and a call of 256 frames has exactly one frame per CU.  So: workgroup = one frame, 8 waves; the frame's
// activations (50-100 KB in bf16) stay in LDS from the first convolution of the chain to the last, the weights stream L2 -> registers
// (A fragments, two k-steps ahead), HBM sees the chain's input once and its output once.
//
// LDS image: the zero-padded plane, flattened.  Slot (y, x) = (y + 1) * P + (x + 1) with row pitch P = W + 1: the right halo of row y IS the
// left halo of row y + 1 (one shared zero column), rows -1 and H are zero rows.  A slot holds the pixel's C channels (2C bytes) + 32 bytes of
// padding: slot stride = 2C + 32 bytes = 16 * m with m / 2 odd, which makes every 16-lane group of a ds_read_b128 (lanes with 16 different
// pixels, half of them on k-group q, half on q + 1: MI355X_MICROARCH, LDS table) hit 16 different 16-byte bank groups: conflict-free.
// With the flattened image a tap is a CONSTANT slot offset (dy * P + dx) for every output, so a wave's MFMA column tile is simply 16
// consecutive slots and all operand addresses are one base register + immediates.  Outputs that fall on the halo column (and past the
// plane) are computed and dropped: 811 of 816 (28x28), 209 of 224 (14x14), 55 of 64 (7x7) columns are real.
//
// Roles (as conv_bf16.hip): A[cout l&15][k = 8(l>>4)+j] = W[tap][cout][cin], B[k][pixel l&15] = slot[pixel + tap][cin], D[cout 4(l>>4)+r][pixel l&15].
// Wave (cb, pg) owns CS x 16 output channels x PS column tiles; per k-step (one tap x 32 input channels): CS weight fragments (global, 16 B
// per lane, prefetched two steps ahead in a ring of three register sets), PS pixel fragments (ds_read_b128), CS x PS MFMAs.
//
// In place: a convolution's outputs are all held in accumulators until every wave has finished READING the plane (barrier), then written
// over it (bias, ReLU, bf16, 8 bytes per lane and tile), barrier, next convolution.  The residual of a BasicBlock costs no registers: when
// conv1's result t replaces x in a lane's slots, the lane first reads x from those very slots and seeds conv2's accumulators with
// x + bias2 -- conv2 then ends with relu(acc).  Halo slots are never written, so they stay zero for the whole chain.
//
// Parity: every intermediate is rounded to bf16 exactly where the unfused path stores it, so the chain equals the launch-per-convolution
// path up to fp32 summation order (output-rounding ties); tests/test_gpu_round5.py compares both and the fp32 oracle on bf16-rounded operands.
#include "kernels.h"

#include <cstdlib>

namespace grk {

#define GRK_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u16 f2bf_c(float f) {           // round to nearest even (finite inputs), as conv_bf16.hip
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__device__ __forceinline__ unsigned pack2_c(float lo, float hi) { return (unsigned)f2bf_c(lo) | ((unsigned)f2bf_c(hi) << 16); }
__device__ __forceinline__ float bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

template <int C, int W>
struct ChainGeom {
    static constexpr int P = W + 1;                         // row pitch in slots
    static constexpr int SB = 2 * C + 32;                   // slot stride, bytes
    static constexpr int O0 = P + 1;                        // slot of pixel (0, 0) = first output column
    static constexpr int NOUT = W * P - 1;                  // output columns o0 .. slot of pixel (W-1, W-1)
    static constexpr int CS = 2;                            // 16-channel blocks per wave
    static constexpr int WCB = C / (16 * CS);               // waves along the output channels
    static constexpr int WPG = 8 / WCB;                     // waves along the pixels
    static constexpr int PS = ((NOUT + 15) / 16 + WPG - 1) / WPG;      // column tiles per wave
    static constexpr int NT = WPG * PS;                     // column tiles
    static constexpr int NSLOT = O0 + NT * 16 + P + 2;      // highest slot a tap reads: O0 + 16 NT - 1 + P + 1; + one spare slot (the read-ahead of a convolution's last step)
    static constexpr int LDS = NSLOT * SB;
    static constexpr int NS = 9 * (C / 32);                 // k-steps per convolution
    static constexpr int UPP = C / 8;                       // 16-byte units per pixel
    static constexpr int NU = (W * W * UPP + 511) / 512;    // units per thread of the plane
    static_assert(C % 32 == 0 && WCB >= 1 && WCB <= 8 && 8 % WCB == 0, "wave grid");
    static_assert((SB / 16) % 2 == 0 && ((SB / 32) % 2) == 1, "slot stride must be 32 * odd bytes (conflict-free b128 reads)");
    static_assert(NS % 3 == 0, "the weight ring has three register sets");
    static_assert(LDS <= 160 * 1024, "the plane must fit the LDS");
    static_assert((PS - 1) * 16 * SB + (2 * P + 2) * SB + (C / 32) * 64 < 65536, "ds_read immediates");
};

template <int C, int W>
__global__ __launch_bounds__(512) void conv_bf16_chain(const ChainArgs a) {
    typedef ChainGeom<C, W> G;
    constexpr int P = G::P, SB = G::SB, CS = G::CS, PS = G::PS, UPP = G::UPP, NU = G::NU, NCH = C / 32;
    extern __shared__ __align__(16) unsigned char plane[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave % G::WCB, pg = wave / G::WCB;
    const int n = blockIdx.x;
    if (n >= a.N) return;

    // ---- the frame: HBM -> registers (all loads in flight), zero the plane meanwhile, then registers -> interior slots
    const u16* inb = reinterpret_cast<const u16*>(a.in) + (size_t)n * W * W * a.in_ctot + a.in_coff;
    u32x4 stage[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = tid + i * 512;
        stage[i] = u32x4{0u, 0u, 0u, 0u};
        if (u < W * W * UPP) {
            const int px = u / UPP, part = u - px * UPP;
            stage[i] = *reinterpret_cast<const u32x4*>(inb + (size_t)px * a.in_ctot + part * 8);
        }
    }
    for (int u = tid; u < G::LDS / 16; u += 512) reinterpret_cast<u32x4*>(plane)[u] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = tid + i * 512;
        if (u < W * W * UPP) {
            const int px = u / UPP, part = u - px * UPP, y = px / W, x = px - y * W;
            *reinterpret_cast<u32x4*>(plane + ((y + 1) * P + x + 1) * SB + part * 16) = stage[i];
        }
    }

    // ---- per-lane constants
    const int o_first = G::O0 + pg * PS * 16 + l15;                          // this lane's output column of tile 0
    const unsigned char* bread = plane + (o_first - P - 1) * SB + lq * 16;    // B operand of tap (0,0), chunk 0, tile 0
    unsigned char* owrite = plane + o_first * SB + (cb * CS * 16 + lq * 4) * 2;      // this lane's 4 channels of tile 0, block 0
    unsigned valid = 0;                                                       // bit ps: column tile ps of this lane is a real pixel
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
        const int o = o_first + ps * 16;
        if (o % P != 0 && o <= W * P + W) valid |= 1u << ps;
    }
    const int wl = ((cb * CS * 16 + l15) * 32 + lq * 8);                     // element offset of this lane in a [C][32] weight row block
    const int co = cb * CS * 16 + lq * 4;                                     // first of this lane's 4 output channels (block 0)

    f32x4 acc[CS][PS];
    {
        const float* b0 = a.bias[0];
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(b0 + co + cs * 16);
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) acc[cs][ps] = bv;
        }
    }
    // weight fragments of k-step s: element offset (s * C + cs * 16) * 32 from the lane's base
    bf16x8 wr[3][CS];
    {
        const u16* w0 = reinterpret_cast<const u16*>(a.w[0]) + wl;
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            wr[0][cs] = *reinterpret_cast<const bf16x8*>(w0 + (0 * C + cs * 16) * 32);
            wr[1][cs] = *reinterpret_cast<const bf16x8*>(w0 + (1 * C + cs * 16) * 32);
        }
    }
    __syncthreads();                                                          // the plane is staged

    for (int ci = 0; ci < a.nconv; ++ci) {
        const u16* wc = reinterpret_cast<const u16*>(a.w[ci]) + wl;
        const int cn = ci + 1 < a.nconv ? ci + 1 : ci;                        // the last convolution re-requests itself (nobody waits for it)
        const u16* wn = reinterpret_cast<const u16*>(a.w[cn]) + wl;
        // Per k-step (tap x 32-channel chunk): CS weight fragments requested two steps ahead (ring of three register sets), and per column tile
        // one pixel fragment: tile ps of step s + 1 is requested right behind the MFMAs of tile ps of step s, into the register set they have
        // just read -- a read has PS - 1 MFMA pairs (and the SIMD's other wave) to land.  Left to itself hipcc sinks every read and every
        // weight load to its first use (one register set, lgkmcnt(0) in front of every MFMA pair, vmcnt(0) per step: the loop ran at LDS
        // latency); the sched_barrier behind every group pins the order written here.  The chunk loop stays a loop (9 taps unrolled: the
        // ring positions are static, 9 = 3 x 3), so every address is a per-chunk base + immediates.
        bf16x8 bfr[PS];
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) bfr[ps] = *reinterpret_cast<const bf16x8*>(bread + ps * 16 * SB);
#pragma unroll 1
        for (int chunk = 0; chunk < NCH; ++chunk) {
            const unsigned char* bch = bread + chunk * 64;
            const u16* wch = wc + (size_t)chunk * 9 * C * 32;
            const bool lastc = chunk == NCH - 1;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                {                                                             // weights of step s + 2 (taps 9, 10 = taps 0, 1 of the next chunk / of the next convolution)
                    const u16* src = wch + (size_t)(tap + 2) * C * 32;
                    if (tap >= 7) src = lastc ? wn + (size_t)(tap - 7) * C * 32 : src;
#pragma unroll
                    for (int cs = 0; cs < CS; ++cs) wr[(tap + 2) % 3][cs] = *reinterpret_cast<const bf16x8*>(src + cs * 16 * 32);
                }
                // the next step's pixel fragments: tap + 1 of this chunk, or tap 0 of the next (the last step of a convolution reads ahead into
                // the slot padding / the spare slot: nobody uses those values)
                const int noff = tap < 8 ? (((tap + 1) / 3) * P + ((tap + 1) % 3)) * SB : 64;
#pragma unroll
                for (int ps = 0; ps < PS; ++ps) {
#pragma unroll
                    for (int cs = 0; cs < CS; ++cs) acc[cs][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[tap % 3][cs], bfr[ps], acc[cs][ps], 0, 0, 0);
                    bfr[ps] = *reinterpret_cast<const bf16x8*>(bch + ps * 16 * SB + noff);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // ---- in-place epilogue
        const bool first = (ci & 1) == 0;                                     // conv1 of a BasicBlock: the plane still holds the block's input x
        f32x4 bnext[CS];
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) bnext[cs] = *reinterpret_cast<const f32x4*>(a.bias[cn] + co + cs * 16);
        __syncthreads();                                                      // every wave has read what it needs of the plane
#pragma unroll
        for (int ps = 0; ps < PS; ++ps)
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                unsigned char* pos = owrite + ps * 16 * SB + cs * 32;
                u32x2 r = u32x2{0u, 0u};
                if (first) r = *reinterpret_cast<const u32x2*>(pos);
                const f32x4 v = acc[cs][ps];
                const u32x2 pk = u32x2{pack2_c(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)), pack2_c(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f))};
                if (valid & (1u << ps)) *reinterpret_cast<u32x2*>(pos) = pk;
                f32x4 nx = bnext[cs];
                if (first) { nx[0] += bf_lo(r[0]); nx[1] += bf_hi(r[0]); nx[2] += bf_lo(r[1]); nx[3] += bf_hi(r[1]); }
                acc[cs][ps] = nx;
            }
        __syncthreads();                                                      // the plane holds this convolution's output
    }

    // ---- the chain's output: interior slots -> HBM, 16 bytes per lane, pixels in memory order
    u16* outb = reinterpret_cast<u16*>(a.out) + (size_t)n * W * W * a.out_ctot + a.out_coff;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = tid + i * 512;
        if (u < W * W * UPP) {
            const int px = u / UPP, part = u - px * UPP, y = px / W, x = px - y * W;
            *reinterpret_cast<u32x4*>(outb + (size_t)px * a.out_ctot + part * 8) = *reinterpret_cast<const u32x4*>(plane + ((y + 1) * P + x + 1) * SB + part * 16);
        }
    }
}

template <int C, int W>
hipError_t set_chain_lds() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_chain<C, W>), hipFuncAttributeMaxDynamicSharedMemorySize, ChainGeom<C, W>::LDS);
}

}  // namespace

hipError_t conv_bf16_chain_init() {
    GRK_TRY((set_chain_lds<64, 28>()));
    GRK_TRY((set_chain_lds<128, 14>()));
    GRK_TRY((set_chain_lds<256, 7>()));
    return hipSuccess;
}

bool conv_bf16_chain_eligible(int c, int w) { return (c == 64 && w == 28) || (c == 128 && w == 14) || (c == 256 && w == 7); }

// a.in / a.out: NHWC bf16 views of (N, W, W, C) tensors (channel strides in_ctot / out_ctot, first channels in_coff / out_coff, multiples of 8);
// a.w[i]: [C/32][9][C][32] bf16 (pack_conv's bf16 layout with CoutPad = C), a.bias[i]: fp32 [C]; nconv even, <= kMaxChain:
// convolutions 2k, 2k+1 are conv1 / conv2 of BasicBlock k.
hipError_t launch_conv_bf16_chain(const ChainArgs& a, int c, int w, hipStream_t s) {
    if (!conv_bf16_chain_eligible(c, w) || a.nconv < 2 || a.nconv > kMaxChain || (a.nconv & 1) || a.N < 1) return hipErrorInvalidValue;
    if (a.in_ctot % 8 != 0 || a.in_coff % 8 != 0 || a.out_ctot % 8 != 0 || a.out_coff % 8 != 0) return hipErrorInvalidValue;
    if (c == 64) return launch_k(conv_bf16_chain<64, 28>, dim3(a.N), dim3(512), ChainGeom<64, 28>::LDS, s, a);
    if (c == 128) return launch_k(conv_bf16_chain<128, 14>, dim3(a.N), dim3(512), ChainGeom<128, 14>::LDS, s, a);
    return launch_k(conv_bf16_chain<256, 7>, dim3(a.N), dim3(512), ChainGeom<256, 7>::LDS, s, a);
}

}  // namespace grk
