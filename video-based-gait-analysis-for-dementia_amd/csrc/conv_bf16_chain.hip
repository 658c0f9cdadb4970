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

#include <cstdio>
#include <cstdlib>

namespace grk {

#define GRK_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// two floats -> two bf16 (round to nearest even) in ONE instruction: v_cvt_pk_bf16_f32.  The integer form ((u + 0x7fff + (u >> 16 & 1)) >> 16, five
// vector instructions per value) made the in-place epilogue -- 4 values x CS x PS tiles per wave and convolution -- cost a third of a k-loop.
typedef __bf16 bf16x2_c __attribute__((ext_vector_type(2)));
typedef float f32x2_c __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_c(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_c{lo, hi}, bf16x2_c)); }
// relu on the bits: negative floats (and -0) are negative integers; one v_max_i32, where fmaxf(x, 0) costs a canonicalising v_max first
__device__ __forceinline__ float relu_c(float x) { const int i = __float_as_int(x); return __int_as_float(i > 0 ? i : 0); }
typedef short s16x2_c __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_pk(unsigned v) {      // max(x, 0) on two packed bf16: v_pk_max_i16
    const s16x2_c a = __builtin_bit_cast(s16x2_c, v);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(a, s16x2_c{0, 0}));
}
__device__ __forceinline__ float bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

// F frames per workgroup (round 6; F = 2 for the 256-channel 7x7 chain): the frames are stacked in ONE flattened plane, a zero row between them (plane rows 1 .. W:
// frame 0, W + 1: zero, W + 2 .. 2 W + 1: frame 1), so a workgroup's weight stream -- 9.4 MB per chain from L2, the bound of that chain: two single-frame workgroups
// per CU pulled 46 B/clk of the CU's 64 -- serves two frames.  The separator row is never written (the column mask below), so it stays the zero padding of both.
template <int C, int W, int F = 1>
struct ChainGeom {
    static constexpr int P = W + 1;                         // row pitch in slots
    static constexpr int SB = 2 * C + 32;                   // slot stride, bytes
    static constexpr int O0 = P + 1;                        // slot of pixel (0, 0) = first output column
    static constexpr int H = F * W + F - 1;                 // plane rows that carry outputs (the separator rows included)
    static constexpr int NOUT = H * P - 1;                  // output columns o0 .. slot of pixel (W-1, W-1) of the last frame
    static constexpr int CS = 2;                            // 16-channel blocks per wave
    static constexpr int WCB = C / (16 * CS);               // waves along the output channels
    static constexpr int WPG = 8 / WCB;                     // waves along the pixels
    static constexpr int PS = ((NOUT + 15) / 16 + WPG - 1) / WPG;      // column tiles per wave
    static constexpr int NT = WPG * PS;                     // column tiles
    static constexpr int NSLOT = O0 + NT * 16 + P + 2;      // highest slot a tap reads: O0 + 16 NT - 1 + P + 1; + one spare slot (the read-ahead of a convolution's last step)
    static constexpr int LDS = NSLOT * SB;
    static constexpr int NS = 9 * (C / 32);                 // k-steps per convolution
    static constexpr int UPP = C / 8;                       // 16-byte units per pixel
    static constexpr int NU = (F * W * W * UPP + 511) / 512;    // units per thread of the plane
    static_assert(C % 32 == 0 && WCB >= 1 && WCB <= 8 && 8 % WCB == 0, "wave grid");
    static_assert((SB / 16) % 2 == 0 && ((SB / 32) % 2) == 1, "slot stride must be 32 * odd bytes (conflict-free b128 reads)");
    static_assert(NS % 3 == 0, "the weight ring has three register sets");
    static_assert(LDS <= 160 * 1024, "the plane must fit the LDS");
    static_assert(F > 1 || (PS - 1) * 16 * SB + (2 * P + 2) * SB + (C / 32) * 64 < 65536, "ds_read immediates");      // (F = 2: 70 KB of plane, hipcc keeps a second base register)
};

// One convolution's k-loop over the LDS plane, shared by the frame-resident chain and the band-resident block kernel.
// Per k-step (tap x 32-channel chunk): CS weight fragments requested two steps ahead (ring of three register sets), and per column tile
// one pixel fragment: tile ps of step s + 1 is requested right behind the MFMAs of tile ps of step s, into the register set they have
// just read -- a read has PS - 1 MFMA pairs (and the SIMD's other wave) to land.  Left to itself hipcc sinks every read and every
// weight load to its first use (one register set, lgkmcnt(0) in front of every MFMA pair, vmcnt(0) per step: the loop ran at LDS
// latency); the sched_barrier behind every group pins the order written here.  The chunk loop stays a loop (9 taps unrolled: the
// ring positions are static, 9 = 3 x 3), so every address is a per-chunk base + immediates.
// bread: B operand of tap (0,0), chunk 0, tile 0 of this lane; wc / wn: the weights of this / the next convolution (UNIFORM pointers: with the
// lane's share kept apart as the 32-bit byte offset wlb every weight load is `global_load v, v_off, s[base]` -- a 64-bit per-lane pointer per
// k-step cost two registers each, which hipcc hoisted out of the band loop of the frame kernel and spilled); wr[0], wr[1] hold steps 0, 1 on
// entry and the next convolution's on exit.
// RING: register sets of the weight ring = prefetch distance + 1.  3 (two k-steps ahead) where a k-step is >= 14 MFMAs; 9 (eight ahead) for the
// 256-channel 7x7 chain, whose k-steps are 8 MFMAs = 128 cycles: two steps did not cover an L2 round trip (SQ_WAIT_ANY 0.73 of its wave cycles).
// RING must divide the 9 taps of a chunk (the ring positions are static in the unrolled tap loop).
template <int C, int P, int SB, int CS, int PS, int RING = 3>
__device__ __forceinline__ void chain_kloop(f32x4 (&acc)[CS][PS], bf16x8 (&wr)[RING][CS], const unsigned char* bread, const u16* wc, const u16* wn, unsigned wlb) {
    constexpr int NCH = C / 32, D = RING - 1;
    static_assert(9 % RING == 0, "the ring must divide the taps");
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
            {                                                             // weights of step s + D (taps 9 .. = the next chunk's / the next convolution's first ones)
                const u16* src = wch + (size_t)(tap + D) * C * 32;
                if (tap + D >= 9) src = lastc ? wn + (size_t)(tap + D - 9) * C * 32 : src;
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) wr[(tap + D) % RING][cs] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(src + cs * 16 * 32) + wlb);
            }
            // the next step's pixel fragments: tap + 1 of this chunk, or tap 0 of the next (the last step of a convolution reads ahead into
            // the slot padding / the spare slot: nobody uses those values)
            const int noff = tap < 8 ? (((tap + 1) / 3) * P + ((tap + 1) % 3)) * SB : 64;
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) {
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) acc[cs][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[tap % RING][cs], bfr[ps], acc[cs][ps], 0, 0, 0);
                bfr[ps] = *reinterpret_cast<const bf16x8*>(bch + ps * 16 * SB + noff);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// Registers: the 256-channel 7x7 chain (L2-bound on its weight stream, MFMA-busy 0.41) is held to 96 (five waves per SIMD; 32 bytes of scratch): two of its workgroups fit
// a CU -- 158 -> 147 us per chain alone at 256 frames -- and one fits BESIDE a workgroup of the 128-channel 14x14 chain (2 x 96 + 2 x 160 registers, 45 + 74 KB of LDS), which
// is launched on another lane at the same time.  The step did not show the latter (10.59-10.68 ms either way, three pairs on one box).
template <int C, int W, int F = 1>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(C == 256 && F == 1 ? 5 : 2))) void conv_bf16_chain(const ChainArgs a) {
    typedef ChainGeom<C, W, F> G;
    constexpr int P = G::P, SB = G::SB, CS = G::CS, PS = G::PS, UPP = G::UPP, NU = G::NU;
    extern __shared__ __align__(16) unsigned char plane[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave % G::WCB, pg = wave / G::WCB;
    const int n = blockIdx.x * F;                            // first frame of this workgroup; frames n .. n + nf - 1 exist
    if (n >= a.N) return;
    const int nf = a.N - n < F ? a.N - n : F;

    // ---- the frames: HBM -> registers (all loads in flight), zero the plane meanwhile, then registers -> interior slots (frame f: plane rows f (W + 1) + 1 ..)
    const u16* inb = reinterpret_cast<const u16*>(a.in) + (size_t)n * W * W * a.in_ctot + a.in_coff;
    u32x4 stage[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = tid + i * 512;
        stage[i] = u32x4{0u, 0u, 0u, 0u};
        if (u < nf * W * W * UPP) {
            const int px = u / UPP, part = u - px * UPP;      // px runs over the frames: they are contiguous in memory
            stage[i] = *reinterpret_cast<const u32x4*>(inb + (size_t)px * a.in_ctot + part * 8);
        }
    }
    for (int u = tid; u < G::LDS / 16; u += 512) reinterpret_cast<u32x4*>(plane)[u] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = tid + i * 512;
        if (u < nf * W * W * UPP) {
            const int px = u / UPP, part = u - px * UPP, f = px / (W * W), pf = px - f * W * W, y = pf / W, x = pf - y * W;
            *reinterpret_cast<u32x4*>(plane + ((f * (W + 1) + y + 1) * P + x + 1) * SB + part * 16) = stage[i];
        }
    }

    // ---- per-lane constants
    const int o_first = G::O0 + pg * PS * 16 + l15;                          // this lane's output column of tile 0
    const unsigned char* bread = plane + (o_first - P - 1) * SB + lq * 16;    // B operand of tap (0,0), chunk 0, tile 0
    unsigned char* owrite = plane + o_first * SB + (cb * CS * 16 + lq * 4) * 2;      // this lane's 4 channels of tile 0, block 0
    unsigned valid = 0;                                                       // bit ps: column tile ps of this lane is a real pixel
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
        const int o = o_first + ps * 16, row = o / P;      // plane row 1 .. H; rows that are a multiple of W + 1 separate two frames
        if (o % P != 0 && o <= G::H * P + W && row % (W + 1) != 0 && (row - 1) / (W + 1) < nf) valid |= 1u << ps;
    }
    const unsigned wlb = ((cb * CS * 16 + l15) * 32 + lq * 8) * 2;           // byte offset of this lane in a [C][32] weight row block
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
    // weight fragments of k-step s: element offset (s * C + cs * 16) * 32 from the lane's base; the ring starts with steps 0 .. RING - 2
    constexpr int RING = 3;                                 // (9 for the 256-channel chain, k-steps of 8 MFMAs: 150 us against 145 -- its weight stream is bound by the CU's 64 B/clk vector-memory path, not by latency)
    bf16x8 wr[RING][CS];
    {
        const unsigned char* w0 = reinterpret_cast<const unsigned char*>(a.w[0]);
#pragma unroll
        for (int st = 0; st < RING - 1; ++st)
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) wr[st][cs] = *reinterpret_cast<const bf16x8*>(w0 + (st * C + cs * 16) * 64 + wlb);
    }
    __syncthreads();                                                          // the plane is staged

    for (int ci = 0; ci < a.nconv; ++ci) {
        const u16* wc = reinterpret_cast<const u16*>(a.w[ci]);
        const int cn = ci + 1 < a.nconv ? ci + 1 : ci;                        // the last convolution re-requests itself (nobody waits for it)
        const u16* wn = reinterpret_cast<const u16*>(a.w[cn]);
        chain_kloop<C, P, SB, CS, PS, RING>(acc, wr, bread, wc, wn, wlb);
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
                const u32x2 pk = u32x2{pack2_c(relu_c(v[0]), relu_c(v[1])), pack2_c(relu_c(v[2]), relu_c(v[3]))};
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
        if (u < nf * W * W * UPP) {
            const int px = u / UPP, part = u - px * UPP, f = px / (W * W), pf = px - f * W * W, y = pf / W, x = pf - y * W;
            *reinterpret_cast<u32x4*>(outb + (size_t)px * a.out_ctot + part * 8) = *reinterpret_cast<const u32x4*>(plane + ((f * (W + 1) + y + 1) * P + x + 1) * SB + part * 16);
        }
    }
}


// ---- ONE BasicBlock of the 32-channel 56x56 branch with a BAND of the frame resident in LDS (a whole 56x56x32 frame is 200 KB: it does not fit).
// The launch-per-convolution kernel (conv_bf16_direct) runs these layers at 37.9 us = 0.16 of the matrix peak at 256 frames: they move 2-3
// tensors of 51 MB each per launch.  Here a workgroup owns R = 19 output rows of one frame (3 bands per frame = 768 workgroups = 3 per CU):
// input rows y0 - 2 .. y0 + R + 1 (rows outside the image are zero) -> LDS in the flattened, pitch-57 image of the chain kernel; conv1 on
// rows y0 - 1 .. y0 + R (its rows outside the image are conv2's zero padding: not written, the slots keep their zeros), in place; conv2 on
// the same R + 2 rows (the two outer ones are dropped: 10 % of the MFMAs buy one tile map and the residual-in-place trick for both
// convolutions); rows y0 .. y0 + R - 1 -> HBM.  HBM sees 23/19 of the input once and the output once instead of five passes.
template <int C, int W, int R>
struct BandGeom {
    static constexpr int P = W + 1, SB = 2 * C + 32;
    static constexpr int ROWS = R + 4;                      // plane rows: image rows y0 - 2 .. y0 + R + 1
    static constexpr int O0 = P + 1;                        // plane row 1, x = 0
    static constexpr int NOUT = (R + 2) * P - 1;            // plane rows 1 .. R + 2
    static constexpr int CS = 2, WCB = C / 32, WPG = 8 / WCB;
    static constexpr int PS = ((NOUT + 15) / 16 + WPG - 1) / WPG, NT = WPG * PS;
    static constexpr int NSLOT = O0 + NT * 16 + P + 2;
    static constexpr int LDS = NSLOT * SB;
    static constexpr int UPP = C / 8;
    static constexpr int NUI = (ROWS * W * UPP + 511) / 512, NUO = (R * W * UPP + 511) / 512;
    static constexpr int NB = (W + R - 1) / R;              // bands per frame
    static_assert(((SB / 32) % 2) == 1 && LDS <= 160 * 1024 && PS <= 32 && NSLOT >= ROWS * P + 1, "band geometry");
    static_assert((PS - 1) * 16 * SB + (2 * P + 2) * SB + (C / 32) * 64 < 65536, "ds_read immediates");
};

template <int C, int W, int R, int WPE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void conv_bf16_block_band(const ChainArgs a) {
    typedef BandGeom<C, W, R> G;
    constexpr int P = G::P, SB = G::SB, CS = G::CS, PS = G::PS, UPP = G::UPP;
    extern __shared__ __align__(16) unsigned char plane[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave % G::WCB, pg = wave / G::WCB;
    const int n = blockIdx.x / G::NB, band = blockIdx.x - n * G::NB;
    if (n >= a.N) return;
    const int y0 = band * R;

    const u16* inb = reinterpret_cast<const u16*>(a.in) + (size_t)n * W * W * a.in_ctot + a.in_coff;
    u32x4 stage[G::NUI];
#pragma unroll
    for (int i = 0; i < G::NUI; ++i) {
        const int u = tid + i * 512, px = u / UPP, part = u - px * UPP, r = px / W, x = px - r * W, y = y0 - 2 + r;
        stage[i] = u32x4{0u, 0u, 0u, 0u};
        if (u < G::ROWS * W * UPP && y >= 0 && y < W) stage[i] = *reinterpret_cast<const u32x4*>(inb + ((size_t)y * W + x) * a.in_ctot + part * 8);
    }
    for (int u = tid; u < G::LDS / 16; u += 512) reinterpret_cast<u32x4*>(plane)[u] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
#pragma unroll
    for (int i = 0; i < G::NUI; ++i) {
        const int u = tid + i * 512, px = u / UPP, part = u - px * UPP, r = px / W, x = px - r * W;
        if (u < G::ROWS * W * UPP) *reinterpret_cast<u32x4*>(plane + (r * P + x + 1) * SB + part * 16) = stage[i];
    }

    const int o_first = G::O0 + pg * PS * 16 + l15;
    const unsigned char* bread = plane + (o_first - P - 1) * SB + lq * 16;
    unsigned char* owrite = plane + o_first * SB + (cb * CS * 16 + lq * 4) * 2;
    unsigned valid1 = 0, valid2 = 0;                          // bit ps: conv1 / conv2 write column tile ps of this lane
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
        const int o = o_first + ps * 16, r = o / P, y = y0 - 2 + r;
        const bool px_ok = o - r * P != 0 && r >= 1 && r <= R + 2 && y >= 0 && y < W;
        if (px_ok) valid1 |= 1u << ps;
        if (px_ok && r >= 2 && r <= R + 1) valid2 |= 1u << ps;
    }
    const unsigned wlb = ((cb * CS * 16 + l15) * 32 + lq * 8) * 2;
    const int co = cb * CS * 16 + lq * 4;

    f32x4 acc[CS][PS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias[0] + co + cs * 16);
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) acc[cs][ps] = bv;
    }
    bf16x8 wr[3][CS];
    const u16* w0 = reinterpret_cast<const u16*>(a.w[0]);
    const u16* w1 = reinterpret_cast<const u16*>(a.w[1]);
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        wr[0][cs] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(w0 + (0 * C + cs * 16) * 32) + wlb);
        wr[1][cs] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(w0 + (1 * C + cs * 16) * 32) + wlb);
    }
    __syncthreads();

    // conv1: t = relu(conv(x) + b1) replaces x (where t exists); conv2's accumulators start from x + b2
    chain_kloop<C, P, SB, CS, PS>(acc, wr, bread, w0, w1, wlb);
    f32x4 b2[CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) b2[cs] = *reinterpret_cast<const f32x4*>(a.bias[1] + co + cs * 16);
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < PS; ++ps)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            unsigned char* pos = owrite + ps * 16 * SB + cs * 32;
            const u32x2 r = *reinterpret_cast<const u32x2*>(pos);
            const f32x4 v = acc[cs][ps];
            if (valid1 & (1u << ps)) *reinterpret_cast<u32x2*>(pos) = u32x2{pack2_c(relu_c(v[0]), relu_c(v[1])), pack2_c(relu_c(v[2]), relu_c(v[3]))};
            f32x4 nx = b2[cs];
            nx[0] += bf_lo(r[0]); nx[1] += bf_hi(r[0]); nx[2] += bf_lo(r[1]); nx[3] += bf_hi(r[1]);
            acc[cs][ps] = nx;
        }
    __syncthreads();
    // conv2: y = relu(conv(t) + b2 + x), rows y0 .. y0 + R - 1
    chain_kloop<C, P, SB, CS, PS>(acc, wr, bread, w1, w1, wlb);
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < PS; ++ps)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            const f32x4 v = acc[cs][ps];
            if (valid2 & (1u << ps))
                *reinterpret_cast<u32x2*>(owrite + ps * 16 * SB + cs * 32) = u32x2{pack2_c(relu_c(v[0]), relu_c(v[1])), pack2_c(relu_c(v[2]), relu_c(v[3]))};
        }
    __syncthreads();
    u16* outb = reinterpret_cast<u16*>(a.out) + (size_t)n * W * W * a.out_ctot + a.out_coff;
#pragma unroll
    for (int i = 0; i < G::NUO; ++i) {
        const int u = tid + i * 512, px = u / UPP, part = u - px * UPP, r = px / W, x = px - r * W, y = y0 + r;
        if (u < R * W * UPP && y < W)
            *reinterpret_cast<u32x4*>(outb + ((size_t)y * W + x) * a.out_ctot + part * 8) = *reinterpret_cast<const u32x4*>(plane + ((r + 2) * P + x + 1) * SB + part * 16);
    }
}

// ---- The same block with the WORKGROUP = ONE FRAME walking its NB = 56 / R bands, the next band arriving by LDS-DMA under the current band's MFMAs.
// The band kernel above spends two thirds of a workgroup's life in its load and store phases (one workgroup per CU: nothing overlaps them; 51 us
// per BasicBlock at 256 frames against 11.8 us of MFMAs at peak and 22 us of HBM time).  A first persistent version kept the next band in
// registers (R = 14): 256 VGPRs + 160-284 bytes of scratch, and hipcc parked the prefetched rows in scratch right behind their loads, i.e.
// waited for them -- no overlap.  So the prefetch takes no registers at all:
//   * the next band's rows travel HBM -> LDS by LDS-DMA into a dense staging area (lane-linear, as the DMA writes) and are copied LDS -> LDS
//     into the padded plane when the current band has left;
//   * vmcnt is in order, and a wave that waits for a weight fragment would wait for every DMA it issued before: the weights of BOTH
//     convolutions (36 KB for C = 32) live in LDS for the whole launch, so the k-loops issue no vector-memory operation at all and the DMAs
//     fly through them.  Weight rows are 64 bytes (no room for padding): the 16-byte part p of row r sits at p ^ 2 (r >> 3 & 1), which puts the
//     16 lanes of every ds_read_b128 group (rows {0-3, 12-15} at part q, rows {4-11} at part q + 1, or the mirror image) on 16 different bank groups;
//   * barriers are raw s_barrier + lgkmcnt(0): __syncthreads() would drain the DMAs (a pending DMA is a pending LDS write to its fence).
// LDS: weights 36 KB + plane (R = 8: 12 rows, 72.7 KB) + staging (12 x 56 x 64 B = 43 KB) = 152 KB.  R = 8 tiles the 56 rows exactly (7 bands);
// 10 of 12 plane rows carry outputs of conv1, 8 of conv2's, 569 of 640 MFMA columns are real: 0.72 of the MFMAs are useful, HBM reads 1.5 x the input.
#define GRNET_GLOBAL_AS __attribute__((address_space(1)))
#define GRNET_LDS_AS __attribute__((address_space(3)))
__device__ __forceinline__ void dma16_c(const u16* src, unsigned char* lds_wave_base) {      // lane l's 16 bytes land at lds_wave_base + 16 l
    __builtin_amdgcn_global_load_lds((const GRNET_GLOBAL_AS void*)src, (GRNET_LDS_AS void*)lds_wave_base, 16, 0, 0);
}
// LDS-DMA pieces the compiler does not know of: lane l's 16 bytes land at lds + 16 l (M0 = LDS byte address of the piece).
// uniform base + 32-bit lane offset, only the lanes of `mask` (all lanes are on around it: the callers are in uniform control flow)
__device__ __forceinline__ void dma16_masked(unsigned off, const void* base, unsigned lds, unsigned long long mask) {
    asm volatile("s_mov_b32 m0, %2\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1" ::"v"(off), "s"(base), "s"(lds), "s"(mask) : "memory");
}
// all lanes (the weights).  One wait state between the M0 write and the DMA that reads it.
__device__ __forceinline__ void dma16_hidden_s(unsigned off, const void* base, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory");
}

__device__ __forceinline__ void lds_barrier() {               // every wave's LDS operations so far are done; vector-memory operations (DMAs, stores) stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// k-loop of one 32 -> 32 convolution with the weights in LDS: per tap CS weight fragments (one tap ahead, two register sets) and PS pixel fragments
// (ring as in chain_kloop); no vector-memory operation.  wl: this lane's fragment of tap 0, block 0 (swizzled part); 2 KB per tap, 1 KB per block.
template <int P, int SB, int CS, int PS>
__device__ __forceinline__ void chain_kloop_ldsw(f32x4 (&acc)[CS][PS], const unsigned char* bread, const unsigned char* wl) {
    bf16x8 bfr[PS], wr[2][CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) wr[0][cs] = *reinterpret_cast<const bf16x8*>(wl + cs * 1024);
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) bfr[ps] = *reinterpret_cast<const bf16x8*>(bread + ps * 16 * SB);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        if (tap < 8) {
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) wr[(tap + 1) & 1][cs] = *reinterpret_cast<const bf16x8*>(wl + (tap + 1) * 2048 + cs * 1024);
        }
        const int noff = (((tap + 1) / 3) * P + ((tap + 1) % 3)) * SB;
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) acc[cs][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[tap & 1][cs], bfr[ps], acc[cs][ps], 0, 0, 0);
            if (tap < 8) bfr[ps] = *reinterpret_cast<const bf16x8*>(bread + ps * 16 * SB + noff);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int W, int R>
struct FrameGeom {
    static constexpr int C = 32;
    typedef BandGeom<C, W, R> B;
    static constexpr int WBYTES = 2 * 9 * C * 64;            // both convolutions' weights, 64-byte rows
    static constexpr int PLANE = B::LDS;
    static constexpr int UR = W * B::UPP;                     // 16-byte units per row
    static constexpr int STAGE_UNITS = B::ROWS * UR;
    static constexpr int LDS = WBYTES + PLANE + STAGE_UNITS * 16;
    static constexpr int NUS = (STAGE_UNITS + 511) / 512, NUO = (R * UR + 511) / 512;
    static_assert(W % R == 0 && LDS <= 160 * 1024 && PLANE % 16 == 0, "frame geometry");
};

template <int W, int R>
__global__ __launch_bounds__(512) void conv_bf16_block_frame(const ChainArgs a) {
    constexpr int C = 32;
    typedef BandGeom<C, W, R> G;
    typedef FrameGeom<W, R> F;
    constexpr int P = G::P, SB = G::SB, CS = G::CS, PS = G::PS, UPP = G::UPP, NB = G::NB, UR = F::UR;
    static_assert(G::WCB == 1, "one channel block of 32");
    extern __shared__ __align__(16) unsigned char lds[];
    unsigned char* wlds = lds;                                 // [conv][tap][row 32][4 parts, swizzled] bf16
    unsigned char* plane = lds + F::WBYTES;
    unsigned char* stg = plane + F::PLANE;                     // [plane row][x][4 parts]: the NEXT band, as it lies in memory
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x;
    if (n >= a.N) return;
    const u16* inb = reinterpret_cast<const u16*>(a.in) + (size_t)n * W * W * a.in_ctot + a.in_coff;
    u16* outb = reinterpret_cast<u16*>(a.out) + (size_t)n * W * W * a.out_ctot + a.out_coff;

    // (round 5) unit i of this thread = (plane row r, column x, 16-byte part) is the same in every band: kept packed (r : 4 | x : 6 | part : 2 bits, two units per
    // register) instead of two divisions per unit and band in request() and again in deposit()
    unsigned upk[(F::NUS + 1) / 2];
#pragma unroll
    for (int i = 0; i < (F::NUS + 1) / 2; ++i) upk[i] = 0u;
#pragma unroll
    for (int i = 0; i < F::NUS; ++i) {
        const int u = i * 512 + tid, r = u / UR, q = u - r * UR, x = q / UPP, part = q - x * UPP;
        upk[i >> 1] |= (unsigned)((r << 8) | (x << 2) | part) << (16 * (i & 1));
    }
    auto request = [&](int y0) {                               // rows y0 - 2 .. y0 + R + 1 that exist -> staging, by LDS-DMA (rows outside the image are not requested)
#pragma unroll
        for (int i = 0; i < F::NUS; ++i) {
            const int ub = i * 512 + wave * 64, u = ub + lane;
            const unsigned pk = upk[i >> 1] >> (16 * (i & 1));
            const int r = (pk >> 8) & 15, x = (pk >> 2) & 63, part = pk & 3, y = y0 - 2 + r;
            if (u < F::STAGE_UNITS && y >= 0 && y < W) dma16_c(inb + ((size_t)y * W + x) * a.in_ctot + part * 8, stg + ub * 16);
        }
    };
    auto deposit = [&](int y0) {                               // staging -> interior slots of every plane row; zeros where the band hangs over the image
#pragma unroll
        for (int i = 0; i < F::NUS; ++i) {
            const int u = i * 512 + tid;
            const unsigned pk = upk[i >> 1] >> (16 * (i & 1));
            const int r = (pk >> 8) & 15, x = (pk >> 2) & 63, part = pk & 3, y = y0 - 2 + r;
            if (u < F::STAGE_UNITS) {
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if (y >= 0 && y < W) v = *reinterpret_cast<const u32x4*>(stg + u * 16);
                *reinterpret_cast<u32x4*>(plane + (r * P + x + 1) * SB + part * 16) = v;
            }
        }
    };
    request(0);
    {   // the weights: global -> registers -> LDS (swizzled); the plane's halo columns and spare slots: zero, once
        constexpr int WU = 2 * 9 * C * 4, NWU = (WU + 511) / 512;      // 16-byte units of both convolutions
        u32x4 wv[NWU];
#pragma unroll
        for (int i = 0; i < NWU; ++i) {
            const int v = i * 512 + tid, cv = v / (WU / 2), vv = v - cv * (WU / 2);
            wv[i] = u32x4{0u, 0u, 0u, 0u};
            if (v < WU) wv[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(a.w[cv]) + vv * 16);
        }
        for (int u = tid; u < F::PLANE / 16; u += 512) reinterpret_cast<u32x4*>(plane)[u] = u32x4{0u, 0u, 0u, 0u};
        // (round 6) the 32 output-channel rows of a tap are PERMUTED on the way in: channel c = 8 q + 4 s + r goes to row s * 16 + 4 q + r, so that lane (pixel, lq) of the
        // two MFMA blocks s = 0, 1 ends up with channels 8 lq .. 8 lq + 3 and 8 lq + 4 .. 8 lq + 7 -- EIGHT consecutive channels: the in-place epilogue, the conv2 seeds and
        // the output stores move 16 bytes per lane and tile instead of two times 8 (the stores were 8.7 of a launch's 50 us as 32-byte pieces: profiles/r06_block_frame_ablation.txt).
        // Every output's dot product is unchanged: bit-identical results.  (The same assignment in the frame-resident chain kernels, whose epilogues are a smaller share of
        // their launches, measured no change: 103.4 / 107.2 / 174 us per chain against 103.4 / 107.7 / 168.5 -- not kept there.)
#pragma unroll
        for (int i = 0; i < NWU; ++i) {
            const int v = i * 512 + tid, c = (v >> 2) & 31, part = v & 3;
            const int row = ((c >> 2) & 1) * 16 + 4 * (c >> 3) + (c & 3);
            if (v < WU) *reinterpret_cast<u32x4*>(wlds + ((v >> 7) * 32 + row) * 64 + ((part ^ (((row >> 3) & 1) << 1)) * 16)) = wv[i];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's DMAs of band 0 have landed
    lds_barrier();                                             // ... and everybody's; the plane is zero, the weights are in place
    deposit(0);
    lds_barrier();                                             // staging is free, the plane holds band 0
    const int o_first = G::O0 + wave * PS * 16 + l15;
    const unsigned char* bread = plane + (o_first - P - 1) * SB + lq * 16;
    unsigned char* owrite = plane + o_first * SB + lq * 16;       // this lane's 8 channels (8 lq ..) of column tile 0
    const unsigned char* wl0 = wlds + l15 * 64 + ((lq ^ (((l15 >> 3) & 1) << 1)) * 16);
    const unsigned char* wl1 = wl0 + 9 * C * 64;
    const int co = lq * 8;                                        // MFMA block cs holds channels co + 4 cs .. + 3 of this lane's pixel
    // conv2 needs rows 2 .. R + 1 of the plane only (conv1: rows 1 .. R + 2): its own, shorter tile run -- PS2 = 4 column tiles per wave from slot 2 P + 1 on
    // instead of conv1's PS = 5 from P + 1 on (a fifth of conv2's MFMAs and fragment reads computed rows nobody stores)
    constexpr int NOUT2 = R * P - 1, PS2 = ((NOUT2 + 15) / 16 + 7) / 8;
    const int o2_first = 2 * P + 1 + wave * PS2 * 16 + l15;
    const unsigned char* bread2 = plane + (o2_first - P - 1) * SB + lq * 16;
    f32x4 b1[CS], b2[CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        b1[cs] = *reinterpret_cast<const f32x4*>(a.bias[0] + co + cs * 4);
        b2[cs] = *reinterpret_cast<const f32x4*>(a.bias[1] + co + cs * 4);
    }
    if (NB > 1) request(R);                                    // behind the bias loads: waiting for those must not mean waiting for these
    const bool direct_store = !(a.flags & 1);

#pragma unroll 1
    for (int band = 0; band < NB; ++band) {
        const int y0 = band * R;
        unsigned valid1 = 0, valid2 = 0;
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const int o = o_first + ps * 16, r = o / P, y = y0 - 2 + r;
            const bool px_ok = o - r * P != 0 && r >= 1 && r <= R + 2 && y >= 0 && y < W;
            if (px_ok) valid1 |= 1u << ps;
            if (px_ok && r >= 2 && r <= R + 1) valid2 |= 1u << ps;
        }
        f32x4 acc[CS][PS];
#pragma unroll
        for (int cs = 0; cs < CS; ++cs)
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) acc[cs][ps] = b1[cs];
#ifdef GRNET_ABLATION
        if (!(a.flags & 16))
#endif
        chain_kloop_ldsw<P, SB, CS, PS>(acc, bread, wl0);
        if (direct_store) {
            // conv2's accumulators start from x + bias2, read at conv2's OWN columns while the plane still holds x (nobody has written yet)
            f32x4 acc2[CS][PS2];
            unsigned v2 = 0;
#pragma unroll
            for (int ps = 0; ps < PS2; ++ps) {
                const int o = o2_first + ps * 16, r = o / P, y = y0 - 2 + r;
                if (o - r * P != 0 && r >= 2 && r <= R + 1 && y >= 0 && y < W) v2 |= 1u << ps;
                const u32x4 rr = *reinterpret_cast<const u32x4*>(plane + (o2_first + ps * 16) * SB + lq * 16);
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) {
                    f32x4 nx = b2[cs];
                    nx[0] += bf_lo(rr[2 * cs]); nx[1] += bf_hi(rr[2 * cs]); nx[2] += bf_lo(rr[2 * cs + 1]); nx[3] += bf_hi(rr[2 * cs + 1]);
                    acc2[cs][ps] = nx;
                }
            }
            lds_barrier();                                     // every wave has read what it needs of x
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) {
                const f32x4 A = acc[0][ps], B = acc[1][ps];
                if (valid1 & (1u << ps)) *reinterpret_cast<u32x4*>(owrite + ps * 16 * SB) = u32x4{pack2_c(relu_c(A[0]), relu_c(A[1])), pack2_c(relu_c(A[2]), relu_c(A[3])), pack2_c(relu_c(B[0]), relu_c(B[1])), pack2_c(relu_c(B[2]), relu_c(B[3]))};
            }
            lds_barrier();
#ifdef GRNET_ABLATION
            if (!(a.flags & 32))
#endif
            chain_kloop_ldsw<P, SB, CS, PS2>(acc2, bread2, wl1);
            // the block's output rows leave straight from the accumulators (round 5: was in place through the plane, a barrier, then 16-byte stores): 8 bytes per lane, the
            // four k-groups of a pixel make 32 contiguous bytes, the two channel blocks its 64-byte row; nothing of this band's output is needed in LDS again.
            // The next band's DMAs (issued a whole band ago) and the previous band's stores are the only vector-memory operations in flight: waited for HERE, in front of
            // this band's stores (round-5 advice: a counted wait behind them assumed CS x PS2 stores per wave, but the stores are predicated -- the wave that owns the
            // column tiles past the band issues fewer, and its DMAs could then still be in flight when deposit() reads the staging area)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int ps = 0; ps < PS2; ++ps) {
                int of = o2_first;
                asm volatile("" : "+v"(of));                   // (tile-independent pixel offsets: not to be hoisted out of the band loop into 2 x PS registers)
                const int o = of + ps * 16, r = o / P, x = o - r * P - 1;
                u16* op = outb + ((size_t)(y0 + r - 2) * W + x) * a.out_ctot + lq * 8;
                const f32x4 A = acc2[0][ps], B = acc2[1][ps];
                if (v2 & (1u << ps)) *reinterpret_cast<u32x4*>(op) = u32x4{pack2_c(relu_c(A[0]), relu_c(A[1])), pack2_c(relu_c(A[2]), relu_c(A[3])), pack2_c(relu_c(B[0]), relu_c(B[1])), pack2_c(relu_c(B[2]), relu_c(B[3]))};       // 16 bytes per lane: a pixel's 64-byte row by four lanes
            }
        } else {
        lds_barrier();
#pragma unroll
        for (int ps = 0; ps < PS; ++ps)
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                unsigned char* pos = owrite + ps * 16 * SB + cs * 8;
                const u32x2 r = *reinterpret_cast<const u32x2*>(pos);
                const f32x4 v = acc[cs][ps];
                if (valid1 & (1u << ps)) *reinterpret_cast<u32x2*>(pos) = u32x2{pack2_c(relu_c(v[0]), relu_c(v[1])), pack2_c(relu_c(v[2]), relu_c(v[3]))};
                f32x4 nx = b2[cs];
                nx[0] += bf_lo(r[0]); nx[1] += bf_hi(r[0]); nx[2] += bf_lo(r[1]); nx[3] += bf_hi(r[1]);
                acc[cs][ps] = nx;
            }
        lds_barrier();
        chain_kloop_ldsw<P, SB, CS, PS>(acc, bread, wl1);
        lds_barrier();
#pragma unroll
        for (int ps = 0; ps < PS; ++ps)
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                const f32x4 v = acc[cs][ps];
                if (valid2 & (1u << ps))
                    *reinterpret_cast<u32x2*>(owrite + ps * 16 * SB + cs * 8) = u32x2{pack2_c(relu_c(v[0]), relu_c(v[1])), pack2_c(relu_c(v[2]), relu_c(v[3]))};
            }
        // the next band's DMAs (issued a whole band ago) and the previous band's stores are the only vector-memory operations in flight: wait for
        // them HERE, in front of this band's stores, so that the wait does not include those stores' round trip
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                                         // the band's rows are in the plane; the next band is in staging
#pragma unroll
        for (int i = 0; i < F::NUO; ++i) {
            const int u = i * 512 + tid, r = u / UR, q = u - r * UR, x = q / UPP, part = q - x * UPP;
            if (u < R * UR)
                *reinterpret_cast<u32x4*>(outb + ((size_t)(y0 + r) * W + x) * a.out_ctot + part * 8) = *reinterpret_cast<const u32x4*>(plane + ((r + 2) * P + x + 1) * SB + part * 16);
        }
        }
        if (band + 1 < NB) {
            lds_barrier();                                     // the band's rows have been read out of the plane
#ifdef GRNET_ABLATION
            if (!(a.flags & 64))
#endif
            deposit(y0 + R);
            lds_barrier();                                     // staging is free, the plane holds the next band
            if (band + 2 < NB) request(y0 + 2 * R);
        }
    }
}

// ---- The whole 32-channel 56x56 chain (HR branch 0 of a module: 4 BasicBlocks = 8 convolutions, hrnet.py:141-187) as ONE launch: a PIPELINE OF ROWS.
// conv_bf16_block_frame above is bound by its per-band skeleton (57 % of a launch is not the k-loops: profiles/r06_block_frame_ablation.txt) and pays it once per
// BasicBlock.  Here a workgroup owns a frame and each of its 8 waves owns ONE convolution of the chain for the whole launch:
//   * the wave's weights (9 taps x 32 x 32 bf16 = 18 KB) live in its REGISTERS (72 per lane: the A fragments of all 18 (tap, channel block) MFMAs), loaded once;
//   * in step t wave s computes output row t - 2 s of its convolution (4 column tiles x 2 channel blocks x 9 taps = 72 MFMAs) from three rows of its input ring in LDS
//     and leaves the row -- ReLU, bf16 -- in the input ring of wave s + 1 (the last wave stores to HBM); conv2 of a block seeds its accumulators with bias + x from
//     the ring its conv1 reads (the block's input row).  A lag of two rows per stage keeps producer and consumers of a ring on different rows within a step, so ONE
//     barrier per step is all the synchronisation there is; no wave ever waits for weights, addresses are one base + immediates;
//   * rings: 6 rows for the chain's input (arrives by LDS-DMA three steps ahead, one wave-instruction per step from each of waves 0 .. 3), 5 for a block's input
//     (its conv1 reads rows r - 1 .. r + 1 while conv2, two steps behind, still takes row r - 2 as the residual), 4 between conv1 and conv2: 37 rows of 4 KB = 148 KB;
//   * a ring row is 64 slots of 64 bytes (pixel x in slot x + 1, slots 0 and 57 .. 63 zero): no padding between pixels, the 16-byte part p of slot q sits at
//     p ^ 2 (q >> 2 & 1), which puts the 16 lanes of every ds_read_b128 group on 16 different bank groups for every tile start and tap.
// 74 steps per frame (56 rows + 14 of pipeline + 4 of input lead); results bit-identical to the launch-per-block kernels (same seeds, same tap order, same rounding
// points).  HBM sees the chain's input once and its output once (the three intermediates between blocks no longer exist).
struct PipeGeom {
    static constexpr int W = 56, C = 32, NS = 8, ROWB = 4096, TILES = 4;
    static constexpr int ROWS = 6 + 4 * 4 + 3 * 5;             // ring 0: 6 rows; rings 1, 3, 5, 7: 4; rings 2, 4, 6: 5
    static constexpr int LDS = ROWS * ROWB + 512;             // + the two slots a tile's right-most taps reach past the last row
    static_assert(ROWS == 37 && LDS <= 160 * 1024, "ring layout");
};

#ifdef GRNET_ABLATION
__device__ unsigned long long g_pipe_phase[8][4];             // diagnostic builds, GRNET_PIPE_PHASES: clock ticks per wave: request / compute + store / DMA wait / barrier
#define PIPE_TICK(k) do { if (a.flags & 128) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc_[k] += t_ - tick_; tick_ = t_; } } while (0)
#else
#define PIPE_TICK(k) do { } while (0)
#endif
__global__ __launch_bounds__(512) void conv_bf16_chain_pipe(const ChainArgs a) {
    typedef PipeGeom G;
    constexpr int W = G::W;
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // waves w and w + 4 share a SIMD: they get conv1 and conv2 of the same block, whose steps are COMPLEMENTARY (below)
    const int s = wv < 4 ? 2 * wv : 2 * (wv - 4) + 1;         // this wave's convolution
    const int n = blockIdx.x;
    if (n >= a.N) return;
    const u16* inb = reinterpret_cast<const u16*>(a.in) + (size_t)n * W * W * a.in_ctot + a.in_coff;
    u16* outb = reinterpret_cast<u16*>(a.out) + (size_t)n * W * W * a.out_ctot + a.out_coff;
    const bool last = s == G::NS - 1, second = (s & 1) != 0;

    // A fragments of this wave's convolution: lane (row i = l15, k group lq) of channel block cs holds W[tap][channel 8 (i >> 2) + 4 cs + (i & 3)][8 lq .. 8 lq + 7]
    // (the row permutation of conv_bf16_block_frame: a lane's two accumulator blocks are the pixel's channels 8 lq .. 8 lq + 7)
    bf16x8 wf[9][2];
    f32x4 bias[2];
    {
        const u16* wsrc = reinterpret_cast<const u16*>(a.w[s]);
        const float* bsrc = a.bias[s];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int cs = 0; cs < 2; ++cs) wf[tap][cs] = *reinterpret_cast<const bf16x8*>(wsrc + ((size_t)tap * 32 + 8 * (l15 >> 2) + 4 * cs + (l15 & 3)) * 32 + 8 * lq);
#pragma unroll
        for (int cs = 0; cs < 2; ++cs) bias[cs] = *reinterpret_cast<const f32x4*>(bsrc + 8 * lq + 4 * cs);
    }
    for (int u = tid; u < G::LDS / 16; u += 512) reinterpret_cast<u32x4*>(lds)[u] = u32x4{0u, 0u, 0u, 0u};
    // ring k = the input of convolution k; this wave reads ring s (+ ring s - 1, the block's input, for the residual of a second convolution), writes ring s + 1
    auto first_of = [](int k) { return k == 0 ? 0 : 6 + ((k - 1) >> 1) * 9 + (((k - 1) & 1) ? 4 : 0); };      // rings 1, 2, 3, ... = 4, 5, 4, 5, ... rows behind ring 0's 6
    auto rows_of = [](int k) { return k == 0 ? 6 : (k & 1) ? 4 : 5; };
    const int f_in = first_of(s), n_in = rows_of(s), f_res = first_of(s > 0 ? s - 1 : 0), n_res = rows_of(s > 0 ? s - 1 : 0), f_out = first_of(s + 1), n_out = rows_of(s + 1);
    // lane offsets inside a ring row for tap column dx: slot 16 j + l15 + dx, part lq swizzled by the slot's bit 2 (16 j leaves that bit alone)
    unsigned off[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) off[dx] = (unsigned)((l15 + dx) * 64 + ((lq ^ ((((l15 + dx) >> 2) & 1) << 1)) * 16));
    // the chain's input, one wave-instruction of a row per step from waves 0 .. 3: unit d = 64 wv + lane of the row = (slot d >> 2, swizzled part d & 3)
    const int dslot = (64 * wv + lane) >> 2, dpart = ((64 * wv + lane) & 3) ^ (((dslot >> 2) & 1) << 1);
    const bool dma_lane = wv < 4 && dslot >= 1 && dslot <= W;
    // The piece is inline asm hipcc does not know of (dma16_masked: uniform row base + 32-bit lane offset, EXEC = the row's real units): with an LDS-DMA it DOES know
    // of in the loop, every wait in front of a fragment's MFMAs is lgkmcnt(0) -- the younger reads included; without, the waits are counted (lgkmcnt(7): one read
    // back of eight).  Its landing is awaited explicitly (vmcnt below, then the step's barrier), as before.
    const unsigned dlane = dma_lane ? (unsigned)(((dslot - 1) * a.in_ctot + dpart * 8) * 2) : 0u;
    const unsigned long long dmask = __ballot(dma_lane);
    const unsigned lds0 = (unsigned)(size_t)lds;
    auto request = [&](int y, int i_row) {                     // row y of the input -> ring 0 (rows outside the image stay zero: the ring starts zeroed, row 56's slot is zeroed below)
        dma16_masked(dlane, reinterpret_cast<const unsigned char*>(inb) + (size_t)y * W * a.in_ctot * 2, lds0 + (unsigned)(i_row * G::ROWB + wv * 1024), dmask);
    };
    f32x4 acc[2][4];
    // row r of this wave's convolution into acc: bias (+ the block's input row r for a second convolution), 9 taps x 4 column tiles x 2 channel blocks
    // ring rows of this step's image rows, kept as counters that wrap (a modulo by a per-wave ring size costs ~40 scalar instructions, six times a step):
    // i_in: row r - 1 in the input ring, i_res: row r in the residual ring, i_out: row r in the output ring, i_dma: row t + 3 in ring 0
    int i_in = 0, i_res = 0, i_out = 0, i_dma = 0;
    auto wrap = [](int i, int nrows) { return i >= nrows ? i - nrows : i; };
    auto compute_row = [&](int r) {
        (void)r;
#ifdef GRNET_ABLATION
        if (a.flags & 64) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[0][j] = bias[0]; acc[1][j] = bias[1]; }
        } else
#endif
        if (second) {
            const unsigned char* xr = lds + (f_res + i_res) * G::ROWB + off[1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 rr = *reinterpret_cast<const u32x4*>(xr + j * 1024);
#pragma unroll
                for (int cs = 0; cs < 2; ++cs) {
                    f32x4 nx = bias[cs];
                    nx[0] += bf_lo(rr[2 * cs]); nx[1] += bf_hi(rr[2 * cs]); nx[2] += bf_lo(rr[2 * cs + 1]); nx[3] += bf_hi(rr[2 * cs + 1]);
                    acc[cs][j] = nx;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[0][j] = bias[0]; acc[1][j] = bias[1]; }
        }
#ifdef GRNET_ABLATION
        if (a.flags & 16) return;
#endif
        const unsigned char* rb[3];
        rb[0] = lds + (f_in + i_in) * G::ROWB;
        rb[1] = lds + (f_in + wrap(i_in + 1, n_in)) * G::ROWB;
        rb[2] = lds + (f_in + wrap(i_in + 2, n_in)) * G::ROWB;
        // pixel fragments as a ring of two sets: fragment j of tap t + 2 is requested right behind the two MFMAs that consumed fragment j of tap t (16 MFMAs = 256
        // cycles ahead of its use), so every wait in front of an MFMA pair is for ONE read with seven younger ones in flight
        bf16x8 px[2][4];
#pragma unroll
        for (int pre = 0; pre < 2; ++pre)
#pragma unroll
            for (int j = 0; j < 4; ++j) px[pre][j] = *reinterpret_cast<const bf16x8*>(rb[0] + off[pre] + j * 1024);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap][0], px[tap & 1][j], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap][1], px[tap & 1][j], acc[1][j], 0, 0, 0);
                if (tap + 2 < 9) px[tap & 1][j] = *reinterpret_cast<const bf16x8*>(rb[(tap + 2) / 3] + off[(tap + 2) % 3] + j * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // acc (row r) -> ReLU, bf16 -> ring s + 1 (the last convolution: HBM); zero: the padding rows -1 and 56 of the next convolution's input
    auto store_row = [&](int r, int i_row, bool zero) {            // i_row: ring row of image row r in the output ring
#ifdef GRNET_ABLATION
        if (a.flags & 32) return;
#endif
        unsigned char* orow = lds + (f_out + i_row) * G::ROWB + off[1];
        if (zero) {
            if (!last) {
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(orow + j * 1024) = u32x4{0u, 0u, 0u, 0u};
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // round to bf16, THEN clamp at zero on the packed halves (v_pk_max_i16: a negative bf16 is a negative 16-bit integer, -0 included) -- the bits of
            // relu-then-round (rounding keeps the sign) in 8 vector instructions per tile instead of 12
            const f32x4 A = acc[0][j], B = acc[1][j];
            u32x4 v = u32x4{relu_pk(pack2_c(A[0], A[1])), relu_pk(pack2_c(A[2], A[3])), relu_pk(pack2_c(B[0], B[1])), relu_pk(pack2_c(B[2], B[3]))};
            const int x = 16 * j + l15;
            if (last) {
                if (j < 3 || x < W) *reinterpret_cast<u32x4*>(outb + ((size_t)r * W + x) * a.out_ctot + lq * 8) = v;
            } else {
                if (j == 3 && x >= W) v = u32x4{0u, 0u, 0u, 0u};     // slots 57 .. 64 (the right halo, the spare slots, the next row's left halo) stay zero
                *reinterpret_cast<u32x4*>(orow + j * 1024) = v;
            }
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // weights and bias are in registers (nothing else of this wave's is in flight from here on but DMAs and stores)
    lds_barrier();                                             // the rings are zero
    // Schedule.  Convolution s works on row t - c(s) in step t, c = 0, 2, 5, 7, 10, 12, 15, 17: a first convolution computes its row and stores it in the same step
    // (matrix pipe, then vector ALU); a second convolution stores the row of the PREVIOUS step first, then seeds and computes the next (vector ALU, then matrix
    // pipe) -- the two waves of a SIMD are in opposite phases.  A second convolution's row therefore appears one step later, and the next block's first
    // convolution follows it by three steps instead of two; within a step no wave reads a row another one writes (ring sizes above), so one barrier per step.
    const int c_s = 5 * (s >> 1) + 2 * (s & 1);
    const int t_end = W + 17 + 2;                              // the last convolution computes row 55 in step 72 and stores it in step 73
    int pending = 0;                                           // second convolutions: 1 = acc holds the previous step's row, 2 = a padding row is due
    {   // the counters at t = -4 (rows below -1 are never touched: only the phase matters; 120 = a multiple of every ring size)
        const int r0 = -4 - c_s;
        i_in = (r0 - 1 + 120) % n_in; i_res = (r0 + 120) % n_res; i_out = (r0 + 120) % n_out; i_dma = (-4 + 3 + 120) % 6;
    }

#ifdef GRNET_ABLATION
    unsigned long long tacc_[4] = {0, 0, 0, 0}, tick_ = __builtin_readcyclecounter();
#endif
#pragma unroll 1
    for (int t = -4; t < t_end; ++t) {
        if (t + 3 == W && wv == 0) {                           // ring 0's row for image row 56 (the bottom padding) held row 50: zero it (slots 0 .. 63)
            unsigned char* z = lds + i_dma * G::ROWB;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(z + (i * 64 + lane) * 16) = u32x4{0u, 0u, 0u, 0u};
        }
        const bool asks = wv < 4 && t + 3 >= 0 && t + 3 < W;
        if (asks) request(t + 3, i_dma);
        const int r = t - c_s;
        PIPE_TICK(0);
        if (second) {
            if (pending) store_row(r - 1, i_out == 0 ? n_out - 1 : i_out - 1, pending == 2);
            pending = 0;
            if (r >= 0 && r < W) { compute_row(r); pending = 1; }
            else if (r == -1 || r == W) pending = 2;
        } else if (r >= -1 && r <= W) {
            const bool pad = r < 0 || r == W;
            if (!pad) compute_row(r);
            store_row(r, i_out, pad);
        }
        i_in = wrap(i_in + 1, n_in); i_res = wrap(i_res + 1, n_res); i_out = wrap(i_out + 1, n_out); i_dma = wrap(i_dma + 1, 6);
        PIPE_TICK(1);
        // the row this wave requested one step ago has landed (this step's may still fly; a step without a request leaves nothing in flight)
        if (asks) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (wv < 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PIPE_TICK(2);
#ifdef GRNET_ABLATION
        if (!(a.flags & 256))
#endif
        lds_barrier();
        PIPE_TICK(3);
    }
#ifdef GRNET_ABLATION
    if ((a.flags & 128) && lane == 0)
        for (int k = 0; k < 4; ++k) atomicAdd(&g_pipe_phase[wv][k], tacc_[k]);
#endif
}

// ---- ONE wide 3x3 stride-1 convolution (upsample heads hrnet.py:440-453, PARE head pare.py:377-400, layer1's 3x3 hrnet.py:80-100) with a band of
// the input resident in LDS.  conv_bf16_nhwc runs these layers at 0.25-0.45 of the matrix peak: 224-pixel x 64-channel tiles, 32 input
// channels per barrier (126 MFMAs per wave between two barriers, each with a vmcnt(0) in front).  Here a workgroup owns R output rows of one
// frame x CT = CP output channels (CP = 128, or 64 for the 64-channel layers): the R + 2 input rows go HBM -> LDS by LDS-DMA straight into the
// padded, flattened plane of the chain kernel (the pad units and every zero -- halo column, rows outside the image -- come from a block of
// zeros: the DMA writes lane-linear, so it cannot skip them), CP input channels per pass; between two barriers a wave issues 9 x CP/32 x 26
// MFMAs (936 for CP = 128).  Layers with more than CP input channels take several passes (480 = 128 + 128 + 128 + 96) into the same
// accumulators; their weights are one contiguous stream ([chunk][tap][CoutPad][32]), so the ring of weight fragments runs across passes.
// The tile leaves through the plane (in place, as in the chain kernel) as whole channel rows.
template <int CP, int CT, int W, int R>
struct WideGeom {
    static constexpr int P = W + 1, SB = 2 * CP + 32, UPS = SB / 16;
    static constexpr int ROWS = R + 2;                      // plane rows: image rows y0 - 1 .. y0 + R
    static constexpr int O0 = P + 1, NOUT = R * P - 1;
    static constexpr int CS = 2, WCB = CT / 32, WPG = 8 / WCB;    // CT output channels per workgroup (CT <= CP: the tile leaves through the first 2 CT bytes of the plane's slots)
    static constexpr int PS = ((NOUT + 15) / 16 + WPG - 1) / WPG, NT = WPG * PS;
    static constexpr int NSLOT = O0 + NT * 16 + P + 2;
    static constexpr int LDS = NSLOT * SB;
    static constexpr int FILL_UNITS = ((ROWS * P + 1) * UPS + 63) / 64 * 64;      // slots 0 .. ROWS * P (the last one: the right halo of the last row), whole wave-instructions
    static constexpr int NFILL = (FILL_UNITS / 64 + 7) / 8;                        // wave-instructions per wave
    static constexpr int UPP = CT / 8, NUO = (R * W * UPP + 511) / 512;
    static constexpr int NB = (W + R - 1) / R;
    static_assert(((SB / 32) % 2) == 1 && LDS <= 160 * 1024 && PS <= 32 && FILL_UNITS * 16 <= LDS && CT <= CP && CT % 32 == 0 && 8 % (CT / 32) == 0, "wide-band geometry");
    // (with CP = 128 the farthest pixel fragment lies 89 KB behind the lane's base: past the 16-bit ds_read immediate, hipcc keeps a second base register)
};

// chain_kloop with a run-time chunk count and weight stride (the output-channel padding of the layer): wc = the pass's first k-step, wtap = elements
// per k-step (CoutPad x 32); the ring's two leading steps of the NEXT pass are simply the next two k-steps of the stream -- except behind the very
// last chunk of the layer (last), where the stream ends: the pass's own first steps are re-requested (nobody waits for them).
template <int P, int SB, int CS, int PS>
__device__ __forceinline__ void wide_kloop(f32x4 (&acc)[CS][PS], bf16x8 (&wr)[3][CS], const unsigned char* bread, const u16* wc, size_t wtap, int nch, bool last, unsigned wlb) {
    bf16x8 bfr[PS];
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) bfr[ps] = *reinterpret_cast<const bf16x8*>(bread + ps * 16 * SB);
#pragma unroll 1
    for (int chunk = 0; chunk < nch; ++chunk) {
        const unsigned char* bch = bread + chunk * 64;
        const u16* wch = wc + (size_t)chunk * 9 * wtap;
        const bool lastc = last && chunk == nch - 1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            {
                const u16* src = wch + (size_t)(tap + 2) * wtap;
                if (tap >= 7) src = lastc ? wc + (size_t)(tap - 7) * wtap : src;
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) wr[(tap + 2) % 3][cs] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(src + cs * 16 * 32) + wlb);
            }
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
}

template <int CP, int CT, int W, int R>
__global__ __launch_bounds__(512) void conv_bf16_wide_band(const ConvArgs a) {
    typedef WideGeom<CP, CT, W, R> G;
    constexpr int P = G::P, SB = G::SB, CS = G::CS, PS = G::PS, UPS = G::UPS, UPP = G::UPP;
    extern __shared__ __align__(16) unsigned char plane[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wcb = wave % G::WCB, pg = wave / G::WCB;
    const int ncb = a.CoutPad / CT;                            // output-channel tiles of the layer
    // workgroups go to the 8 XCDs round-robin by blockIdx: with a.xcd the ids are re-dealt so that CONSECUTIVE tiles -- the ncb output-channel tiles of a band,
    // then the frame's next band (which shares two halo rows) -- run on ONE XCD at about the same time and meet in its L2
    int bid = blockIdx.x;
    if (a.xcd && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int cbo = bid % ncb, nb = bid / ncb, n = nb / G::NB, band = nb - n * G::NB;
    if (n >= a.N) return;
    const int y0 = band * R;
    const u16* inb = reinterpret_cast<const u16*>(a.in) + (size_t)n * W * W * a.in_ctot + a.in_coff;
    const u16* zeros = reinterpret_cast<const u16*>(a.zeros);

    // DMA unit u = (slot, 16-byte part) of the plane in memory order.  The unit -> pixel map does not depend on the pass, and hipcc would hoist it
    // out of the pass loop into 2 x NFILL registers held beside the accumulators (164-212 bytes of scratch): the lane index is laundered through an
    // empty asm per pass, so the map is recomputed (~20 scalar-free instructions per unit, 19 units per pass) instead of kept.
    auto fill = [&](int c0, int cw) {                          // input channels c0 .. c0 + cw - 1 of the band -> plane, by LDS-DMA
        int ln = lane;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int i = 0; i < G::NFILL; ++i) {
            const int ub = (i * 8 + wave) * 64;
            if (ub >= G::FILL_UNITS) break;                    // wave-uniform
            const int u = ub + ln, slot = u / UPS, part = u - slot * UPS, r = slot / P, xx = slot - r * P, y = y0 - 1 + r;
            const bool data = r < G::ROWS && xx != 0 && y >= 0 && y < W && part * 8 < cw;
            dma16_c(data ? inb + (size_t)(y * W + xx - 1) * a.in_ctot + c0 + part * 8 : zeros, plane + ub * 16);
        }
    };

    const int o_first = G::O0 + pg * PS * 16 + l15;
    const unsigned char* bread = plane + (o_first - P - 1) * SB + lq * 16;
    unsigned char* owrite = plane + o_first * SB + (wcb * CS * 16 + lq * 4) * 2;
    unsigned valid = 0;
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
        const int o = o_first + ps * 16, r = o / P;
        if (o - r * P != 0 && r >= 1 && r <= R && y0 + r - 1 < W) valid |= 1u << ps;
    }
    const int co = cbo * CT + wcb * CS * 16;                   // first output channel of this wave
    const unsigned wlb = ((co + l15) * 32 + lq * 8) * 2;
    const size_t wtap = (size_t)a.CoutPad * 32;
    const u16* wg = reinterpret_cast<const u16*>(a.w);
    f32x4 acc[CS][PS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + co + cs * 16 + lq * 4);
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) acc[cs][ps] = bv;
    }
    bf16x8 wr[3][CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        wr[0][cs] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(wg + cs * 16 * 32) + wlb);
        wr[1][cs] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(wg + wtap + cs * 16 * 32) + wlb);
    }
    const int npass = (a.CinPad + CP - 1) / CP;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
        const int c0 = pass * CP, cw = a.CinPad - c0 < CP ? a.CinPad - c0 : CP;
        if (pass) __syncthreads();                             // every wave has finished reading the previous pass's plane
#ifdef GRNET_ABLATION                                                  // timing-only builds (make ABLATION=1; GRNET_WIDE_DBG bit 0: no fill, bit 1: no k-loop)
        if (!(a.dbg & 1) || pass == 0)
#endif
        fill(c0, cw);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#ifdef GRNET_ABLATION
        if (!(a.dbg & 2))
#endif
        wide_kloop<P, SB, CS, PS>(acc, wr, bread, wg + (size_t)(c0 / 32) * 9 * wtap, wtap, cw / 32, pass == npass - 1, wlb);
    }
    // ---- the tile: bias is in the accumulators; ReLU, bf16, in place through the plane, rows y0 .. y0 + R - 1 -> HBM as whole channel rows
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < PS; ++ps)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            f32x4 v = acc[cs][ps];
            if (a.relu) { v[0] = relu_c(v[0]); v[1] = relu_c(v[1]); v[2] = relu_c(v[2]); v[3] = relu_c(v[3]); }
            if (valid & (1u << ps)) *reinterpret_cast<u32x2*>(owrite + ps * 16 * SB + cs * 32) = u32x2{pack2_c(v[0], v[1]), pack2_c(v[2], v[3])};
        }
    __syncthreads();
    u16* outb = reinterpret_cast<u16*>(a.out) + (size_t)n * W * W * a.out_ctot + a.out_coff + cbo * CT;
    const int cstore = a.Cout - cbo * CT;                      // real channels of this tile (CoutPad may exceed Cout)
#pragma unroll
    for (int i = 0; i < G::NUO; ++i) {
        const int u = i * 512 + tid, px = u / UPP, part = u - px * UPP, r = px / W, x = px - r * W;
        if (u < R * W * UPP && y0 + r < W && part * 8 < cstore)
            *reinterpret_cast<u32x4*>(outb + ((size_t)(y0 + r) * W + x) * a.out_ctot + part * 8) = *reinterpret_cast<const u32x4*>(plane + ((r + 1) * P + x + 1) * SB + part * 16);
    }
}

// ---- The wide 3x3 convolution again, with BOTH operands streamed through LDS by DMA under the MFMAs (round 5).  conv_bf16_wide_band holds CP = 128
// channels of the band in one plane and refills it between passes: fill -> vmcnt(0) -> barrier -> 936 MFMAs per wave -> barrier, nothing overlapped (one
// workgroup per CU owns the whole LDS).  Ablated at 256 frames (make ABLATION=1, GRNET_WIDE_DBG): 480 -> 256 @56 takes 1 277 us, 1 054 without the three
// refills, 177 without any k-loop (first fill + store): the k-loop alone runs at 2.0 PFLOP/s -- the MFMA issue rate at the clock the chip holds under this
// load -- and 0.4 of 1.28 ms is exposed fill and store.
// Here a plane holds ONE 32-channel chunk of the band (slot stride 64 + 32 bytes: 32 * odd, conflict-free b128 reads as before) and two planes alternate:
// chunk c is computed from plane c % 2 while the pieces of chunk c + 1 land in the other one.  ONE barrier per chunk, behind tap 7: every wave's reads of
// chunk c are done by then (tap 8's fragments are in registers) and its pieces of chunk c + 1 have landed, so tap 8 reads ahead into chunk c + 1 and the
// first piece of chunk c + 2 goes out; 234 MFMAs per wave between barriers.
// vmcnt retires in order, and hipcc drains it to ZERO in front of every use of a loaded register while an LDS-DMA it knows of is in flight (seen in the ISA of
// a first version with the weights by global_load: one vmcnt(0) per tap).  So NO register-returning vector load is left in the loop: the weights go through
// LDS too -- every wave DMAs the 2 KiB (32 output channels x 32 k) of its own fragments per k-step into a private ring of three slots, two steps ahead, XOR-
// swizzled like the frame kernel's -- every DMA is inline asm the compiler does not count, and the waits are explicit: per tap the wave issues W(t+2) (two
// pieces) and at most one plane piece, and waits in the middle of the tap with vmcnt(2 + I(t-1) + I(t)) -- everything up to W(t+1) has landed, the plane
// pieces of this and the previous tap may still fly -- then reads W(t+1)'s fragments for the next tap.  I(t) = 1 on the taps that carry a plane piece
// (tap 8 and taps 0 .. NFILL-2: every wave issues the same count -- a wave without an own last piece re-requests the plane's last one, and behind the
// last chunk the pieces fetch zeros into the plane nobody reads any more).
template <int CT, int W, int R, bool DIRECT = false>
struct RingGeom {
    static constexpr int P = W + 1, SB = 96, UPS = SB / 16;
    static constexpr int ROWS = R + 2;
    static constexpr int O0 = P + 1, NOUT = R * P - 1;
    static constexpr int CS = 2, WCB = CT / 32, WPG = 8 / WCB;
    static constexpr int PS = ((NOUT + 15) / 16 + WPG - 1) / WPG, NT = WPG * PS;
    static constexpr int NPIECE = ((ROWS * P + 1) * UPS + 63) / 64;                 // one-KiB DMA pieces of a plane: slots 0 .. ROWS * P
    static constexpr int PB = NPIECE * 1024;                                        // plane stride, bytes
    static constexpr int NFILL = (NPIECE + 7) / 8;                                  // pieces per wave and chunk
    static constexpr int WRING = 2 * PB;                                            // the waves' weight rings: 8 x 3 slots x 2 KiB (a plane's dead columns read into them: garbage, never stored)
    static constexpr int OSB = 2 * CT + 32;                                         // slot stride of the output tile (staged over everything)
    static constexpr int LDS = WRING + 8 * 3 * 2048;
    static constexpr int UPP = CT / 8, NUO = (R * W * UPP + 511) / 512;
    static constexpr int NB = (W + R - 1) / R;
    static constexpr bool piece_at(int tap) { return tap == 8 || tap < NFILL - 1; }
    static_assert(LDS <= 160 * 1024 && NFILL >= 2 && NFILL <= 7 && PS <= 32 && (DIRECT || (O0 + NT * 16) * OSB <= LDS) && CT % 32 == 0 && 8 % (CT / 32) == 0, "ring geometry");
    static_assert((PS - 1) * 16 * SB + (2 * P + 2) * SB + 64 < 65536 && (O0 + NT * 16 + P + 2) * SB + 64 <= LDS - PB, "ds_read immediates / the farthest dead read stays inside the allocation");
};

template <int CT, int W, int R, bool DIRECT = false>
__global__ __launch_bounds__(512) void conv_bf16_wide_ring(const ConvArgs a) {
    typedef RingGeom<CT, W, R, DIRECT> G;
    constexpr int P = G::P, SB = G::SB, CS = G::CS, PS = G::PS, UPS = G::UPS, UPP = G::UPP, PB = G::PB, NFILL = G::NFILL, OSB = G::OSB;
    extern __shared__ __align__(16) unsigned char plane[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wcb = wave % G::WCB, pg = wave / G::WCB;
    const int ncb = a.CoutPad / CT;
    int bid = blockIdx.x;                                                           // a.xcd: consecutive tiles on ONE XCD (see conv_bf16_wide_band)
    if (a.xcd && (gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
    const int cbo = bid % ncb, nb = bid / ncb, n = nb / G::NB, band = nb - n * G::NB;
    if (n >= a.N) return;
    const int y0 = band * R;
    const unsigned char* inb = reinterpret_cast<const unsigned char*>(reinterpret_cast<const u16*>(a.in) + (size_t)n * W * W * a.in_ctot + a.in_coff);
    const int nch = a.CinPad / 32, nstep = nch * 9;
    const unsigned lds0 = (unsigned)(size_t)plane;                                  // LDS byte address of the allocation (M0 takes byte addresses)

    // this lane's share of plane piece i: byte offset of its 16 bytes of chunk 0 in the frame (chunk c: + 64 c).  Units that must read zero -- halo column, rows
    // outside the image -- are the same for every chunk: they are zeroed ONCE below and their lanes are switched off in every piece (EXEC), so a piece needs
    // no second source and is `global_load_lds v_off, s[base]`; the pad units (never read) stay on and fetch the frame's first bytes, so that no piece is empty
    // (an instruction without lanes would not count in vmcnt, and the waits below count instructions)
    unsigned poff[NFILL];
    int pdst[NFILL];
    unsigned long long pmask[NFILL];
#pragma unroll
    for (int i = 0; i < NFILL; ++i) {
        int pc = i * 8 + wave;
        pc = pc < G::NPIECE ? pc : G::NPIECE - 1;                                   // no own last piece: the plane's last one again
        const int u = pc * 64 + lane, slot = u / UPS, part = u - slot * UPS, r = slot / P, xx = slot - r * P, y = y0 - 1 + r;
        const bool data = part < 4 && r < G::ROWS && xx != 0 && y >= 0 && y < W;
        poff[i] = data ? (unsigned)(((y * W + xx - 1) * a.in_ctot + part * 8) * 2) : 0u;
        pdst[i] = pc * 1024;
        pmask[i] = __ballot(data || part >= 4);
        if (!data && part < 4) {                                                    // both planes: never written again
            *reinterpret_cast<u32x4*>(plane + pdst[i] + lane * 16) = u32x4{0u, 0u, 0u, 0u};
            *reinterpret_cast<u32x4*>(plane + PB + pdst[i] + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    auto piece = [&](int i, int c, int pl) {                                        // behind the last chunk: chunk 0 again, into the plane nobody reads any more
        int ce = c < nch ? c : 0;
#ifdef GRNET_ABLATION
        if (a.dbg & 8) ce = 0;                                                      // bit 3: every piece fetches chunk 0 (cache hits): issue cost without the memory latency
#endif
        dma16_masked(poff[i], inb + ce * 64, lds0 + pl + pdst[i], pmask[i]);
    };
    // weights of k-step s for this wave: rows co .. co + 31 of [step][CoutPad][32]; piece cs = 16 rows x 64 B, lane (row = l >> 2, unit = l & 3) fetches the unit
    // (l & 3) ^ 2 (row >> 3 & 1) of its row: the fragment read below finds k-group lq of row l15 at unit lq ^ 2 (l15 >> 3) -- conflict-free b128 reads of 64-byte rows
    const int co = cbo * CT + wcb * CS * 16;
    const unsigned wlane = (unsigned)(((co + (lane >> 2)) * 32 + (((lane & 3) ^ (2 * ((lane >> 5) & 1))) * 8)) * 2);
    const size_t wstep = (size_t)a.CoutPad * 64;                                    // bytes per k-step
    const unsigned char* wg = reinterpret_cast<const unsigned char*>(a.w);
    const unsigned wring = lds0 + G::WRING + wave * (3 * 2048);
    auto wdma = [&](int s, int slot) {                                              // behind the layer's last step the stream ends: its first steps again (nobody reads them)
        const unsigned char* base = wg + (size_t)(s < nstep ? s : s - nstep) * wstep;
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) dma16_hidden_s(wlane + cs * 1024, base, wring + slot * 2048 + cs * 1024);
    };
    const unsigned char* aread = plane + G::WRING + wave * (3 * 2048) + l15 * 64 + ((lq ^ (2 * (l15 >> 3))) * 16);

    const int o_first = G::O0 + pg * PS * 16 + l15;
    const unsigned char* bread = plane + (o_first - P - 1) * SB + lq * 16;
    unsigned valid = 0;
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
        const int o = o_first + ps * 16, r = o / P;
        if (o - r * P != 0 && r >= 1 && r <= R && y0 + r - 1 < W) valid |= 1u << ps;
    }
    f32x4 acc[CS][PS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + co + cs * 16 + lq * 4);
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) acc[cs][ps] = bv;
    }
    // ---- prologue: chunk 0 -> plane 0, k-steps 0 and 1, the last piece of chunk 1 (the piece "tap 8 of chunk -1" would have issued)
#pragma unroll
    for (int i = 0; i < NFILL; ++i) piece(i, 0, 0);
    wdma(0, 0);
    wdma(1, 1);
    piece(NFILL - 1, 1, PB);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CS + 1) : "memory");                   // in order: chunk 0 and step 0 have landed; step 1 and the piece of chunk 1 may fly
    __syncthreads();                                                                // (and the zeroed halo units are everybody's)
    bf16x8 bfr[PS], afr[3][CS];                                                     // fragment sets in ring order too: step s0 + tap uses set tap % 3 (9 = 3 x 3)
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) afr[0][cs] = *reinterpret_cast<const bf16x8*>(aread + cs * 1024);
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) bfr[ps] = *reinterpret_cast<const bf16x8*>(bread + ps * 16 * SB);
    int cur = 0, nxt = PB;
#pragma unroll 1
    for (int c = 0; c < nch; ++c) {
        const unsigned char* bc = bread + cur;
        const unsigned char* bn = bread + nxt;
        const int s0 = c * 9;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // k-step s0 + tap + 2 -> ring slot (tap + 2) % 3 (9 = 3 x 3: the slot is static), then this tap's plane piece: chunk c + 1 on taps 0 .. NFILL-2
            // (pieces 0 .. NFILL-2; into the other plane), chunk c + 2 on tap 8 (piece NFILL-1; into THIS plane, which the barrier behind tap 7 has freed)
#ifdef GRNET_ABLATION                                                  // timing-only builds (make ABLATION=1; GRNET_WIDE_DBG bit 0: no plane pieces, bit 1: no weight pieces, bit 2: no waits)
            if (!(a.dbg & 2))
#endif
            wdma(s0 + tap + 2, (tap + 2) % 3);
#ifdef GRNET_ABLATION
            if (!(a.dbg & 1)) {
#endif
            if (tap == 8) piece(NFILL - 1, c + 2, cur);
            else if (tap < NFILL - 1) piece(tap, c + 1, nxt);
#ifdef GRNET_ABLATION
            }
#endif
            const unsigned char* nb_ = tap < 8 ? bc + (((tap + 1) / 3) * P + ((tap + 1) % 3)) * SB : bn;
            constexpr int HALF = PS / 2;
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) {
                if (ps == HALF) {
                    // everything up to k-step s0 + tap + 1 has landed (issued one tap ago); behind it: that tap's plane piece, this tap's two weight pieces and plane piece
#ifdef GRNET_ABLATION
                    if (!(a.dbg & 4))
#endif
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CS + (G::piece_at((tap + 8) % 9) ? 1 : 0) + (G::piece_at(tap) ? 1 : 0)) : "memory");
#pragma unroll
                    for (int cs = 0; cs < CS; ++cs) afr[(tap + 1) % 3][cs] = *reinterpret_cast<const bf16x8*>(aread + ((tap + 1) % 3) * 2048 + cs * 1024);
                }
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) acc[cs][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[tap % 3][cs], bfr[ps], acc[cs][ps], 0, 0, 0);
                bfr[ps] = *reinterpret_cast<const bf16x8*>(nb_ + ps * 16 * SB);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tap == 7) {
                // every read of chunk c has returned (tap 8's fragments included); this wave's pieces of chunk c + 1 landed with the wait in the middle of this tap
                // (the last one went out on tap NFILL-2 <= 5)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        const int t = cur; cur = nxt; nxt = t;
    }
    if constexpr (DIRECT) {
        // 256 output channels per workgroup: the tile (R x W pixels x 512 bytes) does not fit the LDS beside nothing -- it leaves straight from the accumulators, 8 bytes per
        // lane and block (a pixel's 64 bytes of this wave are two stores back to back; the eight channel waves complete its 512-byte row in L2)
        const int cstore = a.Cout - cbo * CT;
        u16* outw = reinterpret_cast<u16*>(a.out) + (size_t)n * W * W * a.out_ctot + a.out_coff + cbo * CT + wcb * CS * 16 + lq * 4;
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const int o = o_first + ps * 16, r = o / P, x = o - r * P - 1;
            u16* op = outw + ((size_t)(y0 + r - 1) * W + x) * a.out_ctot;
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                f32x4 v = acc[cs][ps];
                if (a.relu) { v[0] = relu_c(v[0]); v[1] = relu_c(v[1]); v[2] = relu_c(v[2]); v[3] = relu_c(v[3]); }
                if ((valid & (1u << ps)) && wcb * CS * 16 + cs * 16 + lq * 4 < cstore) *reinterpret_cast<u32x2*>(op + cs * 16) = u32x2{pack2_c(v[0], v[1]), pack2_c(v[2], v[3])};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                           // no piece may land in an LDS that already belongs to somebody else
        return;
    }
    // ---- the tile: ReLU, bf16, staged over planes and rings (every piece has landed, every wave is done reading), rows y0 .. y0 + R - 1 -> HBM as whole channel rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned char* owrite = plane + o_first * OSB + (wcb * CS * 16 + lq * 4) * 2;
#pragma unroll
    for (int ps = 0; ps < PS; ++ps)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            f32x4 v = acc[cs][ps];
            if (a.relu) { v[0] = relu_c(v[0]); v[1] = relu_c(v[1]); v[2] = relu_c(v[2]); v[3] = relu_c(v[3]); }
            if (valid & (1u << ps)) *reinterpret_cast<u32x2*>(owrite + ps * 16 * OSB + cs * 32) = u32x2{pack2_c(v[0], v[1]), pack2_c(v[2], v[3])};
        }
    __syncthreads();
    u16* outb = reinterpret_cast<u16*>(a.out) + (size_t)n * W * W * a.out_ctot + a.out_coff + cbo * CT;
    const int cstore = a.Cout - cbo * CT;
#pragma unroll
    for (int i = 0; i < G::NUO; ++i) {
        const int u = i * 512 + tid, px = u / UPP, part = u - px * UPP, r = px / W, x = px - r * W;
        if (u < R * W * UPP && y0 + r < W && part * 8 < cstore)
            *reinterpret_cast<u32x4*>(outb + ((size_t)(y0 + r) * W + x) * a.out_ctot + part * 8) = *reinterpret_cast<const u32x4*>(plane + ((r + 1) * P + x + 1) * OSB + part * 16);
    }
}

// ---- ONE 3x3 STRIDE-2 convolution (fuse layers' down paths hrnet.py:213-241, transitions hrnet.py:348-387, the stem's second convolution hrnet.py:470-475) with a
// band of the input resident in LDS.  conv_bf16_nhwc runs these layers at 0.05-0.16 of the matrix peak and 2-3 x their HBM time: 112-pixel tiles of ~60
// MFMAs per wave behind a slot table, a DMA wait, an LDS transpose and two barriers, 1 500 ms-scale launches of them per step (1.5 ms of 11).
// Stride 2 breaks the flattened plane of the stride-1 kernels (tap (dy, dx) of output (Y, X) is input (2Y + dy - 1, 2X + dx - 1): not a constant slot offset)
// -- unless the input is DE-INTERLEAVED by row and column parity into four sub-planes sub(py, px)[Y'][X'] = in[2Y' + py][2X' + px]: then tap (dy, dx) reads
// sub(py, px) at (Y + oy, X + ox) with py = (dy != 1), oy = -(dy == 0) and likewise for x, and inside its sub-plane every tap IS a constant offset again.
// The LDS-DMA does the de-interleaving for free: its source address is per lane.  Each sub-plane is flattened with pitch Wo + 1 (column X' = -1 is the shared
// zero column, row Y' = y0 - 1 the zero / halo row), R + 1 rows; output column o = (Y - y0)(Wo + 1) + X.  Everything else -- CP input channels per pass,
// weight ring, MFMA roles, in-place tile through LDS -- is conv_bf16_wide_band's; the epilogue adds the layer's fused addends (nearest-upsampled terms of
// the fuse sum, hrnet.py:258-265) before the ReLU.
template <int CP, int CT, int WO, int R>
struct S2Geom {
    static constexpr int P = WO + 1, SB = 2 * CP + 32, UPS = SB / 16, OSB = 2 * CT + 32;
    static constexpr int NOUT = R * P - 1;
    static constexpr int CS = 2, WCB = CT / 32, WPG = 8 / WCB;
    static constexpr int PS = ((NOUT + 15) / 16 + WPG - 1) / WPG, NT = WPG * PS;
    static constexpr int SUBROWS = R + 1;
    static constexpr int SUB = (SUBROWS * P > NT * 16 + P + 1 ? SUBROWS * P : NT * 16 + P + 1);      // slots per sub-plane: its rows, or what the farthest tap of the last column tile reaches
    static constexpr int FILL_UNITS = (4 * SUB * UPS + 63) / 64 * 64;
    static constexpr int NFILL = (FILL_UNITS / 64 + 7) / 8;
    static constexpr int LDS = (4 * SUB * SB + SB > FILL_UNITS * 16 ? 4 * SUB * SB + SB : FILL_UNITS * 16);
    static constexpr int UPP = CT / 8, NUO = (R * WO * UPP + 511) / 512;
    static constexpr int NB = (WO + R - 1) / R;
    static_assert(((SB / 32) % 2) == 1 && LDS <= 160 * 1024 && PS <= 32 && NT * 16 * OSB <= LDS && CT % 32 == 0 && 8 % (CT / 32) == 0, "stride-2 band geometry");
    // byte offset of tap (dy, dx) from the lane's base (sub-plane (0,0), row 0, column slot 0)
    static constexpr int toff(int tap) {
        const int dy = tap / 3, dx = tap % 3, py = dy != 1, px = dx != 1, oy = dy == 0 ? -1 : 0, ox = dx == 0 ? -1 : 0;
        return ((py * 2 + px) * SUB + (oy + 1) * P + (ox + 1)) * SB;
    }
};

template <typename G, int CS, int PS>
__device__ __forceinline__ void s2_kloop(f32x4 (&acc)[CS][PS], bf16x8 (&wr)[3][CS], const unsigned char* bread, const u16* wc, size_t wtap, int nch, bool last, unsigned wlb) {
    constexpr int SB = G::SB;
    bf16x8 bfr[PS];
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) bfr[ps] = *reinterpret_cast<const bf16x8*>(bread + ps * 16 * SB + G::toff(0));
#pragma unroll 1
    for (int chunk = 0; chunk < nch; ++chunk) {
        const unsigned char* bch = bread + chunk * 64;
        const u16* wch = wc + (size_t)chunk * 9 * wtap;
        const bool lastc = last && chunk == nch - 1;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            {
                const u16* src = wch + (size_t)(tap + 2) * wtap;
                if (tap >= 7) src = lastc ? wc + (size_t)(tap - 7) * wtap : src;
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) wr[(tap + 2) % 3][cs] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(src + cs * 16 * 32) + wlb);
            }
            const int noff = tap < 8 ? G::toff(tap + 1) : G::toff(0) + 64;      // the next chunk's first tap; behind the last chunk it reads ahead into padding / the spare slot
#pragma unroll
            for (int ps = 0; ps < PS; ++ps) {
#pragma unroll
                for (int cs = 0; cs < CS; ++cs) acc[cs][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wr[tap % 3][cs], bfr[ps], acc[cs][ps], 0, 0, 0);
                bfr[ps] = *reinterpret_cast<const bf16x8*>(bch + ps * 16 * SB + noff);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <int CP, int CT, int WO, int R>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(S2Geom<CP, CT, WO, R>::LDS <= 80 * 1024 ? 4 : 2))) void conv_bf16_s2_band(const ConvArgs a) {
    typedef S2Geom<CP, CT, WO, R> G;
    constexpr int P = G::P, SB = G::SB, CS = G::CS, PS = G::PS, UPS = G::UPS, UPP = G::UPP, OSB = G::OSB, WI = 2 * WO;
    extern __shared__ __align__(16) unsigned char plane[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wcb = wave % G::WCB, pg = wave / G::WCB;
    const int ncb = a.CoutPad / CT;
    const int cbo = blockIdx.x % ncb, nb = blockIdx.x / ncb, n = nb / G::NB, band = nb - n * G::NB;
    if (n >= a.N) return;
    const int y0 = band * R;
    const u16* inb = reinterpret_cast<const u16*>(a.in) + (size_t)n * WI * WI * a.in_ctot + a.in_coff;
    const u16* zeros = reinterpret_cast<const u16*>(a.zeros);

    auto fill = [&](int c0, int cw) {                          // input channels c0 .. c0 + cw - 1 of the band, de-interleaved -> the four sub-planes, by LDS-DMA
        int ln = lane;
        asm volatile("" : "+v"(ln));                           // (keeps hipcc from hoisting the unit -> pixel map out of the pass loop into registers, as in the wide kernel)
#pragma unroll
        for (int i = 0; i < G::NFILL; ++i) {
            const int ub = (i * 8 + wave) * 64;
            if (ub >= G::FILL_UNITS) break;                    // wave-uniform
            const int u = ub + ln, slot = u / UPS, part = u - slot * UPS, sub = slot / G::SUB, ss = slot - sub * G::SUB, rr = ss / P, xx = ss - rr * P;
            const int py = sub >> 1, px = sub & 1, row = 2 * (y0 - 1 + rr) + py, col = 2 * (xx - 1) + px;
            const bool data = sub < 4 && rr < G::SUBROWS && !(py == 0 && rr == 0) && row >= 0 && row < WI && col >= 0 && col < WI && part * 8 < cw;
            dma16_c(data ? inb + (size_t)(row * WI + col) * a.in_ctot + c0 + part * 8 : zeros, plane + ub * 16);
        }
    };

    const int o_first = pg * PS * 16 + l15;
    const unsigned char* bread = plane + o_first * SB + lq * 16;
    const int co = cbo * CT + wcb * CS * 16;
    const unsigned wlb = ((co + l15) * 32 + lq * 8) * 2;
    const size_t wtap = (size_t)a.CoutPad * 32;
    const u16* wg = reinterpret_cast<const u16*>(a.w);
    f32x4 acc[CS][PS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + co + cs * 16 + lq * 4);
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) acc[cs][ps] = bv;
    }
    bf16x8 wr[3][CS];
#pragma unroll
    for (int cs = 0; cs < CS; ++cs) {
        wr[0][cs] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(wg + cs * 16 * 32) + wlb);
        wr[1][cs] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned char*>(wg + wtap + cs * 16 * 32) + wlb);
    }
    const int npass = (a.CinPad + CP - 1) / CP;
#pragma unroll 1
    for (int pass = 0; pass < npass; ++pass) {
        const int c0 = pass * CP, cw = a.CinPad - c0 < CP ? a.CinPad - c0 : CP;
        if (pass) __syncthreads();
        fill(c0, cw);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        s2_kloop<G, CS, PS>(acc, wr, bread, wg + (size_t)(c0 / 32) * 9 * wtap, wtap, cw / 32, pass == npass - 1, wlb);
    }
    // ---- epilogue: + fused addends (nearest-upsampled by 2^shift), ReLU, bf16; through LDS as whole channel rows
    const int Ho = WO;
    unsigned valid = 0;
    int pix[PS];                                               // (Y << 8) | X of this lane's column of tile ps
#pragma unroll
    for (int ps = 0; ps < PS; ++ps) {
        const int o = o_first + ps * 16, yy = o / P, X = o - yy * P;
        pix[ps] = ((y0 + yy) << 8) | X;
        if (X < WO && yy < R && y0 + yy < Ho) valid |= 1u << ps;
    }
#pragma unroll
    for (int k = 0; k < kMaxAdd; ++k) {
        if (k >= a.n_add) break;
        const int sh = a.add_shift[k], hs = Ho >> sh, ws = WO >> sh;
        const u16* ab = reinterpret_cast<const u16*>(a.add[k]) + (size_t)n * hs * ws * a.add_ctot[k] + a.add_coff[k] + co + lq * 4;
        u32x2 r[PS][CS];
#pragma unroll
        for (int ps = 0; ps < PS; ++ps) {
            const bool ok = (valid >> ps) & 1u;
            const int Y = ok ? pix[ps] >> 8 : 0, X = ok ? pix[ps] & 255 : 0;
            const u16* ap = ab + ((size_t)(Y >> sh) * ws + (X >> sh)) * a.add_ctot[k];
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) r[ps][cs] = ok ? *reinterpret_cast<const u32x2*>(ap + cs * 16) : u32x2{0u, 0u};
        }
#pragma unroll
        for (int ps = 0; ps < PS; ++ps)
#pragma unroll
            for (int cs = 0; cs < CS; ++cs) {
                acc[cs][ps][0] += bf_lo(r[ps][cs][0]); acc[cs][ps][1] += bf_hi(r[ps][cs][0]);
                acc[cs][ps][2] += bf_lo(r[ps][cs][1]); acc[cs][ps][3] += bf_hi(r[ps][cs][1]);
            }
    }
    __syncthreads();                                           // every wave has finished reading the sub-planes
    unsigned char* owrite = plane + o_first * OSB + (wcb * CS * 16 + lq * 4) * 2;
#pragma unroll
    for (int ps = 0; ps < PS; ++ps)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
            f32x4 v = acc[cs][ps];
            if (a.relu) { v[0] = relu_c(v[0]); v[1] = relu_c(v[1]); v[2] = relu_c(v[2]); v[3] = relu_c(v[3]); }
            if (valid & (1u << ps)) *reinterpret_cast<u32x2*>(owrite + ps * 16 * OSB + cs * 32) = u32x2{pack2_c(v[0], v[1]), pack2_c(v[2], v[3])};
        }
    __syncthreads();
    u16* outb = reinterpret_cast<u16*>(a.out) + (size_t)n * Ho * WO * a.out_ctot + a.out_coff + cbo * CT;
    const int cstore = a.Cout - cbo * CT;
#pragma unroll
    for (int i = 0; i < G::NUO; ++i) {
        const int u = i * 512 + tid, px = u / UPP, part = u - px * UPP, yy = px / WO, X = px - yy * WO;
        if (u < R * WO * UPP && y0 + yy < Ho && part * 8 < cstore)
            *reinterpret_cast<u32x4*>(outb + ((size_t)(y0 + yy) * WO + X) * a.out_ctot + part * 8) = *reinterpret_cast<const u32x4*>(plane + (yy * P + X) * OSB + part * 16);
    }
}

// ---- The narrow 3x3 stride-2 layers of the fuse down paths (32 / 64 input channels) as a WALK OVER OUTPUT ROWS.  These launches are neither compute- nor
// byte-bound in the band kernel above or in conv_bf16_nhwc (0.05-0.15 of the matrix peak at 1.9 TB/s: fill -> barrier -> k-loop -> epilogue -> store, one after
// another per workgroup).  Here a workgroup owns a segment of a frame's output rows; per output row Y it needs input rows 2Y - 1 .. 2Y + 1, of which two are new:
// they arrive by LDS-DMA one step ahead, DE-INTERLEAVED by column parity (odd columns with a leading zero slot, then even columns: tap dx reads odd index x, even
// index x, odd index x + 1 -- 16 consecutive slots for 16 output pixels), 16-byte parts XOR-swizzled by slot bits so that every fragment read is conflict-free.  A
// wave owns one (16-pixel tile, 16-channel block) of the row with that block's weights in its registers for the whole launch (9 x Cin/32 A fragments), so a step is
// 9 or 18 MFMAs per wave, an epilogue straight from the accumulators (bias, fused addends, ReLU, 8-byte stores) and ONE barrier; the ring is 6 input rows (25 KB), so
// four workgroups share a CU and cover each other's memory round trips.  Results equal the band kernel's bit for bit (same k order, same epilogue order).
template <int CIN, int COUT, int WO>
struct S2RowsGeom {
    static constexpr int WI = 2 * WO, SB = 2 * CIN, UPS = SB / 16, KS = CIN / 32;
    static constexpr int MT = (WO + 15) / 16, NBK = COUT / 16, ITEMS = MT * NBK;
    static constexpr int PO = 16 * MT + 1, PE = 16 * MT, SLOTS = PO + PE;      // odd-column plane (index 0 = column -1), even-column plane
    static constexpr int ROWB = SLOTS * SB, RING = 6;
    static constexpr int UNITS = SLOTS * UPS, NDMA = (UNITS + 63) / 64;      // 16-byte units of a row; wave-instructions per row
    static constexpr int LDS = RING * ROWB + 64 * 16;                         // + what the last instruction of the last row writes past it (masked lanes write nothing)
    static_assert(ITEMS <= 8 && 2 * NDMA <= 16 && (CIN == 32 || CIN == 64) && COUT % 16 == 0 && LDS <= 64 * 1024, "stride-2 row geometry");
    static __device__ __forceinline__ int swz(int slot) { return CIN == 32 ? (((slot >> 2) & 1) << 1) : (((slot >> 1) & 3) << 1); }
};

template <int CIN, int COUT, int WO>
__global__ __launch_bounds__(512) void conv_bf16_s2_rows(const ConvArgs a, int segs) {
    typedef S2RowsGeom<CIN, COUT, WO> G;
    constexpr int WI = G::WI, SB = G::SB, KS = G::KS;
    extern __shared__ __align__(16) unsigned char ring[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x / segs, seg = blockIdx.x - n * segs;
    if (n >= a.N) return;
    const int rows_per = (WO + segs - 1) / segs, y_lo = seg * rows_per, y_hi = min(WO, y_lo + rows_per);
    const u16* inb = reinterpret_cast<const u16*>(a.in) + (size_t)n * WI * WI * a.in_ctot + a.in_coff;
    u16* outb = reinterpret_cast<u16*>(a.out) + (size_t)n * WO * WO * a.out_ctot + a.out_coff;
    const bool works = wave < G::ITEMS;
    const int nb = wave % G::NBK, mt = wave / G::NBK, co = nb * 16;

    bf16x8 wf[KS][9];
    f32x4 bias = f32x4{0.f, 0.f, 0.f, 0.f};
    if (works) {
        const u16* wg = reinterpret_cast<const u16*>(a.w);
#pragma unroll
        for (int kc = 0; kc < KS; ++kc)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wf[kc][tap] = *reinterpret_cast<const bf16x8*>(wg + (((size_t)kc * 9 + tap) * a.CoutPad + co + l15) * 32 + 8 * lq);
        bias = *reinterpret_cast<const f32x4*>(a.bias + co + 4 * lq);
    }
    for (int u = tid; u < G::LDS / 16; u += 512) reinterpret_cast<u32x4*>(ring)[u] = u32x4{0u, 0u, 0u, 0u};
    // this wave's share of a step's two new rows: wave-instructions q = wave, wave + 8 of the 2 x NDMA (row q / NDMA, instruction q % NDMA); per lane the unit's
    // (validity, source offset inside an input row): unit d = (slot, stored part) -> column 2 i - 1 (odd plane, i = slot) or 2 j (even plane), part = stored ^ swz(slot)
    int dq_row[2], dq_k[2], dq_off[2];
    bool dq_on[2], dq_lane[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = wave + 8 * i;
        dq_on[i] = q < 2 * G::NDMA;
        dq_row[i] = q / G::NDMA; dq_k[i] = q - dq_row[i] * G::NDMA;
        const int d = dq_k[i] * 64 + lane, slot = d / G::UPS, sp = d - slot * G::UPS;
        const int col = slot < G::PO ? 2 * slot - 1 : 2 * (slot - G::PO);
        dq_lane[i] = dq_on[i] && d < G::UNITS && col >= 0 && col < WI && (slot < G::PO ? slot <= WO : slot - G::PO < WO);
        dq_off[i] = col * a.in_ctot + (sp ^ G::swz(slot)) * 8;
    }
    // fragment offsets of this wave's tile inside a ring row: tap dx reads odd index x (dx = 0), even index x (1), odd index x + 1 (2), x = 16 mt + l15
    unsigned foff[3][KS];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int x = 16 * mt + l15, slot = dx == 1 ? G::PO + x : x + (dx >> 1);
#pragma unroll
        for (int kc = 0; kc < KS; ++kc) foff[dx][kc] = (unsigned)(slot * SB + (((4 * kc + lq) ^ G::swz(slot)) * 16));
    }
    auto request = [&](int y, int rr0) {                       // input rows 2 y, 2 y + 1 -> ring rows rr0, rr0 + 1 (wrapped by the caller)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (dq_on[i]) {
                const int row = 2 * y + dq_row[i];
                int rr = rr0 + dq_row[i];
                rr = rr >= G::RING ? rr - G::RING : rr;
                if (dq_lane[i] && row < WI) dma16_c(inb + (size_t)row * WI * a.in_ctot + dq_off[i], ring + rr * G::ROWB + dq_k[i] * 1024);
            }
    };
    lds_barrier();                                             // the ring is zero (the halo slots and the row above the image stay zero: no DMA ever writes them ... see below)
    // ring row of input row r: (r + 2) mod 6 by a wrapping counter; row 2 y_lo - 1 (or the zero row above the image) sits at `cur`
    int cur = 0;
    if (y_lo > 0 && wave == 0) {                               // the segment's first output row needs input row 2 y_lo - 1: one extra row, by wave 0
#pragma unroll
        for (int k = 0; k < G::NDMA; ++k) {
            const int d = k * 64 + lane, slot = d / G::UPS, sp = d - slot * G::UPS;
            const int col = slot < G::PO ? 2 * slot - 1 : 2 * (slot - G::PO);
            const bool ok = d < G::UNITS && col >= 0 && col < WI && (slot < G::PO ? slot <= WO : slot - G::PO < WO);
            if (ok) dma16_c(inb + ((size_t)(2 * y_lo - 1) * WI + col) * a.in_ctot + (sp ^ G::swz(slot)) * 8, ring + cur * G::ROWB + k * 1024);
        }
    }
    request(y_lo, cur + 1);
#pragma unroll 1
    for (int y = y_lo; y < y_hi; ++y) {
        // fused addends of this row (nearest-upsampled by 2^shift), requested before the wait below so that it covers them
        u32x2 av[kMaxAdd];
        const int x = 16 * mt + l15;
        const bool px_ok = works && x < WO;
#pragma unroll
        for (int k = 0; k < kMaxAdd; ++k) {
            av[k] = u32x2{0u, 0u};
            if (k < a.n_add && px_ok) {
                const int sh = a.add_shift[k], ws = WO >> sh;
                av[k] = *reinterpret_cast<const u32x2*>(reinterpret_cast<const u16*>(a.add[k]) + ((size_t)n * (WO >> sh) * ws + (size_t)(y >> sh) * ws + (x >> sh)) * a.add_ctot[k] + a.add_coff[k] + co + 4 * lq);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of rows 2 y, 2 y + 1 (requested a step ago) have landed
        lds_barrier();                                         // ... and everybody's; every wave is done with row 2 y - 2 (the next request overwrites it)
        f32x4 af[kMaxAdd];                                     // the addends as floats HERE: hipcc's own wait for their loads must not land behind the next request
#pragma unroll
        for (int k = 0; k < kMaxAdd; ++k) af[k] = f32x4{bf_lo(av[k][0]), bf_hi(av[k][0]), bf_lo(av[k][1]), bf_hi(av[k][1])};
        __builtin_amdgcn_sched_barrier(0);
        int nxt = cur + 3;
        nxt = nxt >= G::RING ? nxt - G::RING : nxt;
        if (y + 1 < y_hi) request(y + 1, nxt);
        if (works) {
            f32x4 acc = bias;
            const unsigned char* r0 = ring + cur * G::ROWB;
            int c1 = cur + 1, c2 = cur + 2;
            c1 = c1 >= G::RING ? c1 - G::RING : c1; c2 = c2 >= G::RING ? c2 - G::RING : c2;
            const unsigned char* r1 = ring + c1 * G::ROWB;
            const unsigned char* r2 = ring + c2 * G::ROWB;
#pragma unroll
            for (int kc = 0; kc < KS; ++kc)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const unsigned char* rb = tap < 3 ? r0 : tap < 6 ? r1 : r2;
                    const bf16x8 px = *reinterpret_cast<const bf16x8*>(rb + foff[tap % 3][kc]);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kc][tap], px, acc, 0, 0, 0);
                }
#pragma unroll
            for (int k = 0; k < kMaxAdd; ++k)
                if (k < a.n_add) { acc[0] += af[k][0]; acc[1] += af[k][1]; acc[2] += af[k][2]; acc[3] += af[k][3]; }
            if (a.relu) { acc[0] = relu_c(acc[0]); acc[1] = relu_c(acc[1]); acc[2] = relu_c(acc[2]); acc[3] = relu_c(acc[3]); }
            if (px_ok) *reinterpret_cast<u32x2*>(outb + ((size_t)y * WO + x) * a.out_ctot + co + 4 * lq) = u32x2{pack2_c(acc[0], acc[1]), pack2_c(acc[2], acc[3])};
        }
        cur += 2;
        cur = cur >= G::RING ? cur - G::RING : cur;
    }
}
// (the three shapes the band kernel lost on; on its own shapes -- 64 -> 128, 32 -> 128, 32 -> 32 @28->14 -- the walk ties with it: 15.1 against 14.5 us, not instantiated)
#define GRK_S2R_SHAPES(X) X(32, 64, 28) X(32, 32, 28) X(64, 64, 14)
template <int CIN, int COUT, int WO>
hipError_t launch_s2_rows(const ConvArgs& a, hipStream_t s) {
    typedef S2RowsGeom<CIN, COUT, WO> G;
    int cus = 0;
    GRK_TRY(device_cu_count(&cus));
    // segments of output rows per frame: several workgroups per CU where the frame count allows it, but never fewer than 7 rows per workgroup (each one loads
    // its channel blocks' weights: 18-74 KB from L2)
    int segs = 1;
    if (a.N < 4 * cus) segs = 2;
    if (WO == 28 && a.N * 2 < 4 * cus) segs = 4;
    return launch_k(conv_bf16_s2_rows<CIN, COUT, WO>, dim3(a.N * segs), dim3(512), G::LDS, s, a, segs);
}

template <int CP, int CT, int WO, int R>
hipError_t set_s2_lds() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_s2_band<CP, CT, WO, R>), hipFuncAttributeMaxDynamicSharedMemorySize, S2Geom<CP, CT, WO, R>::LDS);
}
template <int CP, int CT, int WO, int R>
hipError_t launch_s2(const ConvArgs& a, hipStream_t s) {
    typedef S2Geom<CP, CT, WO, R> G;
    return launch_k(conv_bf16_s2_band<CP, CT, WO, R>, dim3(a.N * G::NB * (a.CoutPad / CT)), dim3(512), G::LDS, s, a);
}
// (CP, CT, Wo, R) per layer shape: the band is as tall as the four sub-planes' LDS allows
// Measured at 256 frames against conv_bf16_nhwc (profiles/r05_s2_band_vs_generic.txt): the band kernel wins where a workgroup's MFMA share is large enough to
// carry its fill -> barrier -> store round trip -- 64 -> 128 @28->14 (20.6 / 23.1 us against 30.2 / 35.1), 32 -> 128 @28->14 (10.5 / 17.3), 32 -> 32 @28->14
// (7.8 / 9.8), the stem's 64 -> 64 @112->56 (216 / 237) -- ties on the 14->7 layers and loses on 256 -> 64 (four passes, each with an exposed fill:
// 174 / 144), 32 -> 32 and 32 -> 64 @56->28 (7-row bands 26.7 / 40.7 against 27.6 / 34.4; 3-row bands with two workgroups per CU 35.0 / 47.5),
// 64 -> 64 @28->14 (17.7 / 12.9), 128 -> 256 @14->7 (30.1 / 24.0): those stay on the generic kernel and are not instantiated.
#define GRK_S2_SHAPES(X) X(64, 64, 56, 3) X(64, 128, 14, 14) X(32, 128, 14, 14) X(32, 32, 14, 14)

template <int C, int W>
hipError_t set_chain_lds() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_chain<C, W>), hipFuncAttributeMaxDynamicSharedMemorySize, ChainGeom<C, W>::LDS);
}

}  // namespace

hipError_t conv_bf16_chain_init() {
    GRK_TRY((set_chain_lds<64, 28>()));
    GRK_TRY((set_chain_lds<128, 14>()));
    GRK_TRY((set_chain_lds<256, 7>()));
    GRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_chain<256, 7, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, ChainGeom<256, 7, 2>::LDS));
    GRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_block_band<32, 56, 19, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, BandGeom<32, 56, 19>::LDS));
#define GRK_S2_SET(cp, ct, wo, r) GRK_TRY((set_s2_lds<cp, ct, wo, r>()));
    GRK_S2_SHAPES(GRK_S2_SET)
#undef GRK_S2_SET
    GRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_wide_band<128, 128, 56, 7>), hipFuncAttributeMaxDynamicSharedMemorySize, WideGeom<128, 128, 56, 7>::LDS));
    GRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_wide_band<128, 128, 28, 14>), hipFuncAttributeMaxDynamicSharedMemorySize, WideGeom<128, 128, 28, 14>::LDS));
    GRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_wide_band<64, 64, 56, 14>), hipFuncAttributeMaxDynamicSharedMemorySize, WideGeom<64, 64, 56, 14>::LDS));
    GRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_block_frame<56, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, FrameGeom<56, 8>::LDS));
    GRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_chain_pipe), hipFuncAttributeMaxDynamicSharedMemorySize, PipeGeom::LDS));
    GRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_wide_ring<128, 56, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, RingGeom<128, 56, 8>::LDS));
    GRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_wide_ring<256, 56, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, RingGeom<256, 56, 4, true>::LDS));
    GRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_wide_ring<32, 56, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, RingGeom<32, 56, 8>::LDS));
    GRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_wide_ring<128, 28, 14>), hipFuncAttributeMaxDynamicSharedMemorySize, RingGeom<128, 28, 14>::LDS));
    GRK_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16_block_band<32, 56, 8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, BandGeom<32, 56, 8>::LDS));
    return hipSuccess;
}

bool conv_bf16_chain_eligible(int c, int w) { return (c == 64 && w == 28) || (c == 128 && w == 14) || (c == 256 && w == 7) || (c == 32 && w == 56); }
// the 56x56 branch: ONE launch for a whole module's chain (conv_bf16_chain_pipe, 8 convolutions), otherwise one band-resident launch per BasicBlock
int conv_bf16_chain_launches(int c, int w, int nconv) { return c == 32 && w == 56 && !(nconv == 8 && GRNET_AB(BF16_PIPE, 1)) ? nconv / 2 : 1; }

// a.in / a.out: NHWC bf16 views of (N, W, W, C) tensors (channel strides in_ctot / out_ctot, first channels in_coff / out_coff, multiples of 8);
// a.w[i]: [C/32][9][C][32] bf16 (pack_conv's bf16 layout with CoutPad = C), a.bias[i]: fp32 [C]; nconv even, <= kMaxChain:
// convolutions 2k, 2k+1 are conv1 / conv2 of BasicBlock k.
hipError_t launch_conv_bf16_chain(const ChainArgs& a, int c, int w, hipStream_t s) {
    if (!conv_bf16_chain_eligible(c, w) || a.nconv < 2 || a.nconv > kMaxChain || (a.nconv & 1) || a.N < 1) return hipErrorInvalidValue;
    if (a.in_ctot % 8 != 0 || a.in_coff % 8 != 0 || a.out_ctot % 8 != 0 || a.out_coff % 8 != 0) return hipErrorInvalidValue;
    if (c == 32 && a.nconv == 8 && GRNET_AB(BF16_PIPE, 1)) {   // the whole chain as one pipeline of rows, a wave per convolution (a.mid is not used)
        ChainArgs b = a;
        b.flags = (GRNET_AB(BF16_FRAME_DBG, 0) << 4) | (GRNET_AB(BF16_PIPE_VSEED, 0) ? 2 : 0);              // (diagnostic builds: timing-only ablation bits 16 / 32 / 64 = no k-loop / no epilogue / no seeds)
#ifdef GRNET_ABLATION
        if (GRNET_AB_SET(PIPE_PHASES)) {
            b.flags |= 128;
            unsigned long long z[32] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_phase), z, sizeof(z));
            GRK_TRY(launch_k(conv_bf16_chain_pipe, dim3(a.N), dim3(512), PipeGeom::LDS, s, b));
            (void)hipStreamSynchronize(s);
            unsigned long long h[32] = {};
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pipe_phase), sizeof(h));
            const double d = (double)a.N * 79.0;                 // per workgroup and step
            for (int w = 0; w < 8; ++w)
                fprintf(stderr, "[pipe phases] wave %d (conv %d): ticks per step  request %.0f  compute+store %.0f  dma wait %.0f  barrier %.0f\n", w, w < 4 ? 2 * w : 2 * (w - 4) + 1,
                        h[w * 4] / d, h[w * 4 + 1] / d, h[w * 4 + 2] / d, h[w * 4 + 3] / d);
            return hipSuccess;
        }
#endif
        GRK_TRY(launch_k(conv_bf16_chain_pipe, dim3(a.N), dim3(512), PipeGeom::LDS, s, b));
        return hipSuccess;
    }
    if (c == 32) {                                           // one launch per BasicBlock; block k > 0 reads what block k - 1 wrote: a.mid holds the intermediates
        ChainArgs b = a;
        for (int k = 0; k < a.nconv / 2; ++k) {
            b.nconv = 2;
            b.w[0] = a.w[2 * k]; b.w[1] = a.w[2 * k + 1]; b.bias[0] = a.bias[2 * k]; b.bias[1] = a.bias[2 * k + 1];
            if (k > 0) { b.in = a.mid[k - 1]; b.in_ctot = a.mid_ctot[k - 1]; b.in_coff = a.mid_coff[k - 1]; }
            if (k < a.nconv / 2 - 1) {
                if (!a.mid[k] || a.mid_ctot[k] % 8 != 0 || a.mid_coff[k] % 8 != 0) return hipErrorInvalidValue;
                b.out = a.mid[k]; b.out_ctot = a.mid_ctot[k]; b.out_coff = a.mid_coff[k];
            } else { b.out = a.out; b.out_ctot = a.out_ctot; b.out_coff = a.out_coff; }
            // 0 (default): workgroup = frame, seven bands of 8 rows, the next band by LDS-DMA under the current band's MFMAs; 19: workgroup = band, one per CU; 8: two per CU
            static const int frame_direct = GRNET_AB(BF16_FRAME_DIRECT, 1);
            b.flags = (frame_direct ? 0 : 1) | (GRNET_AB(BF16_FRAME_DBG, 0) << 4);      // (diagnostic builds: timing-only ablation bits 16 / 32 / 64 = no conv1 k-loop / no conv2 k-loop / no deposit)
            static const int band_rows = GRNET_AB(BF16_BAND, 0);
            if (band_rows == 0) GRK_TRY(launch_k(conv_bf16_block_frame<56, 8>, dim3(a.N), dim3(512), FrameGeom<56, 8>::LDS, s, b));
            else if (band_rows == 19) GRK_TRY(launch_k(conv_bf16_block_band<32, 56, 19, 2>, dim3(a.N * BandGeom<32, 56, 19>::NB), dim3(512), BandGeom<32, 56, 19>::LDS, s, b));
            else GRK_TRY(launch_k(conv_bf16_block_band<32, 56, 8, 4>, dim3(a.N * BandGeom<32, 56, 8>::NB), dim3(512), BandGeom<32, 56, 8>::LDS, s, b));
        }
        return hipSuccess;
    }
    if (c == 64) return launch_k(conv_bf16_chain<64, 28>, dim3(a.N), dim3(512), ChainGeom<64, 28>::LDS, s, a);
    if (c == 128) return launch_k(conv_bf16_chain<128, 14>, dim3(a.N), dim3(512), ChainGeom<128, 14>::LDS, s, a);
    if (GRNET_AB(BF16_CHAIN7_PAIR, 1) && a.N >= 2) return launch_k(conv_bf16_chain<256, 7, 2>, dim3((a.N + 1) / 2), dim3(512), ChainGeom<256, 7, 2>::LDS, s, a);
    return launch_k(conv_bf16_chain<256, 7>, dim3(a.N), dim3(512), ChainGeom<256, 7>::LDS, s, a);
}

// Wide-band kernel: 3x3, stride 1, no fused addend, 56x56 or 28x28 maps, CinPad a multiple of 32; output channels in tiles of 128 (Cin >= 128) or
// 64 (Cin = 64 -> 64); every 16-byte group of the output view must lie inside the buffer.  (A 32-channel tile for transition1's 256 -> 32 -- 6-row bands,
// 3 column tiles per wave -- measured 245 us against the generic kernel's 228 at 256 frames: every pixel fragment feeds two MFMAs only.  Not instantiated.)
bool conv_bf16_wide_eligible(const ConvArgs& a) {
    if (a.ks != 3 || a.stride != 1 || a.n_add != 0 || a.H != a.W || a.Ho != a.H || a.Wo != a.W || a.CinPad % 32 != 0) return false;
    if (a.in_ctot % 8 != 0 || a.in_coff % 8 != 0 || a.out_ctot % 8 != 0 || a.out_coff % 8 != 0 || a.Cout % 8 != 0) return false;
    if (a.W == 56 && a.CinPad == 64 && a.CoutPad == 64) return true;
    // transition1's 256 -> 32 (hrnet.py:348-387): the ring kernel with ONE 32-channel block, every wave a column group (conv_bf16_nhwc ran it at 0.18 of the peak, 2.0 TB/s)
    static const int ct32_env = GRNET_AB(BF16_WIDE_CT32, 1);
    if (ct32_env && a.W == 56 && a.CinPad >= 128 && a.CoutPad == 32) return true;
    return (a.W == 56 || a.W == 28) && a.CinPad >= 128 && a.CoutPad % 128 == 0;
}
hipError_t launch_conv_bf16_wide(const ConvArgs& a0, hipStream_t s) {
    if (!conv_bf16_wide_eligible(a0) || a0.N < 1) return hipErrorInvalidValue;
    static const int xcd_env = GRNET_AB(BF16_XCD, 1);
    ConvArgs a = a0;
    a.xcd = xcd_env;
#ifdef GRNET_ABLATION
    a.dbg = GRNET_AB(WIDE_DBG, 0);
#endif
    // (64 -> 64 on the ring kernel -- 7-row bands, two chunks: the second streams under the first -- measured 93 us against the band kernel's 79: stays here)
    if (a.CinPad == 64) return launch_k(conv_bf16_wide_band<64, 64, 56, 14>, dim3(a.N * WideGeom<64, 64, 56, 14>::NB), dim3(512), WideGeom<64, 64, 56, 14>::LDS, s, a);
    if (a.CoutPad == 32) return launch_k(conv_bf16_wide_ring<32, 56, 8>, dim3(a.N * RingGeom<32, 56, 8>::NB), dim3(512), RingGeom<32, 56, 8>::LDS, s, a);
    const int ncb = a.CoutPad / 128;
    // 1 (default): the ring of one-chunk planes; 0: the 128-channel plane refilled between passes (A/B)
    static const int ring_env = GRNET_AB(BF16_WIDE_RING, 1);
    if (ring_env && a.CinPad >= 64) {
        // 256 output channels at 56x56: ONE workgroup for all channels of a 4-row band (eight channel waves, 15 column tiles each) -- the band is fetched once, not once per
        // 128-channel tile: 6 input rows per 4 output rows instead of 2 x 10 per 8, 5 plane pieces per chunk and wave instead of 7; the tile leaves from the accumulators.
        // Alone it ties the 128-channel tiles (1 222 / 1 221 us for 480 -> 256, 644 / 647 for 256 -> 256 at 256 frames); in the step 10.49 against 10.51-10.53 ms (two pairs).
        // GRNET_BF16_WIDE_CT256=0: 128-channel tiles
        static const int ct256_env = GRNET_AB(BF16_WIDE_CT256, 1);
        if (ct256_env && a.W == 56 && a.CoutPad == 256)
            return launch_k(conv_bf16_wide_ring<256, 56, 4, true>, dim3(a.N * RingGeom<256, 56, 4, true>::NB), dim3(512), RingGeom<256, 56, 4, true>::LDS, s, a);
        // 56x56: 8-row bands (7 per frame, 15 column tiles per wave, 244 registers) measured against 7-row ones (8 per frame): 480 -> 256 1 240 / 1 240 us,
        // 256 -> 256 648 / 666, 128 -> 128 210 / 216 at 256 frames
        if (a.W == 56) return launch_k(conv_bf16_wide_ring<128, 56, 8>, dim3(a.N * RingGeom<128, 56, 8>::NB * ncb), dim3(512), RingGeom<128, 56, 8>::LDS, s, a);
        return launch_k(conv_bf16_wide_ring<128, 28, 14>, dim3(a.N * RingGeom<128, 28, 14>::NB * ncb), dim3(512), RingGeom<128, 28, 14>::LDS, s, a);
    }
    if (a.W == 56) return launch_k(conv_bf16_wide_band<128, 128, 56, 7>, dim3(a.N * WideGeom<128, 128, 56, 7>::NB * ncb), dim3(512), WideGeom<128, 128, 56, 7>::LDS, s, a);
    return launch_k(conv_bf16_wide_band<128, 128, 28, 14>, dim3(a.N * WideGeom<128, 128, 28, 14>::NB * ncb), dim3(512), WideGeom<128, 128, 28, 14>::LDS, s, a);
}

// Stride-2 band kernel: 3x3, stride 2, even input size, <= 3 fused addends; (input channels per pass, output-channel tile) by the layer's channel counts.
static int s2_cp(const ConvArgs& a) { return a.CinPad >= 64 ? 64 : 32; }
static int s2_ct(const ConvArgs& a) { return a.CoutPad >= 128 ? 128 : a.CoutPad; }
static bool s2_rows_shape(const ConvArgs& a) {                 // the row-walking kernel's shapes (every add view 4-channel aligned: checked by the caller)
    if (!GRNET_AB(BF16_S2_ROWS, 1)) return false;
#define GRK_S2R_HAS(ci_, co_, wo_) if (a.CinPad == ci_ && a.Cin == ci_ && a.Cout == co_ && a.Wo == wo_) return true;
    GRK_S2R_SHAPES(GRK_S2R_HAS)
#undef GRK_S2R_HAS
    return false;
}
bool conv_bf16_s2_eligible(const ConvArgs& a) {
    if (a.ks != 3 || a.stride != 2 || a.H != a.W || a.Ho != a.Wo || a.H != 2 * a.Ho || a.CinPad % 32 != 0 || a.n_add > kMaxAdd || a.relu_from != 0) return false;
    if (a.in_ctot % 8 != 0 || a.in_coff % 8 != 0 || a.out_ctot % 8 != 0 || a.out_coff % 8 != 0 || a.Cout % 32 != 0 || a.CoutPad != a.Cout) return false;
    for (int k = 0; k < a.n_add; ++k)
        if (a.add_ctot[k] % 4 != 0 || a.add_coff[k] % 4 != 0) return false;
    if (s2_rows_shape(a)) return true;
    const int cp = s2_cp(a), ct = s2_ct(a);
    if (a.CinPad != cp) return false;                          // one pass (layers with more input channels lose to the generic kernel: see GRK_S2_SHAPES)
#define GRK_S2_HAS(cp_, ct_, wo_, r_) if (cp == cp_ && ct == ct_ && a.Wo == wo_) return true;
    GRK_S2_SHAPES(GRK_S2_HAS)
#undef GRK_S2_HAS
    return false;
}
hipError_t launch_conv_bf16_s2(const ConvArgs& a, hipStream_t s) {
    if (!conv_bf16_s2_eligible(a) || a.N < 1) return hipErrorInvalidValue;
    if (s2_rows_shape(a)) {
#define GRK_S2R_GO(ci_, co_, wo_) if (a.CinPad == ci_ && a.Cout == co_ && a.Wo == wo_) return launch_s2_rows<ci_, co_, wo_>(a, s);
        GRK_S2R_SHAPES(GRK_S2R_GO)
#undef GRK_S2R_GO
    }
    const int cp = s2_cp(a), ct = s2_ct(a);
#define GRK_S2_GO(cp_, ct_, wo_, r_) if (cp == cp_ && ct == ct_ && a.Wo == wo_) return launch_s2<cp_, ct_, wo_, r_>(a, s);
    GRK_S2_SHAPES(GRK_S2_GO)
#undef GRK_S2_GO
    return hipErrorInvalidValue;
}

}  // namespace grk
