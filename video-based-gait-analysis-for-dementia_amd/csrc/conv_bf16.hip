// bf16 path (grnet_create dtype = 1; BASELINE configs 3 and 5): activations and weights are bf16 in HBM, every
// convolution accumulates in fp32 on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16), the tail (pooling sums, MLPs,
// SMPL) stays fp32.  Same fused op as the fp32 path:
//     out = act( conv(in, W_folded) + b_folded + sum_k nearest_up(add_k, 2^shift_k) )
// Layout: NHWC ("channels last"), element (n, y, x, c) at ((n*H + y)*W + x)*ctot + coff + c.  The MFMA wants 8
// consecutive K elements per lane (16 bytes): with K = (tap, channel) that is 8 consecutive channels of one pixel,
// one ds_read_b128 -- NCHW planes would need eight 2-byte reads.  Roles are M = output channels, N = pixels, so a
// lane's 4 accumulator rows are 4 consecutive channels of ONE pixel: one 8-byte NHWC store.
//   A[row = cout l&15][k = 8(l>>4)+j] = W[tap][cout][cin]      (packed [cin/32][tap][CoutPad][32] bf16)
//   B[k = 8(l>>4)+j][col = pixel l&15] = patch[slot(pixel)+tap][cin]
//   D[row = 4(l>>4)+r][col = l&15]     = out[pixel][cout]
// A workgroup owns TPS*16 pixels (R whole output rows of one image, or G whole small images) x TCS*16 channels and
// walks K in chunks of 32 input channels: the zero-padded input patch and the chunk's weights go HBM/L2 -> LDS by
// LDS-DMA (no staging registers), double-buffered where two workgroups still fit a CU, one barrier per chunk.
// Patch slots are 80 bytes (64 B of data + 16 B of padding) and weight rows are XOR-swizzled, so the 16-lane b128
// operand reads fall on distinct banks.
#include "kernels.h"

#include <cstring>

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

constexpr int kCK = 32;        // input channels per K chunk (= one MFMA k-step per tap)

__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float((unsigned)h << 16); }
// two floats -> two bf16 (round to nearest even) in ONE instruction, v_cvt_pk_bf16_f32 (round 5; the integer form (u + 0x7fff + (u >> 16 & 1)) >> 16 is five vector instructions per value,
// and the epilogues of the 1x1 and narrow layers are dozens of such values per handful of MFMAs)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2)); }
// q / d for 0 <= q < 2^20 through one fp32 reciprocal multiply (exact: the +0.5 keeps exact multiples off the rounding edge);
// an integer division by a run-time value costs ~40 VALU instructions, and a tile of a 32-channel layer has only ~1000 cycles of MFMAs
__device__ __forceinline__ int fdiv(int q, float inv_d) { return (int)(((float)q + 0.5f) * inv_d); }

// LDS-DMA: lane l's 16 bytes land at lds_wave_base + 16*l; the source address is per lane.
#define GRNET_GLOBAL_AS __attribute__((address_space(1)))
#define GRNET_LDS_AS __attribute__((address_space(3)))
// (round 6) Inline asm, NOT __builtin_amdgcn_global_load_lds: while an LDS-DMA hipcc knows of is in flight, every wait it puts in front of an LDS operand read is
// lgkmcnt(0) and the reads are not hoisted -- conv_bf16_nhwc's tap loop was `ds_read, s_waitcnt lgkmcnt(0), v_mfma` 63 times over, one exposed LDS round trip per
// MFMA.  The kernel waits for its pieces itself (s_waitcnt vmcnt(0) in front of every chunk's barrier), so the compiler does not have to know.  M0 = the wave's LDS
// byte address (one wait state between its write and the DMA); every lane is on (padding units fetch zeros).
__device__ __forceinline__ void dma16(const u16* src, u16* lds_wave_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(__builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base)) : "memory");
}

// Diagnostic build only (make ABLATION=1, GRNET_BF16_PHASES=1): where a workgroup of the bf16 kernel spends its life, in shader-clock
// ticks summed over workgroups: [0] table build, [1] wait for the first chunk, [2] chunk loop (MFMAs + waits), [3] epilogue, [4] workgroups.
#ifdef GRNET_ABLATION
__device__ unsigned long long g_phase[8];
#define GRK_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#define GRK_PHASE(i, t0, t1) do { if (threadIdx.x == 0) atomicAdd(&g_phase[i], (t1) - (t0)); } while (0)
#else
#define GRK_TICK(var) do { } while (0)
#define GRK_PHASE(i, t0, t1) do { } while (0)
#endif

constexpr int kSlotU = 5;      // 16-byte DMA units per patch slot: 4 of data + 1 of padding (80-byte stride: conflict-free b128 reads)

constexpr int kRegUnits = 8;   // DMA units a thread may own for their source offsets to live in registers (REG variants)

// OCC: register budget.  true = four waves per SIMD for the 16-channel-per-wave variants (<= 128 registers), for launches with many
// workgroups per CU, where residency hides the per-workgroup latencies; false = the compiler's own choice (136-140 registers, three
// waves per SIMD, shorter code per wave), for launches of a few workgroups per CU, where one workgroup's latency is the launch's.
template <int KS, int S, int TPS, int TCS, int WP, int WC, bool REG = false, bool OCC = false>
__global__ __launch_bounds__(WP * WC * 64) __attribute__((amdgpu_waves_per_eu(TCS / WC == 1 ? (OCC ? 4 : 3) : 2)))     // 16 channels per wave: <= 128 registers (four waves per SIMD: 32 -> 32 @56x56 43 -> 40.7 us, 64 -> 256 1x1 312 -> 272), else <= 256
void conv_bf16_nhwc(const ConvArgs a) {
    constexpr int NT = WP * WC * 64, PSW = TPS / WP, CSW = TCS / WC, TC = TCS * 16, TAPS = KS * KS;
    static_assert(TPS % WP == 0 && TCS % WC == 0, "wave grid must divide the tile");
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const int aunits = (a.PSTR * kSlotU + 63) & ~63;                     // patch units per chunk, whole wave-instructions
    const int nbuf = a.nbuf;                                             // plan_bf16: 2 = double-buffered chunks, 1 = single
    u16* a_lds = reinterpret_cast<u16*>(smem_raw);                       // [nbuf][aunits][8]       80-byte slots
    int* tab = reinterpret_cast<int*>(a_lds + (size_t)nbuf * aunits * 8);   // [PSTR] input pixel of each patch slot, -1 = zero

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave / WC, wc = wave % WC, l15 = lane & 15, lq = lane >> 4;
    // XCD-aware block order (same rule as conv_kernels.hip:xcd_block): the channel blocks of one pixel tile are consecutive
    // blocks of ONE XCD (ids congruent mod 8) and an XCD owns a contiguous range of tiles, so the 2nd..gy-th read of an input
    // line hits that XCD's L2; a speed heuristic only.
    int bx, by;
    if (a.xcd) {
        const int id = blockIdx.x, j = id >> 3, x = id & 7, q = j / a.gy;
        by = j - q * a.gy;
        bx = ((x * a.gx) >> 3) + q;
        if (bx >= (((x + 1) * a.gx) >> 3)) return;
    } else {
        by = blockIdx.x / a.gx;
        bx = blockIdx.x - by * a.gx;
    }
    const int ty = bx % a.tiles_y, grp = bx / a.tiles_y;
    const int y0 = ty * a.R, g0 = grp * a.G, co0 = by * TC;
    const int HW = a.H * a.W, RW = a.R * a.Wo, RinWp = a.Rin * a.Wp, HoWo = a.Ho * a.Wo;
    constexpr int pad = KS / 2;
    const u16* in = reinterpret_cast<const u16*>(a.in);
    const u16* wg = reinterpret_cast<const u16*>(a.w);
    const u16* zeros = reinterpret_cast<const u16*>(a.zeros);

    const float inv_RinWp = 1.0f / (float)RinWp, inv_Wp = 1.0f / (float)a.Wp, inv_RW = 1.0f / (float)RW, inv_Wo = 1.0f / (float)a.Wo;
    GRK_TICK(t_start);
    auto slot_pixel = [&](int idx) {                       // input pixel of patch slot idx, -1 = zero padding / outside the batch
        const int gl = fdiv(idx, inv_RinWp), rem = idx - gl * RinWp;
        const int ry = fdiv(rem, inv_Wp), rx = rem - ry * a.Wp;
        const int yin = y0 * S + ry - pad, xin = rx - pad;
        const bool ok = gl < a.G && (g0 + gl) < a.N && yin >= 0 && yin < a.H && xin >= 0 && xin < a.W;
        return ok ? (g0 + gl) * HW + yin * a.W + xin : -1;
    };
    // REG: a thread owns the same <= kRegUnits DMA units (slot, 16-byte part q) in every chunk, so their source offsets are worked out
    // once, into registers: element offset of the slot's pixel (a multiple of 8) | q, or all ones for a unit that stays zero.  No table,
    // no barrier before the first DMA, and a chunk's staging is a compare, an add and the DMA per unit instead of an LDS round trip,
    // a branch and 64-bit address arithmetic (the per-chunk staging stood in front of every chunk's MFMAs: ~40 % of the chunk loop).
    unsigned uoff[REG ? kRegUnits : 1];
    if (REG) {
#pragma unroll
        for (int i = 0; i < kRegUnits; ++i) {
            const int u = wave * 64 + i * NT + lane, slot = (int)(((unsigned)u * 52429u) >> 18), q = u - slot * kSlotU;
            uoff[i] = 0xffffffffu;
            if (u < aunits && q < 4 && slot < a.PSTR) {
                const int px = slot_pixel(slot);
                if (px >= 0) uoff[i] = (unsigned)px * (unsigned)a.in_ctot + (unsigned)a.in_coff + (unsigned)q;
            }
        }
    } else {
        for (int idx = tid; idx < a.PSTR; idx += NT) tab[idx] = slot_pixel(idx);
        __syncthreads();
    }

    // One chunk = 32 input channels: weights [tap][TC] rows of 64 B straight from the packed [chunk][tap][CoutPad][32] array
    // (the 16-byte part a lane fetches is XOR-ed with (row >> 2) & 3, the read side applies the same involution), and the
    // zero-padded input patch, 5 DMA lanes per slot.
    auto stage = [&](int chunk, int buf) {
        u16* adst = a_lds + (size_t)buf * aunits * 8;
        const int c0 = chunk * kCK;
        if (REG) {
#pragma unroll
            for (int i = 0; i < kRegUnits; ++i) {
                const int ub = wave * 64 + i * NT;
                if (ub >= aunits) break;                     // wave-uniform
                const unsigned o = uoff[i];
                const int c = c0 + (int)(o & 7u) * 8;
                const u16* src = (o != 0xffffffffu && c < a.Cin) ? in + (size_t)(o & ~7u) + c : zeros;
#ifdef GRNET_ABLATION
                if (a.dbg & 2) src = zeros;
#endif
                dma16(src, adst + ub * 8);
            }
            return;
        }
        for (int ub = wave * 64; ub < aunits; ub += NT) {
            const int u = ub + lane, slot = (int)(((unsigned)u * 52429u) >> 18), q = u - slot * kSlotU;      // u / 5 for u < 2^16
            const u16* src = zeros;
#ifdef GRNET_ABLATION
            if (a.dbg & 2) { dma16(src, adst + ub * 8); continue; }      // timing only: no HBM reads of the patch
#endif
            if (q < 4 && slot < a.PSTR) {
                const int off = tab[slot], c = c0 + q * 8;
                if (off >= 0 && c < a.Cin) {
                    if (KS == 1 && a.in2 && c0 >= a.cin_split) src = reinterpret_cast<const u16*>(a.in2) + (size_t)off * a.in2_ctot + a.in2_coff + (c - a.cin_split);      // the chunk's second source
                    else src = in + (size_t)off * a.in_ctot + a.in_coff + c;
                }
            }
            dma16(src, adst + ub * 8);
        }
    };

    f32x4 acc[CSW][PSW];
#pragma unroll
    for (int cs = 0; cs < CSW; ++cs)
#pragma unroll
        for (int ps = 0; ps < PSW; ++ps) acc[cs][ps] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.CinPad / kCK;
    GRK_TICK(t_tab);
    stage(0, 0);
    // 1x1 layers (layer1's 64 -> 256 with its residual: 925 MB per launch at 256 frames, 3.6 TB/s): a workgroup's life is a chain of round trips --
    // patch DMA, a handful of MFMAs, addend loads, LDS transpose, stores.  The first addend does not depend on anything the workgroup computes: it
    // is requested HERE, beside the patch DMA, and is in registers when the epilogue starts (round 5).
    constexpr bool EARLY_ADD = KS == 1;
    const int cstore_e = a.out_ctot - a.out_coff < a.CoutPad ? a.out_ctot - a.out_coff : a.CoutPad;
    u32x2 radd[EARLY_ADD ? PSW : 1][EARLY_ADD ? CSW : 1];
    if (EARLY_ADD && a.n_add > 0) {
        const int sh = a.add_shift[0], hs = a.Ho >> sh, ws = a.Wo >> sh;
#pragma unroll
        for (int ps = 0; ps < PSW; ++ps) {
            const int q = (wp * PSW + ps) * 16 + l15;
            const int gl = fdiv(q, inv_RW), rem = q - gl * RW;
            const bool ok = q < a.G * RW && g0 + gl < a.N && y0 * a.Wo + rem < HoWo;
            const int pix = ok ? y0 * a.Wo + rem : 0, img = ok ? g0 + gl : 0;
            const int y = fdiv(pix, inv_Wo), x = pix - y * a.Wo;
            const u16* ap = reinterpret_cast<const u16*>(a.add[0]) + ((size_t)(img * hs + (y >> sh)) * ws + (x >> sh)) * a.add_ctot[0] + a.add_coff[0];
#pragma unroll
            for (int cs = 0; cs < CSW; ++cs) {
                const int co = co0 + (wc * CSW + cs) * 16 + lq * 4;
                radd[ps][cs] = co < cstore_e ? *reinterpret_cast<const u32x2*>(ap + co) : u32x2{0u, 0u};
            }
        }
    }

    int abase[PSW];                                        // bf16 offset of tap (0,0) of this lane's pixel + its k-group, per pixel sub-tile
#pragma unroll
    for (int ps = 0; ps < PSW; ++ps) {
        const int q = (wp * PSW + ps) * 16 + l15;
        const int gl = fdiv(q, inv_RW), rem = q - gl * RW;
        const int yl = fdiv(rem, inv_Wo), x = rem - yl * a.Wo;
        const int slot = (q < a.G * RW) ? gl * RinWp + yl * S * a.Wp + x * S : 0;     // masked pixels read slot 0, never stored
        abase[ps] = slot * (kSlotU * 8) + lq * 8;
    }
    // A operand (weights) straight from global memory: lane (cout row l15, k-group lq) of a fragment reads its 16 bytes of the packed
    // [chunk][tap][CoutPad][32] array -- 1 KB contiguous per fragment, the same for every workgroup of a channel block (L1/L2 hits).
    // No weight slab in LDS: more workgroups fit a CU and a chunk needs no weight DMA.
    const u16* wlane[CSW];
#pragma unroll
    for (int cs = 0; cs < CSW; ++cs) wlane[cs] = wg + ((size_t)co0 + (wc * CSW + cs) * 16 + l15) * 32 + lq * 8;
    const size_t wtap = (size_t)a.CoutPad * 32;            // elements between taps

#ifdef GRNET_ABLATION
    unsigned long long t_first = 0;
#endif
    // Weight fragments live in registers a chunk ahead: chunk 0's are requested here, under the first patch's flight, and every tap
    // re-requests ITS registers for the next chunk right behind the MFMAs that consumed them -- the L2 round trip of a chunk's
    // weights (1-2 us, formerly exposed after every barrier: the chunk loop ran at 4-8x its MFMA time) hides under the rest of the
    // chunk.  The last chunk re-requests itself (an L2 hit nobody waits for) so that the tap loop carries no branch.
    bf16x8 af[TAPS][CSW];
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs) af[tap][cs] = *reinterpret_cast<const bf16x8*>(wlane[cs] + tap * wtap);
    f32x4 biasv[CSW];
    {
        const int cstore0 = a.out_ctot - a.out_coff < a.CoutPad ? a.out_ctot - a.out_coff : a.CoutPad;
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs) {
            const int co = co0 + (wc * CSW + cs) * 16 + lq * 4;
            biasv[cs] = co < cstore0 ? *reinterpret_cast<const f32x4*>(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = nbuf == 2 ? (ch & 1) : 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of chunk ch has landed, and so have its weight fragments
        __syncthreads();                                     // ... and everybody else's; with two buffers the other one is free
#ifdef GRNET_ABLATION
        if (ch == 0) t_first = __builtin_readcyclecounter();
#endif
        if (nbuf == 2 && ch + 1 < nchunks) stage(ch + 1, buf ^ 1);
        const size_t wnext = (size_t)(ch + 1 < nchunks ? ch + 1 : ch) * TAPS * wtap;
        const u16* al = a_lds + (size_t)buf * aunits * 8;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int toff = ((tap / KS) * a.Wp + (tap % KS)) * (kSlotU * 8);
            bf16x8 bfr[PSW];
#pragma unroll
            for (int ps = 0; ps < PSW; ++ps) bfr[ps] = *reinterpret_cast<const bf16x8*>(al + abase[ps] + toff);
#pragma unroll
            for (int cs = 0; cs < CSW; ++cs)
#pragma unroll
                for (int ps = 0; ps < PSW; ++ps) acc[cs][ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[tap][cs], bfr[ps], acc[cs][ps], 0, 0, 0);
#pragma unroll
            for (int cs = 0; cs < CSW; ++cs) af[tap][cs] = *reinterpret_cast<const bf16x8*>(wlane[cs] + wnext + tap * wtap);
        }
        if (nbuf == 1 && ch + 1 < nchunks) {
            __syncthreads();                                 // single buffer: everybody is done reading before it is refilled
            stage(ch + 1, 0);
        }
    }

    GRK_TICK(t_loop);
    // ---- epilogue: lane holds channels co..co+3 of pixel l15 of each sub-tile.  Every load (bias, addends) is issued before the
    // first store: loads and stores share vmcnt on CDNA, so a load behind a store waits for that store's round trip to memory
    // (measured with the phase ticks above: 0.75 us per tile when the bias was loaded tile by tile).
    u16* out = reinterpret_cast<u16*>(a.out);
    const int cstore = a.out_ctot - a.out_coff < a.CoutPad ? a.out_ctot - a.out_coff : a.CoutPad;
    int img_[PSW], pix_[PSW];                              // pix_ < 0: masked pixel
#pragma unroll
    for (int ps = 0; ps < PSW; ++ps) {
        const int q = (wp * PSW + ps) * 16 + l15;
        const int gl = fdiv(q, inv_RW), rem = q - gl * RW;
        img_[ps] = g0 + gl;
        pix_[ps] = (q < a.G * RW && img_[ps] < a.N && y0 * a.Wo + rem < HoWo) ? y0 * a.Wo + rem : -1;
    }
#pragma unroll
    for (int k = 0; k < kMaxAdd; ++k) {
        if (k >= a.n_add) break;
        const int sh = a.add_shift[k], hs = a.Ho >> sh, ws = a.Wo >> sh;
        u32x2 r[PSW][CSW];
        if (EARLY_ADD && k == 0) {
#pragma unroll
            for (int ps = 0; ps < PSW; ++ps)
#pragma unroll
                for (int cs = 0; cs < CSW; ++cs) r[ps][cs] = radd[ps][cs];
        } else
#pragma unroll
        for (int ps = 0; ps < PSW; ++ps) {
            const int pix = pix_[ps] < 0 ? 0 : pix_[ps], img = pix_[ps] < 0 ? 0 : img_[ps];
            const int y = fdiv(pix, inv_Wo), x = pix - y * a.Wo;
            const u16* ap = reinterpret_cast<const u16*>(a.add[k]) + ((size_t)(img * hs + (y >> sh)) * ws + (x >> sh)) * a.add_ctot[k] + a.add_coff[k];
#pragma unroll
            for (int cs = 0; cs < CSW; ++cs) {
                const int co = co0 + (wc * CSW + cs) * 16 + lq * 4;
                r[ps][cs] = co < cstore ? *reinterpret_cast<const u32x2*>(ap + co) : u32x2{0u, 0u};
            }
        }
#pragma unroll
        for (int ps = 0; ps < PSW; ++ps)
#pragma unroll
            for (int cs = 0; cs < CSW; ++cs) {
                acc[cs][ps][0] += bf2f((u16)(r[ps][cs][0] & 0xffffu)); acc[cs][ps][1] += bf2f((u16)(r[ps][cs][0] >> 16));
                acc[cs][ps][2] += bf2f((u16)(r[ps][cs][1] & 0xffffu)); acc[cs][ps][3] += bf2f((u16)(r[ps][cs][1] >> 16));
            }
    }
    // The finished tile goes through LDS ([pixel][TC channels], 16 B of padding per pixel) and leaves as 16 bytes per lane with the
    // lanes walking the channels of a pixel and then the next pixel: whole 64/128-byte channel rows per request instead of sixteen
    // 32-byte pieces per store instruction.  (Per-workgroup phase ticks: with the direct 8-byte stores the epilogue was 33-57 % of a
    // small layer's workgroup life -- only ~4 such stores per wave are in flight for free, every further one waits for a drain.)
    constexpr int TCP = TC + 8;                            // bf16 per pixel in LDS
    u16* o_lds = a_lds;                                    // the patch buffers are free after the barrier below
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < PSW; ++ps) {
        const int pl = (wp * PSW + ps) * 16 + l15;
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs) {
            f32x4 v = acc[cs][ps] + biasv[cs];
            if (a.relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            *reinterpret_cast<u32x2*>(o_lds + pl * TCP + (wc * CSW + cs) * 16 + lq * 4) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
        }
    }
    __syncthreads();
    constexpr int UPP = TC / 8;                            // 16-byte units per pixel
    for (int u = tid; u < TPS * 16 * UPP; u += NT) {
        const int pl = u / UPP, part = u - pl * UPP;
        if (pl >= a.G * RW) continue;
        const int gl = fdiv(pl, inv_RW), rem = pl - gl * RW;
        const int img = g0 + gl, pix = y0 * a.Wo + rem;
        if (img >= a.N || pix >= HoWo || co0 + part * 8 >= cstore) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(o_lds + pl * TCP + part * 8);
#ifdef GRNET_ABLATION
        if ((a.dbg & 4) && v[0] != 0x12345678u) continue;               // timing only: no stores
#endif
        *reinterpret_cast<u32x4*>(out + ((size_t)img * HoWo + pix) * a.out_ctot + a.out_coff + co0 + part * 8) = v;
    }
    // ---- second stage of a 1x1 PAIR (round 5; layer1, hrnet.py:80-100): the workgroup holds all 256 output channels of its 112 pixels in LDS, bf16, exactly
    // as the next Bottleneck's 256 -> 64 reduction would read them from HBM -- so it runs that reduction now: 16 output channels per wave, K = 256 in 8
    // k-steps, B = the tile (one ds_read_b128 per pixel tile and step), A = the reduction's weights from L2.  The 411 MB tensor (256 frames) is still
    // written (it is the next block's residual) but is not read back for the reduction.  Same k order and operands as the stand-alone launch: bit-identical.
    if constexpr (KS == 1 && TCS == 16 && WP == 1 && WC == 4) {
        if (a.w2) {
            const u16* w2 = reinterpret_cast<const u16*>(a.w2);
            f32x4 acc2[PSW];
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(a.bias2 + wave * 16 + lq * 4);
#pragma unroll
            for (int ps = 0; ps < PSW; ++ps) acc2[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 af2[TC / 32];
#pragma unroll
            for (int ch = 0; ch < TC / 32; ++ch) af2[ch] = *reinterpret_cast<const bf16x8*>(w2 + ((size_t)ch * 64 + wave * 16 + l15) * 32 + lq * 8);
#pragma unroll
            for (int ch = 0; ch < TC / 32; ++ch)
#pragma unroll
                for (int ps = 0; ps < PSW; ++ps) {
                    const bf16x8 bt = *reinterpret_cast<const bf16x8*>(o_lds + (ps * 16 + l15) * TCP + ch * 32 + lq * 8);
                    acc2[ps] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af2[ch], bt, acc2[ps], 0, 0, 0);
                }
            u16* out2 = reinterpret_cast<u16*>(a.out2);
#pragma unroll
            for (int ps = 0; ps < PSW; ++ps) {
                if (pix_[ps] < 0) continue;
                f32x4 v = acc2[ps] + b2;
                if (a.relu2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                *reinterpret_cast<u32x2*>(out2 + ((size_t)img_[ps] * HoWo + pix_[ps]) * a.out2_ctot + a.out2_coff + wave * 16 + lq * 4) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
            }
        }
    }
#ifdef GRNET_ABLATION
    if (a.dbg & 8) {                                       // phase accounting on request only: 5 atomics per workgroup on one line distort the timing
        GRK_TICK(t_end);
        GRK_PHASE(0, t_start, t_tab);
        GRK_PHASE(1, t_tab, t_first);
        GRK_PHASE(2, t_first, t_loop);
        GRK_PHASE(3, t_loop, t_end);
        if (threadIdx.x == 0) atomicAdd(&g_phase[4], 1ull);
    }
#endif
}

// ---- Register-resident kernel for the narrow 3x3 stride-1 layers with as many output as input channels (C = 32 on 56x56 maps, C = 64
// on 28x28 and 56x56 maps: the BasicBlocks of HR branches 0 and 1 and of layer1's bottlenecks).  An ablation of conv_bf16_nhwc on
// these layers (patch DMA from a zero block, stores skipped: tools/bf16_micro.py on the diagnostic build) leaves 80-90 % of their
// time standing: they are bound by that kernel's per-workgroup structure -- slot table, barrier, LDS-DMA issue loop, wait, LDS reads
// with a wait in front of every MFMA group, output through LDS with two barriers -- not by HBM.  Here nothing goes through the LDS
// and there is no barrier:
//   wave  = one output row of one frame x 32 output channels; its weights of ALL taps (9 x C/32 x 2 fragments = 72 / 144 registers)
//           are loaded once and stay in registers for the row;
//   tile  = 16 loaded pixels xs .. xs+15 of an input row, of which the middle 14 are outputs (a 56- / 28-wide row is 4 / 2 tiles):
//           lane (pixel l15, k-group lq) loads the 16 bytes = 8 channels of ITS pixel once per filter row and 32-channel chunk; that
//           IS the B operand of the centre tap (NHWC: K = 8 consecutive channels per lane), and the left / right taps are the same
//           registers shifted by one lane within the 16-lane row (DPP row_shr / row_shl: 4 moves per operand);
//   zero padding: rows outside the image are not loaded (wave-uniform), pixels -1 and W are a select on lanes 0 / 15 of the edge tiles;
//   output: D[row = channel 4 lq + r][col = pixel l15] -> bias, residual (8 bytes per lane), ReLU, 8-byte stores of 4 channels;
//           lanes 0 and 15 hold no output.  The next tile's loads are requested before the current tile's MFMAs.
//   A wave walks a strip of R consecutive output rows with the three input rows it needs in registers: after tile t of row y is
//   done, the registers of its TOP row (y - 1, not needed again) are re-requested with row y + 2, which is the bottom row of output
//   row y + 1 -- one new row of loads per output row instead of three, requested a whole row of work ahead, and the weights are
//   loaded once per strip instead of once per row (a first version, one row per wave with the next TILE's loads in flight, ran at
//   the LDS kernel's speed: 18 KB of weights per 3.5 KB row and a 0.2 us prefetch distance).  The roles of the three row buffers
//   rotate, so the row loop is unrolled by three.
template <int C, int NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C == 32 ? 3 : 2))) void conv_bf16_direct(const ConvArgs a, int R, int strips_per_frame) {
    constexpr int NCH = C / 32, HALVES = C / 32;
    constexpr bool RES_AHEAD = C == 32;                    // residual of row y + 1 requested during row y (8 registers; the 64-channel variant has none to spare)
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), half = gw % HALVES, strip = gw / HALVES;
    const int cols = a.W / (14 * NT);                      // column parts of a row: a strip is R rows x NT tiles (56-wide maps with NT = 2: two parts)
    const int n = strip / (strips_per_frame * cols), sr = strip - n * strips_per_frame * cols;
    const int y0 = (sr / cols) * R, t0 = (sr % cols) * NT;
    if (n >= a.N || y0 >= a.H) return;
    const int y1 = y0 + R < a.H ? y0 + R : a.H, co0 = half * 32;
    const u16* wg = reinterpret_cast<const u16*>(a.w);
    bf16x8 wq[NCH][9][2];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                wq[ch][tap][mt] = *reinterpret_cast<const bf16x8*>(wg + ((size_t)(ch * 9 + tap) * a.CoutPad + co0 + mt * 16 + l15) * 32 + lq * 8);
    f32x4 biasv[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) biasv[mt] = *reinterpret_cast<const f32x4*>(a.bias + co0 + mt * 16 + lq * 4);
    const u16* inb = reinterpret_cast<const u16*>(a.in) + (size_t)n * a.H * a.W * a.in_ctot + a.in_coff + lq * 8;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    typedef u32x4 RowBuf[NT][NCH];
    auto load_tile = [&](int yin, int t, u32x4 (&dst)[NCH]) {     // this lane's 8 channels of pixel 14 t - 1 + l15 of input row yin, per chunk
        const int x = 14 * (t0 + t) - 1 + l15;
        const bool ok = yin >= 0 && yin < a.H && x >= 0 && x < a.W;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            dst[ch] = zero4;
            if (ok) dst[ch] = *reinterpret_cast<const u32x4*>(inb + ((size_t)yin * a.W + x) * a.in_ctot + ch * 32);
        }
    };
    u16* outb = reinterpret_cast<u16*>(a.out) + (size_t)n * a.H * a.W * a.out_ctot + a.out_coff + co0 + lq * 4;
    const u16* addb = a.n_add ? reinterpret_cast<const u16*>(a.add[0]) + (size_t)n * a.H * a.W * a.add_ctot[0] + a.add_coff[0] + co0 + lq * 4 : nullptr;
    const bool olane = l15 >= 1 && l15 <= 14;
    u32x2 resrow[RES_AHEAD ? NT : 1][2];                   // RES_AHEAD: the residual of the row about to be computed
    auto load_res = [&](int y, int t, u32x2 (&dst)[2]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            dst[mt] = u32x2{0u, 0u};
            if (addb && olane && y < y1) dst[mt] = *reinterpret_cast<const u32x2*>(addb + ((size_t)y * a.W + 14 * (t0 + t) - 1 + l15) * a.add_ctot[0] + mt * 16);
        }
    };
    if (RES_AHEAD) {
#pragma unroll
        for (int t = 0; t < NT; ++t) load_res(y0, t, resrow[t]);
    }
    auto do_row = [&](int y, RowBuf& top, RowBuf& mid, RowBuf& bot) {
        u32x2 res[2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (RES_AHEAD) { res[0] = resrow[t][0]; res[1] = resrow[t][1]; load_res(y + 1, t, resrow[t]); }
            else load_res(y, t, res);                        // under this tile's 36 MFMAs
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const u32x4 c = ky == 0 ? top[t][ch] : ky == 1 ? mid[t][ch] : bot[t][ch];
                    u32x4 lft, rgt;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        lft[k] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c[k], 0x111, 0xf, 0xf, true);      // row_shr:1: pixel x - 1
                        rgt[k] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)c[k], 0x101, 0xf, 0xf, true);      // row_shl:1: pixel x + 1
                    }
                    const bf16x8 b0 = __builtin_bit_cast(bf16x8, lft), b1 = __builtin_bit_cast(bf16x8, c), b2 = __builtin_bit_cast(bf16x8, rgt);
                    // the two channel blocks alternate: back-to-back MFMAs into one accumulator wait for each other
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ch][ky * 3 + 0][0], b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ch][ky * 3 + 0][1], b0, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ch][ky * 3 + 1][0], b1, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ch][ky * 3 + 1][1], b1, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ch][ky * 3 + 2][0], b2, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[ch][ky * 3 + 2][1], b2, acc[1], 0, 0, 0);
                }
            if (y + 1 < y1) load_tile(y + 2, t, top[t]);           // row y - 1 is done with: its registers take the bottom row of output row y + 1
            if (olane) {
                const int x = 14 * (t0 + t) - 1 + l15;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x4 v = acc[mt] + biasv[mt];
                    v[0] += bf2f((u16)(res[mt][0] & 0xffffu)); v[1] += bf2f((u16)(res[mt][0] >> 16));
                    v[2] += bf2f((u16)(res[mt][1] & 0xffffu)); v[3] += bf2f((u16)(res[mt][1] >> 16));
                    if (a.relu) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    *reinterpret_cast<u32x2*>(outb + ((size_t)y * a.W + x) * a.out_ctot + mt * 16) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
                }
            }
        }
    };
    RowBuf A, B, Cb;
#pragma unroll
    for (int t = 0; t < NT; ++t) { load_tile(y0 - 1, t, A[t]); load_tile(y0, t, B[t]); load_tile(y0 + 1, t, Cb[t]); }
    for (int y = y0; y < y1; y += 3) {
        do_row(y, A, B, Cb);
        if (y + 1 < y1) do_row(y + 1, B, Cb, A);
        if (y + 2 < y1) do_row(y + 2, Cb, A, B);
    }
}

// ---- The stem's first convolution on the bf16 path (3 -> 64, 3x3, stride 2, 224 -> 112; hrnet.py:470-475), round 4.  Through the generic kernel
// it read an NHWC image padded from 3 to 8 channels and multiplied 27 real of 288 K elements: 604 us at 256 frames (0.01 of the matrix peak, 4 % of
// the step) behind a 55 us conversion launch.  Here K = (channel, tap) is flattened to ONE 32-wide MFMA k-step (27 real), and the kernel reads the
// caller's fp32 NCHW frames itself (the conversion launch and its buffer are gone): lane (pixel l15, k-group lq) gathers its 8 K elements -- 4-byte
// loads, 16 lanes covering 128 contiguous bytes of a row at stride 2 -- rounds them to bf16 (nearest even, as the conversion kernel did) and that
// is the B operand; A = the 64 x 32 weight block in 16 registers; D -> bias, ReLU, NHWC bf16, 8 bytes per lane and channel block.  A wave owns one
// output row (7 tiles of 16 pixels); the next tile's gathers are requested before the current tile's four MFMAs.  Memory-bound: 0.6 MB read and
// 1.6 MB written per frame.
__global__ __launch_bounds__(256) void conv_bf16_stem(const float* __restrict__ frames, const u16* __restrict__ wpk, const float* __restrict__ bias,
                                                      u16* __restrict__ out, int out_ctot, int out_coff, int N, int relu) {
    constexpr int H = 224, W = 224, HO = 112, WO = 112;
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);          // (frame, output row)
    if (row >= N * HO) return;
    const int n = row / HO, y = row - n * HO;
    bf16x8 wq[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) wq[mt] = *reinterpret_cast<const bf16x8*>(wpk + ((size_t)mt * 64 + lane) * 8);
    f32x4 biasv[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) biasv[mt] = *reinterpret_cast<const f32x4*>(bias + mt * 16 + lq * 4);
    // this lane's 8 K elements: k = 8 lq + j = channel * 9 + ky * 3 + kx (k >= 27: zero)
    int off[8];
    bool rowok[8], real[8], left[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * lq + j, c = k / 9, tap = k - 9 * c, ky = tap / 3, kx = tap - 3 * ky;
        real[j] = k < 27;
        rowok[j] = real[j] && (2 * y + ky - 1 >= 0);               // the bottom row 2 * 111 + 1 = 223 exists; only the top row is padding
        left[j] = kx == 0;                                         // column 2 x - 1 is padding for x = 0
        off[j] = (c * H + (2 * y + ky - 1)) * W + (kx - 1);
    }
    const float* fb = frames + (size_t)n * 3 * H * W;
    auto gather = [&](int t, float (&v)[8]) {
        const int x = 16 * t + l15;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = rowok[j] && !(left[j] && x == 0);
            v[j] = ok ? fb[off[j] + 2 * x] : 0.f;
        }
    };
    float cur[8], nxt[8];
    gather(0, cur);
    // Round 5: the tile leaves through a wave-private 2 KB of LDS ([16 pixels][64 channels + 16 B]) as 16 bytes per lane, 8 lanes per pixel: every store
    // instruction writes eight whole 128-byte pixel rows.  The direct form -- 8 bytes per lane and channel block, four partial writes per row -- held the
    // kernel at 1.75 TB/s of its 411 MB of output (235 us at 256 frames).  No barrier: a wave's LDS operations complete in order.
    __shared__ __align__(16) u16 tile[4][16 * 72];
    u16* tw = tile[threadIdx.x >> 6];
    u16* ob = out + ((size_t)(n * HO + y) * WO) * out_ctot + out_coff;
#pragma unroll 1
    for (int t = 0; t < WO / 16; ++t) {
        if (t + 1 < WO / 16) gather(t + 1, nxt);
        const u32x4 bp = {pack2(cur[0], cur[1]), pack2(cur[2], cur[3]), pack2(cur[4], cur[5]), pack2(cur[6], cur[7])};
        const bf16x8 b = __builtin_bit_cast(bf16x8, bp);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            f32x4 v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[mt], b, biasv[mt], 0, 0, 0);
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            *reinterpret_cast<u32x2*>(tw + l15 * 72 + mt * 16 + lq * 4) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                  // 16 pixels x 8 units of 16 bytes = 128 units: two per lane
            const int u = h * 64 + lane, px = u >> 3, part = u & 7;
            *reinterpret_cast<u32x4*>(ob + (size_t)(16 * t + px) * out_ctot + part * 8) = *reinterpret_cast<const u32x4*>(tw + px * 72 + part * 8);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) cur[j] = nxt[j];
    }
}

bool bf16_direct_eligible(const ConvArgs& a) {
    return a.ks == 3 && a.stride == 1 && a.Cin == a.Cout && (a.Cin == 32 || a.Cin == 64) && a.CinPad == a.Cin && a.CoutPad >= a.Cout && (a.W == 28 || (a.W == 56 && a.Cin == 32)) &&
           a.Ho == a.H && a.Wo == a.W && a.n_add <= 1 && (a.n_add == 0 || a.add_shift[0] == 0) && a.out_ctot - a.out_coff >= a.Cout &&
           a.in_ctot % 8 == 0 && a.in_coff % 8 == 0 && a.out_ctot % 4 == 0 && a.out_coff % 4 == 0 && (a.n_add == 0 || (a.add_ctot[0] % 4 == 0 && a.add_coff[0] % 4 == 0));
}

// (N,C,H,W) f32 -> (N,H,W,Cp) bf16, channels C..Cp-1 zero.  The caller's frames (C = 3 -> 8) and the test hooks.  A thread owns 8 stored
// channels of one pixel: its reads walk each channel plane with the lanes (coalesced), its write is one 16-byte store.
__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_bf16_kernel(const float* __restrict__ in, u16* __restrict__ out, int C, int HW, int Cp, long total) {
    const int c8n = Cp / 8;                                            // 16-byte groups per pixel
    const long units = total / 8;                                      // (n, pixel, group)
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long)gridDim.x * 256) {
        long px = u, g = 0;
        if (c8n > 1) { px = u / c8n; g = u - px * c8n; }
        const long n = px / HW, hw = px - n * HW;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int c = (int)g * 8 + k;
            v[k] = c < C ? in[(n * C + c) * HW + hw] : 0.f;
        }
        *reinterpret_cast<u32x4*>(out + u * 8) = u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    }
}
// (N,H,W,ctot)[coff .. coff+C) bf16 -> (N,C,H,W) f32 (debug taps, test hooks)
__global__ __launch_bounds__(256) void nhwc_bf16_to_nchw_f32_kernel(const u16* __restrict__ in, float* __restrict__ out, int C, int HW, int ctot, int coff,
                                                                      long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long hw = i % HW, nc = i / HW, n = nc / C, c = nc - n * C;
        out[i] = bf2f(in[(n * HW + hw) * ctot + coff + c]);
    }
}

// nn.Upsample(scale_factor=2, bilinear, align_corners=True) on NHWC bf16 (hrnet.py:443); one workgroup = one output row of one frame,
// one thread = 8 channels of one output pixel (32-bit index arithmetic: the flat 64-bit index cost four 64-bit divisions per thread).
// (fp contract off: hipcc's default fuses fy - y0 = yo * sy - y0 into ONE fma on the unrounded product -- a last-bit difference in the tap weight that the
// reference's separate multiply, truncate and subtract (UpSampleBilinear2d: h1r = rheight * h2; h1lambda = h1r - h1) does not have)
__global__ __launch_bounds__(256) void bilinear2x_bf16_kernel(const u16* __restrict__ in, u16* __restrict__ out, int N, int C, int H, int W) {
#pragma clang fp contract(off)
    const int Ho = 2 * H, Wo = 2 * W, C8 = C / 8;
    // workgroups are dealt to the 8 XCDs round-robin: with consecutive output rows on consecutive ids every input row (shared by ~4 output rows) was fetched into four
    // L2s.  Re-dealt so that an XCD walks a contiguous range of rows (round 5)
    const int bid = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int n = bid / Ho, yo = bid - n * Ho;
    const float sy = (float)(H - 1) / (float)(Ho - 1), sx = (float)(W - 1) / (float)(Wo - 1);
    const float fy = __fmul_rn((float)yo, sy);
    const int y0 = (int)fy, y1 = y0 + 1 < H ? y0 + 1 : H - 1;
    const float wy = fy - (float)y0;
    const u16* r0 = in + ((size_t)n * H + y0) * W * C;
    const u16* r1 = in + ((size_t)n * H + y1) * W * C;
    u16* orow = out + ((size_t)n * Ho + yo) * Wo * C;
    const float inv_c8 = 1.0f / (float)C8;
    for (int i = threadIdx.x; i < Wo * C8; i += 256) {
        const int xo = fdiv(i, inv_c8), c8 = i - xo * C8;
        const float fx = __fmul_rn((float)xo, sx);
        const int x0 = (int)fx, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
        const float wx = fx - (float)x0;
        const u32x4 v00 = *reinterpret_cast<const u32x4*>(r0 + (size_t)x0 * C + c8 * 8), v01 = *reinterpret_cast<const u32x4*>(r0 + (size_t)x1 * C + c8 * 8);
        const u32x4 v10 = *reinterpret_cast<const u32x4*>(r1 + (size_t)x0 * C + c8 * 8), v11 = *reinterpret_cast<const u32x4*>(r1 + (size_t)x1 * C + c8 * 8);
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float r[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float a00 = bf2f((u16)(v00[k] >> (16 * h))), a01 = bf2f((u16)(v01[k] >> (16 * h)));
                const float a10 = bf2f((u16)(v10[k] >> (16 * h))), a11 = bf2f((u16)(v11[k] >> (16 * h)));
                const float top = __fmaf_rn(wx, a01 - a00, a00), bot = __fmaf_rn(wx, a11 - a10, a10);
                r[h] = __fmaf_rn(wy, bot - top, top);
            }
            o[k] = pack2(r[0], r[1]);
        }
        *reinterpret_cast<u32x4*>(orow + (size_t)i * 8) = o;
    }
}

// The same with the two input rows of a workgroup staged in LDS (round 5): the kernel above pulls 64 bytes through the CU's vector L1 for every 16 it writes (four taps
// per output unit: 3.1 TB/s of algorithmic bytes at 256 frames).  Here a workgroup = one input row pair (y0, y0 + 1) of one frame: both rows come in once, as whole
// rows of 16-byte units, and the output rows whose upper tap row is y0 (about two) are interpolated from LDS.  Same arithmetic per element (bit-identical).
__global__ __launch_bounds__(256) void bilinear2x_bf16_rows_kernel(const u16* __restrict__ in, u16* __restrict__ out, int N, int C, int H, int W) {
#pragma clang fp contract(off)
    extern __shared__ __align__(16) u32x4 brow[];              // [2][W * C / 8]
    const int Ho = 2 * H, Wo = 2 * W, C8 = C / 8, units = W * C8;
    const int bid = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;     // an XCD walks a contiguous range of rows
    const int n = bid / H, y0 = bid - n * H, y1 = y0 + 1 < H ? y0 + 1 : H - 1;
    const u32x4* r0 = reinterpret_cast<const u32x4*>(in + ((size_t)n * H + y0) * W * C);
    const u32x4* r1 = reinterpret_cast<const u32x4*>(in + ((size_t)n * H + y1) * W * C);
    for (int u = threadIdx.x; u < units; u += 256) {
        brow[u] = r0[u];
        brow[units + u] = r1[u];
    }
    __syncthreads();
    const float sy = (float)(H - 1) / (float)(Ho - 1), sx = (float)(W - 1) / (float)(Wo - 1);
    const float inv_c8 = 1.0f / (float)C8;
    for (int yo = 2 * y0 - 1 < 0 ? 0 : 2 * y0 - 1; yo <= 2 * y0 + 3 && yo < Ho; ++yo) {      // (+ 3: the last row's fy = (Ho - 1) * sy may round to just below H - 1)
        const float fy = __fmul_rn((float)yo, sy);
        if ((int)fy != y0) continue;                           // (uniform) this row's upper tap row belongs to another workgroup
        const float wy = fy - (float)y0;
        u16* orow = out + ((size_t)n * Ho + yo) * Wo * C;
        for (int i = threadIdx.x; i < Wo * C8; i += 256) {
            const int xo = fdiv(i, inv_c8), c8 = i - xo * C8;
            const float fx = __fmul_rn((float)xo, sx);
            const int x0 = (int)fx, x1 = x0 + 1 < W ? x0 + 1 : W - 1;
            const float wx = fx - (float)x0;
            const u32x4 v00 = brow[x0 * C8 + c8], v01 = brow[x1 * C8 + c8], v10 = brow[units + x0 * C8 + c8], v11 = brow[units + x1 * C8 + c8];
            u32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float r[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float a00 = bf2f((u16)(v00[k] >> (16 * h))), a01 = bf2f((u16)(v01[k] >> (16 * h)));
                    const float a10 = bf2f((u16)(v10[k] >> (16 * h))), a11 = bf2f((u16)(v11[k] >> (16 * h)));
                    const float top = __fmaf_rn(wx, a01 - a00, a00), bot = __fmaf_rn(wx, a11 - a10, a10);
                    r[h] = __fmaf_rn(wy, bot - top, top);
                }
                o[k] = pack2(r[0], r[1]);
            }
            *reinterpret_cast<u32x4*>(orow + (size_t)i * 8) = o;
        }
    }
}

// out = relu?( sum_k nearest_up(add_k) ) on NHWC bf16 (hrnet.py:258-265, output 0 of a fuse layer); one workgroup = one row of one
// frame, one thread = 8 channels of a pixel (32-bit index arithmetic).
__global__ __launch_bounds__(256) void fuse_sum_bf16_kernel(const SumArgs a) {
    const int C8 = a.C / 8, rpb = a.H % 4 == 0 ? 4 : 1;        // rows per workgroup (a 32-channel row of 56 pixels is only 224 units)
    const int bid = (gridDim.x & 7) == 0 ? (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;     // an XCD walks a contiguous range of rows: the low-resolution addends' rows are shared by 2-8 output rows
    const int n = (bid * rpb) / a.H, yb = bid * rpb - n * a.H;
    u16* out = reinterpret_cast<u16*>(a.out);
    const float inv_c8 = 1.0f / (float)C8, inv_wc = 1.0f / (float)(a.W * C8);
    const int total = rpb * a.W * C8;
    // four units per thread and pass, ALL their loads requested before the first sum (round 6: one unit per iteration left a thread with four 16-byte loads in
    // flight, the launch at 2.3 TB/s -- 46 us for the 105 MB of the 56x56 output; the sums and their order are unchanged)
    for (int base = 0; base < total; base += 4 * 256) {
        u32x4 v[4][4];
        int yy[4], xx[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i0 = base + u * 256 + (int)threadIdx.x, ic = i0 < total ? i0 : total - 1;      // (past the end: the last unit again, not stored)
            const int yr = fdiv(ic, inv_wc), i = ic - yr * a.W * C8;
            yy[u] = yb + yr;
            xx[u] = fdiv(i, inv_c8);
            cc[u] = i - xx[u] * C8;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[u][k] = u32x4{0u, 0u, 0u, 0u};
                if (k < a.n_add) {
                    const int sh = a.add_shift[k], hs = a.H >> sh, ws = a.W >> sh;
                    v[u][k] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const u16*>(a.add[k]) + ((size_t)(n * hs + (yy[u] >> sh)) * ws + (xx[u] >> sh)) * a.add_ctot[k] + a.add_coff[k] + cc[u] * 8);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k >= a.n_add) break;
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[2 * j] += bf2f((u16)(v[u][k][j] & 0xffffu)); acc[2 * j + 1] += bf2f((u16)(v[u][k][j] >> 16)); }
            }
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float lo = a.relu ? fmaxf(acc[2 * j], 0.f) : acc[2 * j], hi = a.relu ? fmaxf(acc[2 * j + 1], 0.f) : acc[2 * j + 1];
                o[j] = pack2(lo, hi);
            }
            if (base + u * 256 + (int)threadIdx.x < total) *reinterpret_cast<u32x4*>(out + (((size_t)n * a.H + yy[u]) * a.W + xx[u]) * a.out_ctot + a.out_coff + cc[u] * 8) = o;
        }
    }
}

// (attn_pool_bf16x_kernel below is the form that runs since round 5: the same workgroup, but the GEMM on the bf16 matrix cores -- the probabilities as hi + lo bf16 pairs,
// the features transposed out of their NHWC rows by ds_read_b64_tr_b16 -- and the range's features requested at once: 254 -> 171 us at 256 frames, results equal to 1e-6.
// This fp32-MFMA form stays as the A/B reference, GRNET_BF16_POOL_X16=0.)
// Attention pooling on NHWC bf16 maps (keypoint_attention.py:42-48): per frame the GEMM out[c][j] = sum_p feat[p][c] * prob[p][j] on the
// fp32 matrix cores, the structure of the fp32 path's attn_pool_kernel (head_kernels.hip): workgroup = (frame, 96 channels, one of
// kPoolSplit position ranges) = 6 waves; the range's exp(h - range max) is built once per workgroup in LDS (fp32) and the softmax over
// all positions is finished by head_tail_kernel from the (max, sum) pairs of the ranges.  NHWC makes the position axis the slow one, so
// a lane's A operand of a k-step is ONE bf16 (channel l15 of position p0 + 4 lq + s: 16 lanes = 32 contiguous bytes of a pixel), widened
// to fp32 in the register; the products are exact and the sums fp32, as in the vector-ALU kernel this replaces (284 + 137 us for the
// pooling and its statistics pass at 256 frames).
constexpr int kPoolChunkB = 448, kPoolStrideB = kPoolChunkB + 4;
__device__ __forceinline__ float wave_max_b(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum_b(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
constexpr int kPoolPB = 64;                                // positions per staging buffer (448 = 7 x 64)
constexpr int kPoolFS = 96 + 8;                            // bf16 per staged position: 96 channels + 16 bytes (the four positions of a k-step land on different banks)
__global__ __launch_bounds__(384) void attn_pool_bf16_kernel(const u16* __restrict__ heat, int hc, const u16* __restrict__ featA, int CA, int ctA,
                                                               const u16* __restrict__ featB, int CB, int ctB, float* __restrict__ stats,
                                                               float* __restrict__ part, int P) {
    __shared__ __align__(16) float prob[24 * kPoolStrideB];
    __shared__ __align__(16) u16 fst[2][kPoolPB * kPoolFS];    // two buffers of 64 positions x 96 channels, as they lie in memory (NHWC); 70 KB of LDS in all
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int pbeg = blockIdx.z * kPoolChunkB, cb = blockIdx.y * 96;      // first channel of this workgroup in [featA | featB]
    // staging: 64 positions x 12 units of 16 bytes = two units per thread (positions spp and spp + 32); a unit lies entirely in featA or in featB
    // (128 = 8 x 16).  Round 5: 64 instead of 32 positions per barrier -- the loop is a chain of load -> LDS -> barrier -> 16 MFMAs round trips
    const int spp = tid / 12, sq = tid - spp * 12, sc = cb + sq * 8;
    const u16* ssrc = sc < CA ? featA + ((size_t)n * P + pbeg + spp) * ctA + sc : featB + ((size_t)n * P + pbeg + spp) * ctB + (sc - CA);
    const size_t sstride = sc < CA ? ctA : ctB;
    auto stage = [&](int p0, int buf) {
        const u32x4 v0 = *reinterpret_cast<const u32x4*>(ssrc + (size_t)p0 * sstride), v1 = *reinterpret_cast<const u32x4*>(ssrc + (size_t)(p0 + 32) * sstride);
        *reinterpret_cast<u32x4*>(&fst[buf][spp * kPoolFS + sq * 8]) = v0;
        *reinterpret_cast<u32x4*>(&fst[buf][(spp + 32) * kPoolFS + sq * 8]) = v1;
    };
    stage(0, 0);
    // heat rows of the range -> LDS: thread = (position, 8 joints)
    for (int u = tid; u < kPoolChunkB * 3; u += 384) {
        const int p = u / 3, jg = u - p * 3;
        const u32x4 h8 = *reinterpret_cast<const u32x4*>(heat + ((size_t)n * P + pbeg + p) * hc + jg * 8);      // channels 8 jg .. 8 jg + 7 (channel 0 = background)
        const u16 nx = heat[((size_t)n * P + pbeg + p) * hc + jg * 8 + 8];                                          // channel 8 jg + 8 = joint 8 jg + 7
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = k + 1;                                                                                   // joint 8 jg + k is channel 8 jg + k + 1
            const u16 v = ch < 8 ? (u16)(h8[ch >> 1] >> (16 * (ch & 1))) : nx;
            prob[(jg * 8 + k) * kPoolStrideB + p] = bf2f(v);
        }
    }
    __syncthreads();
    for (int j = (tid >> 6) * 4; j < (tid >> 6) * 4 + 4; ++j) {
        float* row = prob + j * kPoolStrideB;
        float hv[kPoolChunkB / 64], m = -INFINITY;
#pragma unroll
        for (int i = 0; i < kPoolChunkB / 64; ++i) { hv[i] = row[lane + 64 * i]; m = fmaxf(m, hv[i]); }
        m = wave_max_b(m);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < kPoolChunkB / 64; ++i) { const float e = expf(hv[i] - m); row[lane + 64 * i] = e; sum += e; }
        sum = wave_sum_b(sum);
        if (lane == 0 && blockIdx.y == 0) {
            float* st = stats + (((size_t)n * kPoolSplit + blockIdx.z) * 24 + j) * 2;      // [n][range][joint][max, sum]
            st[0] = m;
            st[1] = sum;
        }
    }
    const int wv = tid >> 6;                                            // row tile of this wave: channels cb + 16 wv .. + 15
    const float* b0 = prob + l15 * kPoolStrideB + 4 * lq;
    const float* b1 = prob + (16 + (l15 & 7)) * kPoolStrideB + 4 * lq;  // joints 16..23; lanes 8..15 of the second column tile are zero columns
    const bool j1 = l15 < 8;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < kPoolChunkB; p0 += kPoolPB) {
        const int buf = (p0 / kPoolPB) & 1;
        __syncthreads();                                                // buffer `buf` is staged (and the probabilities are final); the other one is free
        if (p0 + kPoolPB < kPoolChunkB) stage(p0 + kPoolPB, buf ^ 1);
        const u16* fs = &fst[buf][(4 * lq) * kPoolFS + wv * 16 + l15];
#pragma unroll
        for (int g = 0; g < kPoolPB / 16; ++g) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(b0 + p0 + 16 * g);
            f32x4 v = *reinterpret_cast<const f32x4*>(b1 + p0 + 16 * g);
            if (!j1) v = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float f = bf2f(fs[(16 * g + k) * kPoolFS]);       // channel l15 of position p0 + 16 g + 4 lq + k
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f, u[k], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f, v[k], acc1, 0, 0, 0);
            }
        }
    }
    const int ct = blockIdx.y * 6 + wv;
    float* o = part + (((size_t)n * kPoolSplit + blockIdx.z) * (CA + CB) + ct * 16 + 4 * lq) * 24;      // [n][split][192][24]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o[r * 24 + l15] = acc0[r];
        if (l15 < 8) o[r * 24 + 16 + l15] = acc1[r];
    }
}

// workgroup barrier for LDS data only: s_waitcnt lgkmcnt(0) + s_barrier.  (__syncthreads() also waits for vmcnt(0): with the range's features requested up front it held
// every barrier of attn_pool_bf16x_kernel until all 86 KB had arrived)
__device__ __forceinline__ void pool_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// NW: waves = 16-channel tiles of the workgroup.  6: two workgroups per (frame, range), 96 channels each, both building the range's softmax; 12 (round 5): ONE workgroup
// for all 192 channels -- the softmax (heat staging, exp, hi / lo split: most of the kernel's vector instructions) is built once, by twice the waves.
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn_pool_bf16x_kernel(const u16* __restrict__ heat, int hc, const u16* __restrict__ featA, int CA, int ctA,
                                                               const u16* __restrict__ featB, int CB, int ctB, float* __restrict__ stats,
                                                               float* __restrict__ part, int P) {
    constexpr int NT = NW * 64, NU = NW * 2, FS = NW * 16 + 8;        // threads; 16-byte units per staged position; bf16 per staged position (+ 16 bytes: the four positions of a k-step on different banks)
    extern __shared__ __align__(16) unsigned char pool_lds[];
    float* prob = reinterpret_cast<float*>(pool_lds);                  // [24][kPoolStrideB]
    u16 (*fst)[kPoolPB * FS] = reinterpret_cast<u16 (*)[kPoolPB * FS]>(pool_lds + 24 * kPoolStrideB * 4);      // two buffers of 64 positions x NW * 16 channels, as they lie in memory (NHWC)
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int pbeg = blockIdx.z * kPoolChunkB, cb = blockIdx.y * NW * 16;      // first channel of this workgroup in [featA | featB]
    // staging: 64 positions x 12 units of 16 bytes = two units per thread (positions spp and spp + 32); a unit lies entirely in featA or in featB
    // (128 = 8 x 16).  Round 5: 64 instead of 32 positions per barrier -- the loop is a chain of load -> LDS -> barrier -> 16 MFMAs round trips
    const int spp = tid / NU, sq = tid - spp * NU, sc = cb + sq * 8;
    const u16* ssrc = sc < CA ? featA + ((size_t)n * P + pbeg + spp) * ctA + sc : featB + ((size_t)n * P + pbeg + spp) * ctB + (sc - CA);
    const size_t sstride = sc < CA ? ctA : ctB;
    // the range's 448 positions x 96 channels (86 KB) are requested AT ONCE, 14 16-byte units per thread, and wait in registers under the softmax below; the loop
    // only moves them to LDS one 64-position block ahead (with the bf16 matrix cores a block is 8 MFMAs per wave: a block-ahead request would wait for HBM every time)
    constexpr int NBLK = kPoolChunkB / kPoolPB;
    // the heat rows are requested FIRST (round 5): vector-memory loads return in order, so behind the 14 feature units the softmax below could not start before the
    // whole 86 KB had arrived -- now it runs under that transfer
    constexpr int NH = (kPoolChunkB * 3 + NT - 1) / NT;
    u32x4 hq[NH];
    u16 hx[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int u = i * NT + tid, p = u / 3, jg = u - p * 3;
        if (u < kPoolChunkB * 3) {
            hq[i] = *reinterpret_cast<const u32x4*>(heat + ((size_t)n * P + pbeg + p) * hc + jg * 8);      // channels 8 jg .. 8 jg + 7 (channel 0 = background)
            hx[i] = heat[((size_t)n * P + pbeg + p) * hc + jg * 8 + 8];                                      // channel 8 jg + 8 = joint 8 jg + 7
        }
    }
    u32x4 fv[2 * NBLK];
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        fv[2 * b] = *reinterpret_cast<const u32x4*>(ssrc + (size_t)(b * kPoolPB) * sstride);
        fv[2 * b + 1] = *reinterpret_cast<const u32x4*>(ssrc + (size_t)(b * kPoolPB + 32) * sstride);
    }
    auto deposit = [&](int b, int buf) {
        *reinterpret_cast<u32x4*>(&fst[buf][spp * FS + sq * 8]) = fv[2 * b];
        *reinterpret_cast<u32x4*>(&fst[buf][(spp + 32) * FS + sq * 8]) = fv[2 * b + 1];
    };
    // heat rows of the range -> LDS: thread = (position, 8 joints)
#pragma unroll
    for (int i = 0; i < NH; ++i) {
        const int u = i * NT + tid, p = u / 3, jg = u - p * 3;
        if (u >= kPoolChunkB * 3) break;
        const u32x4 h8 = hq[i];
        const u16 nx = hx[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int ch = k + 1;                                                                                   // joint 8 jg + k is channel 8 jg + k + 1
            const u16 v = ch < 8 ? (u16)(h8[ch >> 1] >> (16 * (ch & 1))) : nx;
            prob[(jg * 8 + k) * kPoolStrideB + p] = bf2f(v);
        }
    }
    pool_lds_sync();
    for (int j = (tid >> 6) * (24 / NW); j < (tid >> 6) * (24 / NW) + 24 / NW; ++j) {
        float* row = prob + j * kPoolStrideB;
        float hv[kPoolChunkB / 64], m = -INFINITY;
#pragma unroll
        for (int i = 0; i < kPoolChunkB / 64; ++i) { hv[i] = row[lane + 64 * i]; m = fmaxf(m, hv[i]); }
        m = wave_max_b(m);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < kPoolChunkB / 64; ++i) { hv[i] = __builtin_amdgcn_exp2f((hv[i] - m) * 1.442695040888963f); sum += hv[i]; }      // v_exp_f32 (1 ulp; arguments <= 0): libm's expf was a third of this workgroup's instructions
        sum = wave_sum_b(sum);
        // the row in place as TWO bf16 rows: e = hi + lo (hi = bf16(e), lo = bf16(e - hi): 16 bits of mantissa, 2^-17 relative) -- [hi 448][lo 448] in the 452 floats
        // of the fp32 row; every lane holds its 7 values, and the wave (one row at a time) has read the whole row before it writes
        {
            u16* hrow = reinterpret_cast<u16*>(row);
#pragma unroll
            for (int i = 0; i < kPoolChunkB / 64; ++i) {
                const unsigned hb = pack2(hv[i], 0.f);                                        // v_cvt_pk_bf16_f32: the same round-to-nearest-even as the integer form, one instruction
                const float lf = hv[i] - __uint_as_float(hb << 16);
                hrow[lane + 64 * i] = (u16)hb;
                hrow[kPoolChunkB + lane + 64 * i] = (u16)pack2(lf, 0.f);
            }
        }
        if (lane == 0 && blockIdx.y == 0) {
            float* st = stats + (((size_t)n * kPoolSplit + blockIdx.z) * 24 + j) * 2;      // [n][range][joint][max, sum]
            st[0] = m;
            st[1] = sum;
        }
    }
    const int wv = tid >> 6;                                            // row tile of this wave: channels cb + 16 wv .. + 15
    // bf16 matrix cores: out[c][j] += sum over 32 positions of feat[p][c] * (hi + lo)[j][p] -- products of bf16 pairs are exact in fp32, the sums fp32 as before.
    // A (channels x positions) comes TRANSPOSED out of the staged NHWC rows: ds_read_b64_tr_b16 hands lane i of a 16-lane group channel i of four positions
    // (lane 4q + p supplies row q, channels 4p .. 4p + 3), two of them make the lane's 8 positions of the k-step; B (joints x positions) is 16 contiguous bytes
    // of the joint's hi / lo row.  4 MFMAs (16 cycles) per 32 positions instead of 16 fp32 ones (32 cycles).
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x8p __attribute__((ext_vector_type(8)));
    const unsigned char* pb0 = reinterpret_cast<const unsigned char*>(prob) + (size_t)l15 * kPoolStrideB * 4 + lq * 16;                // joints 0 .. 15
    const unsigned char* pb1 = reinterpret_cast<const unsigned char*>(prob) + (size_t)(16 + (l15 & 7)) * kPoolStrideB * 4 + lq * 16;    // joints 16 .. 23 (lanes 8 .. 15: copies, columns never stored)
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    deposit(0, 0);
#pragma unroll
    for (int b = 0; b < NBLK; ++b) {
        const int buf = b & 1, p0 = b * kPoolPB;
        pool_lds_sync();                                                // buffer `buf` is staged (and the probabilities are final); the other one is free
        if (b + 1 < NBLK) deposit(b + 1, buf ^ 1);
        const u16* fs = &fst[buf][(8 * lq + (l15 >> 2)) * FS + wv * 16 + 4 * (l15 & 3)];
#pragma unroll
        for (int g = 0; g < kPoolPB / 32; ++g) {
            const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(fs + (32 * g) * FS));
            const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(fs + (32 * g + 4) * FS));
            const bf16x8p af = __builtin_bit_cast(bf16x8p, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
            const int pe = (p0 + 32 * g) * 2;
            const bf16x8p h0 = *reinterpret_cast<const bf16x8p*>(pb0 + pe), l0 = *reinterpret_cast<const bf16x8p*>(pb0 + kPoolChunkB * 2 + pe);
            const bf16x8p h1 = *reinterpret_cast<const bf16x8p*>(pb1 + pe), l1 = *reinterpret_cast<const bf16x8p*>(pb1 + kPoolChunkB * 2 + pe);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, h0, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, l0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, h1, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, l1, acc1, 0, 0, 0);
        }
    }
    const int ct = blockIdx.y * NW + wv;
    float* o = part + (((size_t)n * kPoolSplit + blockIdx.z) * (CA + CB) + ct * 16 + 4 * lq) * 24;      // [n][split][192][24]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o[r * 24 + l15] = acc0[r];
        if (l15 < 8) o[r * 24 + 16 + l15] = acc1[r];
    }
}

size_t lds_bytes_bf16(const ConvArgs& a, int tc, int nbuf, int tps_px) {
    const size_t aunits = ((size_t)a.PSTR * kSlotU + 63) & ~(size_t)63;
    // weights are read from global memory: LDS holds the input patch and, afterwards, the output tile [pixels][tc + 8] bf16
    const size_t patch = (size_t)nbuf * aunits * 16, tile = (size_t)tps_px * (tc + 8) * 2;
    return (patch > tile ? patch : tile) + (size_t)a.PSTR * 4;
}

bool plan_bf16(ConvArgs& a, int tps, int tc) {
    const int TP = tps * 16, HoWo = a.Ho * a.Wo;
    if (a.Wo > TP) return false;
    if (HoWo <= TP) { a.G = TP / HoWo; if (a.G > a.N) a.G = a.N; a.R = a.Ho; }
    else { a.G = 1; a.R = TP / a.Wo; if (a.R > a.Ho) a.R = a.Ho; }
    a.tiles_y = (a.Ho + a.R - 1) / a.R;
    a.groups = (a.N + a.G - 1) / a.G;
    a.Rin = (a.R - 1) * a.stride + a.ks;
    a.Wp = (a.Wo - 1) * a.stride + a.ks;
    a.PSTR = a.G * a.Rin * a.Wp;                           // patch slots
    a.gx = a.tiles_y * a.groups;
    a.gy = a.CoutPad / tc;
    if ((long)a.PSTR * kSlotU >= 65536) return false;
    a.gx8 = (a.gx + 7) / 8;
    a.xcd = (a.gy > 1 || a.ks > 1) && a.gx >= 16;
    // chunks double-buffered when two workgroups still fit a CU that way (or nothing else fits), else single-buffered
    // (thresholds of 32..80 KB for double buffering measure within 1.5 % of each other at 256 frames)
    a.nbuf = lds_bytes_bf16(a, tc, 2, TP) <= 80 * 1024 ? 2 : (lds_bytes_bf16(a, tc, 1, TP) <= 80 * 1024 ? 1 : (lds_bytes_bf16(a, tc, 2, TP) <= 160 * 1024 ? 2 : 1));
    // (measured at 256 frames: double buffers at one workgroup per CU are 1.7x slower than single buffers at two)
    if (a.CinPad == kCK) a.nbuf = 1;                          // a single chunk has nothing to double-buffer: half the LDS, twice the workgroups per CU
    return lds_bytes_bf16(a, tc, a.nbuf, TP) <= 160 * 1024;
}

template <int KS, int S>
hipError_t dispatch_bf16(const ConvArgs& a, int tps, int tc, hipStream_t s) {
    const size_t lds = lds_bytes_bf16(a, tc, a.nbuf, tps * 16);
    const dim3 grid((a.xcd ? a.gx8 * 8 : a.gx) * a.gy);
    // Register-held DMA offsets (REG) for the 64-channel tiles of the 56-wide 3x3 layers with two or more chunks: their chunk loop is
    // where the per-chunk staging code stood in front of the MFMAs (256 -> 256 @56x56 at 256 frames: 1 127 -> 1 045 us, 480 -> 256:
    // 1 924 -> 1 770).  The 16-channel-per-wave variants lose more to the registers (152 instead of 116: three waves per SIMD instead
    // of four) than they gain: 32 -> 32 @56x56 40.7 -> 48.5 us, 64 -> 64 @28x28 32.3 -> 36.6, 64 -> 256 1x1 272 -> 300.
    static const int reg_env = GRNET_AB(BF16_REG, 1);
    const int aunits = (a.PSTR * kSlotU + 63) & ~63;
    if (reg_env && KS == 3 && tps == 14 && tc == 64 && a.CinPad >= 2 * kCK && aunits <= kRegUnits * 256 &&
        (size_t)a.N * a.H * a.W * a.in_ctot < 0xfffffff0u)                                  // the offsets are 32-bit element counts
        return launch_k(conv_bf16_nhwc<KS, S, 14, 4, 2, 2, true>, grid, dim3(256), lds, s, a);
    if constexpr (KS == 1) {
        // 1x1 layers with 256 output channels (layer1's expansions): ONE workgroup writes all 256 channels of its 112 pixels = 56 KB of consecutive
        // bytes.  With 64-channel tiles four workgroups each write 128 bytes of every 512-byte pixel row, at different times (round 5)
        if (tps == 7 && tc == 256) return launch_k(conv_bf16_nhwc<1, 1, 7, 16, 1, 4>, grid, dim3(256), lds, s, a);
    }
    if (tps == 14 && tc == 64) return launch_k(conv_bf16_nhwc<KS, S, 14, 4, 2, 2>, grid, dim3(256), lds, s, a);
    static const int occ_env = GRNET_AB(BF16_OCC, 3);      // workgroups per CU from which the four-wave variants run
    if ((long)a.gx * a.gy >= (long)occ_env * 256) {
        if (tps == 14 && tc == 32) return launch_k(conv_bf16_nhwc<KS, S, 14, 2, 2, 2, false, true>, grid, dim3(256), lds, s, a);
        if (tps == 7 && tc == 64) return launch_k(conv_bf16_nhwc<KS, S, 7, 4, 1, 4, false, true>, grid, dim3(256), lds, s, a);
        if (tps == 7 && tc == 32) return launch_k(conv_bf16_nhwc<KS, S, 7, 2, 1, 2, false, true>, grid, dim3(128), lds, s, a);
    }
    if (tps == 14 && tc == 32) return launch_k(conv_bf16_nhwc<KS, S, 14, 2, 2, 2>, grid, dim3(256), lds, s, a);
    if (tps == 7 && tc == 64) return launch_k(conv_bf16_nhwc<KS, S, 7, 4, 1, 4>, grid, dim3(256), lds, s, a);
    if (tps == 7 && tc == 32) return launch_k(conv_bf16_nhwc<KS, S, 7, 2, 1, 2>, grid, dim3(128), lds, s, a);
    return hipErrorInvalidValue;
}

template <typename K>
hipError_t set_lds_bf16(K kern) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <int KS, int S>
hipError_t init_bf16_ks() {
    GRK_TRY(set_lds_bf16(conv_bf16_nhwc<KS, S, 14, 4, 2, 2>));
    GRK_TRY(set_lds_bf16(conv_bf16_nhwc<KS, S, 14, 2, 2, 2>));
    GRK_TRY(set_lds_bf16(conv_bf16_nhwc<KS, S, 7, 4, 1, 4>));
    GRK_TRY(set_lds_bf16(conv_bf16_nhwc<KS, S, 7, 2, 1, 2>));
    GRK_TRY(set_lds_bf16(conv_bf16_nhwc<KS, S, 14, 4, 2, 2, true>));
    GRK_TRY(set_lds_bf16(conv_bf16_nhwc<KS, S, 14, 2, 2, 2, false, true>));
    GRK_TRY(set_lds_bf16(conv_bf16_nhwc<KS, S, 7, 4, 1, 4, false, true>));
    GRK_TRY(set_lds_bf16(conv_bf16_nhwc<KS, S, 7, 2, 1, 2, false, true>));
    return hipSuccess;
}

inline int blocks_for(long total) { return (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384); }

// ---- layer1's 64 -> 256 1x1 convolutions as a STREAM (round 5).  These launches are HBM traffic and nothing else (256 frames: read 103 + 411 MB, write
// 411 MB; 32 KFLOP per 1.2 KB), and conv_bf16_nhwc moves it at 3.6-3.9 TB/s: a workgroup's life there is request -> wait -> MFMA -> epilogue -> store, and with
// two workgroups per CU the memory system idles while both compute.  The same bytes as a plain copy-shaped kernel run at 5.7-6.0 TB/s
// (tools/micro/stream_mix.hip: 153-162 us against 239-254).  Here the loads never stop:
//   * persistent workgroups (4 waves) walk 32-pixel tiles; tile t + 1's input and residual are requested into REGISTERS (16 bytes per lane, whole rows:
//     5 loads per lane) before tile t is touched, and are written to LDS a whole tile of work later -- ordinary loads, no LDS-DMA, so hipcc's counted
//     vmcnt waits apply and the stores of tile t do not wait for them;
//   * barriers are s_waitcnt lgkmcnt(0) + s_barrier (a __syncthreads() would drain the prefetch);
//   * wave w holds the weights of output channels 64 w .. 64 w + 63 (and, for a pair, 16 w .. + 15 of the reduction) in registers for the whole launch;
//   * the residual goes global -> registers -> LDS tile ([pixel][256] bf16, 544-byte slots) and is added in the accumulators' layout exactly where and as
//     conv_bf16_nhwc adds it ((acc + residual) + bias, ReLU, round to bf16: bit-identical); the result replaces it IN PLACE, leaves as whole 512-byte
//     rows (16 bytes per lane), and is the B operand of the pair's 256 -> 64 reduction (second stage of the generic kernel: same operands, same order).
// KC: 32-channel chunks of the input (2: one tensor; 4: [t ; x], the Bottleneck's last 1x1 and its downsample as one GEMM); RES: one same-size addend.
// CT: output channels (= CoutPad) -- 256: 64 per wave (layer1's expansions); 64 / 32: one 16-channel block per wave (waves beyond CT / 16 only move rows): layer1's first
// 64 -> 64 and the PARE head's 128 -> 64 / 128 -> 25 (round 5).  TWO: input chunks 2, 3 come from a.in2.
template <int KC, int CT, bool TWO, bool RES, bool PAIR>
__global__ __launch_bounds__(256) void conv_bf16_pw_stream(const ConvArgs a, int ntiles) {
    constexpr int TP = 32, ISB = KC * 64 + 32, OSB = CT * 2 + 32, IU = KC * 4, NIN = TP * IU / 256, NBW = CT >= 64 ? CT / 64 : 1, OU = CT / 8, NOUT = (TP * OU + 255) / 256;      // pixels per tile; LDS slot strides (32 x odd bytes); 16-byte input units per pixel / per thread
    __shared__ __align__(16) unsigned char inl[TP * ISB];
    __shared__ __align__(16) unsigned char tile[TP * OSB];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u16* w = reinterpret_cast<const u16*>(a.w);
    static_assert(!PAIR || CT == 256, "the pair's reduction reads 256 channels");
    const int co = (wave * 16 * NBW) % CT;                     // first output channel of this wave (a wave without a block of its own recomputes another's and stores nothing)
    const bool own = wave * 16 * NBW < CT;
    bf16x8 af[KC][NBW];
    f32x4 bv[NBW];
#pragma unroll
    for (int cb = 0; cb < NBW; ++cb) {
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) af[kc][cb] = *reinterpret_cast<const bf16x8*>(w + ((size_t)kc * CT + co + cb * 16 + l15) * 32 + lq * 8);
        bv[cb] = *reinterpret_cast<const f32x4*>(a.bias + co + cb * 16 + lq * 4);
    }
    bf16x8 af2[8];
    f32x4 b2 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (PAIR) {
        const u16* w2 = reinterpret_cast<const u16*>(a.w2);
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) af2[ch] = *reinterpret_cast<const bf16x8*>(w2 + ((size_t)ch * 64 + wave * 16 + l15) * 32 + lq * 8);
        b2 = *reinterpret_cast<const f32x4*>(a.bias2 + wave * 16 + lq * 4);
    }
    const u16* in = reinterpret_cast<const u16*>(a.in) + a.in_coff;
    const u16* in2 = TWO ? reinterpret_cast<const u16*>(a.in2) + a.in2_coff : nullptr;
    const u16* res = RES ? reinterpret_cast<const u16*>(a.add[0]) + a.add_coff[0] : nullptr;
    u16* out = reinterpret_cast<u16*>(a.out) + a.out_coff;
    u16* out2 = PAIR ? reinterpret_cast<u16*>(a.out2) + a.out2_coff : nullptr;
    const float lo1 = a.relu ? 0.f : -__builtin_inff(), lo2 = a.relu2 ? 0.f : -__builtin_inff();      // max(v, -inf) = v: no branch per value
    u32x4 nin[NIN], nres[NOUT];
    auto request = [&](int t) {                               // tile t's rows -> registers (clamped: the last iteration re-requests its own tile instead of branching)
        const size_t p0 = (size_t)(t < ntiles ? t : ntiles - 1) * TP;
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int u = i * 256 + tid, px = u / IU, part = u - px * IU;
            if (TWO && part >= 8) nin[i] = *reinterpret_cast<const u32x4*>(in2 + (p0 + px) * a.in2_ctot + (part - 8) * 8);
            else nin[i] = *reinterpret_cast<const u32x4*>(in + (p0 + px) * a.in_ctot + part * 8);
        }
        if (RES) {
#pragma unroll
            for (int i = 0; i < NOUT; ++i) {
                const int u = i * 256 + tid, px = u / OU, part = u - px * OU;
                if (u < TP * OU) nres[i] = *reinterpret_cast<const u32x4*>(res + (p0 + px) * a.add_ctot[0] + part * 8);
            }
        }
    };
    auto lds_sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    request(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const size_t p0 = (size_t)t * TP;
#pragma unroll
        for (int i = 0; i < NIN; ++i) {
            const int u = i * 256 + tid, px = u / IU, part = u - px * IU;
            *reinterpret_cast<u32x4*>(inl + px * ISB + part * 16) = nin[i];
        }
        if (RES) {
#pragma unroll
            for (int i = 0; i < NOUT; ++i) {
                const int u = i * 256 + tid, px = u / OU, part = u - px * OU;
                if (u < TP * OU) *reinterpret_cast<u32x4*>(tile + px * OSB + part * 16) = nres[i];
            }
        }
        request(t + gridDim.x);                               // in flight under everything below
        lds_sync();
        f32x4 acc[NBW][2];
#pragma unroll
        for (int cb = 0; cb < NBW; ++cb)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) acc[cb][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const bf16x8 bt = *reinterpret_cast<const bf16x8*>(inl + (pt * 16 + l15) * ISB + kc * 64 + lq * 16);
#pragma unroll
                for (int cb = 0; cb < NBW; ++cb) acc[cb][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kc][cb], bt, acc[cb][pt], 0, 0, 0);
            }
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int cb = 0; cb < NBW; ++cb) {
                unsigned char* pos = tile + (pt * 16 + l15) * OSB + (co + cb * 16) * 2 + lq * 8;
                f32x4 v = acc[cb][pt];
                if (RES) {
                    const u32x2 r = *reinterpret_cast<const u32x2*>(pos);
                    v[0] += bf2f((u16)(r[0] & 0xffffu)); v[1] += bf2f((u16)(r[0] >> 16));
                    v[2] += bf2f((u16)(r[1] & 0xffffu)); v[3] += bf2f((u16)(r[1] >> 16));
                }
                v = v + bv[cb];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], lo1);
                if (own) *reinterpret_cast<u32x2*>(pos) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};
            }
        lds_sync();                                           // the tile is complete (and nobody reads inl any more)
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int u = i * 256 + tid, px = u / OU, part = u - px * OU;
            if (u < TP * OU) *reinterpret_cast<u32x4*>(out + (p0 + px) * a.out_ctot + part * 8) = *reinterpret_cast<const u32x4*>(tile + px * OSB + part * 16);
        }
        if (PAIR) {
            f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ch = 0; ch < 8; ++ch)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const bf16x8 bt = *reinterpret_cast<const bf16x8*>(tile + (pt * 16 + l15) * OSB + ch * 64 + lq * 16);
                    acc2[pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af2[ch], bt, acc2[pt], 0, 0, 0);
                }
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                f32x4 v = acc2[pt] + b2;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], lo2);
                *reinterpret_cast<u32x2*>(inl + (pt * 16 + l15) * ISB + wave * 32 + lq * 8) = u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])};      // inl: staging for whole 128-byte rows
            }
            lds_sync();
            const int px = tid >> 3, part = tid & 7;
            *reinterpret_cast<u32x4*>(out2 + (p0 + px) * a.out2_ctot + part * 8) = *reinterpret_cast<const u32x4*>(inl + px * ISB + part * 16);
        }
        lds_sync();                                           // everybody is done with inl and the tile: the next iteration writes both
    }
}

bool pw_stream_eligible(const ConvArgs& a) {
    if (!(a.ks == 1 && a.stride == 1 && a.Cout <= a.CoutPad && a.relu_from == 0 && a.H == a.Ho && a.W == a.Wo)) return false;
    static const int small_env = GRNET_AB(BF16_PW_STREAM_SMALL, 3);     // bit 0: 64 / 32 output channels on 56x56 maps, bit 1: on smaller maps
    if (!(a.CoutPad == 256 || ((a.CoutPad == 64 || a.CoutPad == 32) && (small_env & (a.W >= 56 ? 1 : 2))))) return false;
    if (a.in2 ? !(a.Cin == 128 && a.CinPad == 128 && a.cin_split == 64 && a.in2_ctot % 8 == 0 && a.in2_coff % 8 == 0 && a.in2_ctot - a.in2_coff >= 64 && a.in_ctot - a.in_coff >= 64)
              : !((a.Cin == 64 && a.CinPad == 64) || (a.Cin == 128 && a.CinPad == 128)) || a.in_ctot - a.in_coff < a.CinPad) return false;
    if (a.n_add > 1 || (a.n_add == 1 && (a.add_shift[0] != 0 || a.add_ctot[0] % 8 != 0 || a.add_coff[0] % 8 != 0 || a.add_ctot[0] - a.add_coff[0] < a.CoutPad))) return false;
    if (a.out_ctot - a.out_coff < a.CoutPad) return false;                                      // whole rows of CoutPad channels are written (padding channels: zeros, as the generic kernel's)
    if (a.w2 && (a.CoutPad != 256 || a.out2_ctot % 8 != 0 || a.out2_coff % 8 != 0 || a.out2_ctot - a.out2_coff < 64)) return false;
    // instantiated: 64 -> 256 {residual, none} x {pair, none}; [t ; x] -> 256 + pair; 64 / 128 -> 64 / 32 without addend
    if (a.CoutPad == 256 ? (a.in2 ? !(a.w2 && a.n_add == 0) : a.Cin != 64) : (a.in2 || a.n_add != 0 || a.w2)) return false;
    const long px = (long)a.N * a.H * a.W;
    return px % 32 == 0 && (a.pw_stream == 2 || px / 32 >= 512L * 8);      // persistent workgroups: at least eight tiles each, or their fixed cost (the weights) does not pay
}
hipError_t launch_pw_stream(const ConvArgs& a, hipStream_t s) {
    const int ntiles = (int)((long)a.N * a.H * a.W / 32);
    static const int wgs_env = GRNET_AB(BF16_PW_STREAM_WGS, 2);      // workgroups per CU
    int cus = 0;
    GRK_TRY(device_cu_count(&cus));
    const dim3 grid(std::min(ntiles, cus * wgs_env));
#define GRK_PWS(KC, CT, TWO, RES, PAIR) launch_k(conv_bf16_pw_stream<KC, CT, TWO, RES, PAIR>, grid, dim3(256), 0, s, a, ntiles)
    if (a.CoutPad == 256) {
        if (a.in2) return GRK_PWS(4, 256, true, false, true);
        if (a.n_add == 1) return a.w2 ? GRK_PWS(2, 256, false, true, true) : GRK_PWS(2, 256, false, true, false);
        return a.w2 ? GRK_PWS(2, 256, false, false, true) : GRK_PWS(2, 256, false, false, false);
    }
    if (a.CoutPad == 64) return a.Cin == 64 ? GRK_PWS(2, 64, false, false, false) : GRK_PWS(4, 64, false, false, false);
    return a.Cin == 64 ? GRK_PWS(2, 32, false, false, false) : GRK_PWS(4, 32, false, false, false);
#undef GRK_PWS
}

}  // namespace

hipError_t conv_bf16_init() {
    GRK_TRY(set_lds_bf16(conv_bf16_nhwc<1, 1, 7, 16, 1, 4>));
    GRK_TRY(set_lds_bf16(attn_pool_bf16x_kernel<6>));
    GRK_TRY(set_lds_bf16(attn_pool_bf16x_kernel<12>));
    GRK_TRY((init_bf16_ks<1, 1>()));
    GRK_TRY((init_bf16_ks<3, 1>()));
    GRK_TRY((init_bf16_ks<3, 2>()));
    return hipSuccess;
}

// a.in / a.out / a.w / a.add[] point at bf16 data (NHWC activations, [CinPad/32][tap][CoutPad][32] weights); a.bias is fp32;
// a.zeros: >= 16 zero bytes in HBM.
// Requirements: CinPad % 32 == 0, CoutPad % 32 == 0, in_ctot / in_coff / out_ctot / out_coff multiples of 8, add_ctot / add_coff multiples of 4.
hipError_t launch_conv_bf16(ConvArgs a, hipStream_t s, int tile_hint) {
    if (!((a.ks == 1 && a.stride == 1) || (a.ks == 3 && (a.stride == 1 || a.stride == 2)))) return hipErrorInvalidValue;
    if (a.CinPad % kCK != 0 || a.CoutPad % 32 != 0 || a.in_ctot % 8 != 0 || a.in_coff % 8 != 0 || a.out_ctot % 8 != 0 || a.out_coff % 8 != 0)
        return hipErrorInvalidValue;
    if (a.pw_stream && tile_hint == 0 && pw_stream_eligible(a)) return launch_pw_stream(a, s);      // a.pw_stream 0: these layers on conv_bf16_nhwc (GRNET_OPT_BF16_CHAIN bit 7)
    static const int direct_env = GRNET_AB(BF16_DIRECT, 1);
    if (direct_env && tile_hint == 0 && bf16_direct_eligible(a)) {
        // rows per strip: enough strips to give every SIMD about two waves, at most 8 rows (weights and the first rows are a strip's fixed cost)
        const int halves = a.Cin / 32, cols = a.W / 28;       // a strip is R rows x 28 pixels (2 tiles) x 32 output channels
        int R = (int)((long)a.N * a.H * halves * cols / (a.Cin == 64 ? 2048 : 3072));      // 64 channels: 36 KB of weights per strip, fewer and longer strips
        R = R < 2 ? 2 : (R > 8 ? 8 : R);
        const int spf = (a.H + R - 1) / R, waves = a.N * spf * cols * halves;
        const dim3 grid((waves + 3) / 4);
        if (a.Cin == 32) return launch_k(conv_bf16_direct<32, 2>, grid, dim3(256), 0, s, a, R, spf);
        return launch_k(conv_bf16_direct<64, 2>, grid, dim3(256), 0, s, a, R, spf);
    }
    int tc = a.CoutPad % 64 == 0 ? 64 : 32;                // measured at 256 frames: 32 everywhere is 1.5x slower
    // 224-pixel tiles for the 56-wide maps, 112 otherwise (measured: 224 everywhere ties at 256 frames and loses 13 % at 16; 112 everywhere loses 28 %)
    int tps = (tile_hint == 7 || tile_hint == 14) ? tile_hint : (a.Wo >= 56 ? 14 : 7);
    static const int pw256_env = GRNET_AB(BF16_PW256, 0);      // measured: 263 vs 237 us for 64 -> 256 with its residual at 256 frames -- off
    if ((pw256_env || a.w2) && tile_hint == 0 && a.ks == 1 && a.CoutPad == 256 && a.Wo == 56 && (long)a.N * a.Ho * a.Wo >= 256L * 112 * 2) { tc = 256; tps = 7; }
    if (a.w2 && tc != 256) return hipErrorInvalidValue;    // a pair needs the 256-channel tile
    if (!plan_bf16(a, tps, tc)) {
        tps = tps == 14 ? 7 : 14;
        if (!plan_bf16(a, tps, tc)) return hipErrorInvalidValue;
    }
#ifdef GRNET_ABLATION
    if (GRNET_AB_SET(BF16_PHASES)) a.dbg |= 8;
#endif
    hipError_t e = a.ks == 1 ? dispatch_bf16<1, 1>(a, tps, tc, s) : (a.stride == 1 ? dispatch_bf16<3, 1>(a, tps, tc, s) : dispatch_bf16<3, 2>(a, tps, tc, s));
#ifdef GRNET_ABLATION
    static const bool phases = GRNET_AB_SET(BF16_PHASES);
    if (phases && e == hipSuccess) {
        unsigned long long h[8] = {}, z[8] = {};
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h));
        hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z));
        const double n = h[4] ? (double)h[4] : 1.0;
        fprintf(stderr, "[bf16 phases] %d->%d k%d s%d %dx%d N%d tps %d tc %d nbuf %d wgs %llu: per WG ticks  table %.0f  first-wait %.0f  loop %.0f (%d chunks)  epilogue %.0f\n",
                a.Cin, a.Cout, a.ks, a.stride, a.H, a.W, a.N, tps, tc, a.nbuf, h[4], h[0] / n, h[1] / n, h[2] / n, a.CinPad / kCK, h[3] / n);
    }
#endif
    return e;
}

// weights (64, 3, 3, 3) folded, fp64 -> [4 channel blocks][64 lanes][8] bf16: lane (l15, lq) of block mt holds W[16 mt + l15][k = 8 lq .. + 7], k = c * 9 + tap
void pack_stem_weights_bf16(const double* w, unsigned short* out) {
    for (int mt = 0; mt < 4; ++mt)
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * (l >> 4) + j, co = 16 * mt + (l & 15);
                float f = k < 27 ? (float)w[(size_t)co * 27 + k] : 0.f;
                unsigned u;
                memcpy(&u, &f, 4);
                u += 0x7fffu + ((u >> 16) & 1u);
                out[((size_t)mt * 64 + l) * 8 + j] = (unsigned short)(u >> 16);
            }
}

hipError_t launch_conv_bf16_stem(const float* frames, const void* wpk, const float* bias, void* out, int out_ctot, int out_coff, int N, int relu, hipStream_t s) {
    if (N < 1 || out_ctot % 8 != 0 || out_coff % 8 != 0) return hipErrorInvalidValue;
    return launch_k(conv_bf16_stem, dim3((N * 112 + 3) / 4), dim3(256), 0, s, frames, static_cast<const u16*>(wpk), bias, static_cast<u16*>(out), out_ctot, out_coff, N, relu);
}

hipError_t launch_nchw_f32_to_nhwc_bf16(const float* in, void* out, int N, int C, int H, int W, int Cp, hipStream_t s) {
    if (Cp % 8 != 0) return hipErrorInvalidValue;
    const long total = (long)N * H * W * Cp;
    return launch_k(nchw_f32_to_nhwc_bf16_kernel, dim3(blocks_for(total / 8)), dim3(256), 0, s, in, reinterpret_cast<u16*>(out), C, H * W, Cp, total);
}
hipError_t launch_nhwc_bf16_to_nchw_f32(const void* in, float* out, int N, int C, int H, int W, int ctot, int coff, hipStream_t s) {
    const long total = (long)N * C * H * W;
    return launch_k(nhwc_bf16_to_nchw_f32_kernel, dim3(blocks_for(total)), dim3(256), 0, s, reinterpret_cast<const u16*>(in), out, C, H * W, ctot, coff, total);
}
hipError_t launch_bilinear2x_bf16(const void* in, void* out, int N, int C, int H, int W, hipStream_t s) {
    if (C % 8 != 0) return hipErrorInvalidValue;
    static const int rows_env = GRNET_AB(BF16_BILINEAR_ROWS, 1);     // 0: four taps per output unit through the vector L1
    const size_t lds = (size_t)2 * W * C * 2;
    if (rows_env && lds <= 48 * 1024)
        return launch_k(bilinear2x_bf16_rows_kernel, dim3(N * H), dim3(256), lds, s, reinterpret_cast<const u16*>(in), reinterpret_cast<u16*>(out), N, C, H, W);
    return launch_k(bilinear2x_bf16_kernel, dim3(N * 2 * H), dim3(256), 0, s, reinterpret_cast<const u16*>(in), reinterpret_cast<u16*>(out), N, C, H, W);
}
hipError_t launch_fuse_sum_bf16(const SumArgs& a, hipStream_t s) {
    if (a.C % 8 != 0 || a.n_add < 1 || a.n_add > 4) return hipErrorInvalidValue;
    return launch_k(fuse_sum_bf16_kernel, dim3(a.N * a.H / (a.H % 4 == 0 ? 4 : 1)), dim3(256), 0, s, a);
}
// heat (N,P,hc) with channel 0 = background; featA / featB: first channel of the view, ctA / ctB channels per pixel in memory;
// pool_ws as launch_softmax_pool fills it.
hipError_t launch_softmax_pool_bf16(const void* heat, int hc, const void* featA, int CA, int ctA, const void* featB, int CB, int ctB, float* pool_ws, int N,
                                    int P, hipStream_t s) {
    if (CA != 128 || CB != 64 || P != kPoolChunkB * kPoolSplit || hc < 32 || hc % 8 != 0 || ctA % 8 != 0 || ctB % 8 != 0) return hipErrorInvalidValue;
    float* part = pool_ws + (size_t)N * kPoolStatsFloats;
    const int x16 = GRNET_AB(BF16_POOL_X16, 1);     // 0: the fp32-MFMA form (A/B; read per launch)
    if (x16)
        {
            static const int nw_env = GRNET_AB(BF16_POOL_WAVES, 12);      // 12 (default): one workgroup for all 192 channels; 6: two of 96
            if (nw_env == 12) return launch_k(attn_pool_bf16x_kernel<12>, dim3(N, 1, kPoolSplit), dim3(768), (size_t)24 * kPoolStrideB * 4 + 2 * kPoolPB * (12 * 16 + 8) * 2, s, reinterpret_cast<const u16*>(heat), hc, reinterpret_cast<const u16*>(featA),
                        CA, ctA, reinterpret_cast<const u16*>(featB), CB, ctB, pool_ws, part, P);
            return launch_k(attn_pool_bf16x_kernel<6>, dim3(N, (CA + CB) / 96, kPoolSplit), dim3(384), (size_t)24 * kPoolStrideB * 4 + 2 * kPoolPB * (6 * 16 + 8) * 2, s, reinterpret_cast<const u16*>(heat), hc, reinterpret_cast<const u16*>(featA),
                        CA, ctA, reinterpret_cast<const u16*>(featB), CB, ctB, pool_ws, part, P);
        }
    return launch_k(attn_pool_bf16_kernel, dim3(N, (CA + CB) / 96, kPoolSplit), dim3(384), 0, s, reinterpret_cast<const u16*>(heat), hc, reinterpret_cast<const u16*>(featA),
                    CA, ctA, reinterpret_cast<const u16*>(featB), CB, ctB, pool_ws, part, P);
}

}  // namespace grk
