// Fused convolution for the HRNet-W32 / PARE conv stack (reference: every nn.Conv2d +
// BatchNorm2d(eval) [+ residual] [+ ReLU] group of lib/models/hrnet.py:43-59,80-100,199-241,
// 357-385,444-451,470-475 and lib/models/pare.py:197-210,388-397), written for gfx950.
//
// Implicit GEMM on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 fma chains):
//   M = output pixels (16 per MFMA, lane&15), N = output channels (16 per MFMA),
//   K = (tap, input channel); the 4 k-values of one MFMA are 4 consecutive input channels
//   of one filter tap.  Each operand is ONE f32 per lane, so NCHW needs no transposition:
//   16 lanes read 16 neighbouring pixels of one channel plane from LDS.
// Per workgroup: a tile of TPS*16 pixels (R output rows of one image, or G whole small
// images) x TCS*16 output channels.  Per K-chunk of 8 input channels the input patch
// (with halo, zero padded) and the weight slab are written to LDS by LDS-DMA
// (global_load_lds, no VGPR round trip), double-buffered: chunk c+1 is in flight while the
// MFMAs of chunk c run; one barrier per chunk.
// Epilogue (registers -> HBM, 16 B per lane where alignment allows): + folded-BN bias,
// + up to 3 addends each optionally nearest-upsampled by 2^shift (the HR fuse layers),
// ReLU, store into a channel slice of the destination buffer.
#include "kernels.h"

#include <cstdlib>

namespace grk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Phase-ablation switches of tools/conv_micro.py (skip MFMAs / re-staging / epilogue to see what a launch spends where).
// They exist only in a diagnostic build (make ABLATION=1); in the product build the conditions fold to false.
#ifdef GRNET_ABLATION
#define GRK_DBG(a, bit) (((a).dbg & (bit)) != 0)
#else
#define GRK_DBG(a, bit) false
#endif
// Diagnostic build only: shader-clock ticks of the phases of a split-K workgroup, summed over workgroups (wave 0 reports):
// [0] index math up to the first stage's DMA, [1] wait for it, [2] stage loop, [3] cross-wave reduction, [4] epilogue, [5] workgroups.
#ifdef GRNET_ABLATION
__device__ unsigned long long g_phase_f32[8];
__device__ unsigned long long g_phase_wk[8];    // whole-K: [0] index math up to the first chunk's DMA, [1] more math + wait for it, [2] chunk loop, [3] epilogue, [4] workgroups
#define GRK_PHASE_WK(i, t0, t1) do { if (threadIdx.x == 0) atomicAdd(&g_phase_wk[i], (t1) - (t0)); } while (0)
#define GRK_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#define GRK_PHASE(i, t0, t1) do { if (threadIdx.x == 0) atomicAdd(&g_phase_f32[i], (t1) - (t0)); } while (0)
#else
#define GRK_TICK(var) do { } while (0)
#define GRK_PHASE(i, t0, t1) do { } while (0)
#endif
#define GRNET_GLOBAL_AS __attribute__((address_space(1)))
#define GRNET_LDS_AS __attribute__((address_space(3)))

#ifndef GRNET_PLAIN_STAGING
// LDS-DMA: lane l's 4 (16) bytes land at lds_wave_base + 4*l (16*l); the source is per lane.
__device__ __forceinline__ void stage4(const float* src, float* lds_wave_base, int) {
    __builtin_amdgcn_global_load_lds((const GRNET_GLOBAL_AS void*)src, (GRNET_LDS_AS void*)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ void stage16(const float* src, float* lds_wave_base, int) {
    __builtin_amdgcn_global_load_lds((const GRNET_GLOBAL_AS void*)src, (GRNET_LDS_AS void*)lds_wave_base, 16, 0, 0);
}
#else
// Debug variant: same LDS image through registers.
__device__ __forceinline__ void stage4(const float* src, float* lds_wave_base, int lane) {
    lds_wave_base[lane] = *src;
}
__device__ __forceinline__ void stage16(const float* src, float* lds_wave_base, int lane) {
    reinterpret_cast<f32x4*>(lds_wave_base)[lane] = *reinterpret_cast<const f32x4*>(src);
}
#endif

// Epilogue shared by both kernel families: + bias, + addends (optionally nearest-upsampled), ReLU,
// 16-byte store when the 4 pixels of the lane are contiguous and aligned in the NCHW plane.
__device__ __forceinline__ int floor4(int v) { return v & ~3; }   // two's complement: floors negatives too

// Block -> (pixel tile bx, output-channel block by) of the 1-D grid.  Workgroups go to the 8 XCDs round-robin in
// dispatch order and every XCD has its own 4 MB L2, so the gy channel blocks that read the SAME input tile are
// made consecutive blocks of ONE XCD (ids congruent mod 8): they are co-resident, stream the tile's channels in
// step and the second..gy-th read of every line hits that L2 instead of the fabric (measured, tools/fetch_calib.py:
// plain (tile, block) order re-fetches the 480-channel head input once per channel block).  The tiles of an XCD
// are a contiguous range, so the halo rows two neighbouring row tiles both read are fetched once as well.  The placement is a
// speed heuristic only -- no correctness depends on it.  Tiles past gx (grid rounded up to 8) exit at once.
__device__ __forceinline__ bool xcd_block(const ConvArgs& a, int& bx, int& by) {
    const int id = blockIdx.x;
    if (!a.xcd) { by = id / a.gx; bx = id - by * a.gx; return true; }
    const int j = id >> 3;
    if (a.xcd == 2) {                  // weights larger than the input (7x7 maps): an XCD owns channel blocks instead, so
        const int gyp = a.gy >> 3;     // every weight slab is fetched by one L2 only (gy is a multiple of 8 here)
        bx = j / gyp;
        by = (j - bx * gyp) * 8 + (id & 7);
        return true;
    }
    const int q = j / a.gy;
    by = j - q * a.gy;
    const int x = id & 7, first = (x * a.gx) >> 3;      // an XCD owns a contiguous, balanced range of tiles: vertically
    bx = first + q;                                      // adjacent tiles share their halo rows in its L2
    return bx < (((x + 1) * a.gx) >> 3);
}
// q / d for 0 <= q < 2^20, 0 < d < 2^20 through one fp32 reciprocal-multiply (exact: the +0.5 keeps the quotient of an
// exact multiple away from the rounding edge); an integer division by a run-time value costs ~20 VALU instructions
__device__ __forceinline__ int fdiv(int q, float inv_d) { return (int)(((float)q + 0.5f) * inv_d); }

struct EpiCtx { int y0, g0, HoWo, RW, qlimit; float inv_RW, inv_Wo; bool vec_ok, pre0; };
__device__ __forceinline__ EpiCtx make_epi_ctx(const ConvArgs& a, int y0, int g0) {
    EpiCtx e;
    e.y0 = y0; e.g0 = g0; e.HoWo = a.Ho * a.Wo; e.RW = a.R * a.Wo; e.qlimit = a.G * e.RW;
    e.inv_RW = a.inv_RW; e.inv_Wo = a.inv_Wo;
    e.vec_ok = (e.RW % 4 == 0) && (e.HoWo % 4 == 0) && ((y0 * a.Wo) % 4 == 0);
    e.pre0 = e.vec_ok && a.n_add >= 1 && a.add_shift[0] == 0;     // addend 0 (the residual) can be fetched before the main loop
    return e;
}
// The residual / identity addend of a tile, loaded at kernel start so its HBM latency is hidden under the main loop.
__device__ __forceinline__ f32x4 prefetch_add0(const ConvArgs& a, const EpiCtx& e, int q, int co) {
    f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!e.pre0 || co >= a.Cout || q >= e.qlimit) return z;
    const int gl = fdiv(q, e.inv_RW), rem = q - gl * e.RW;
    const int img = e.g0 + gl, pix = e.y0 * a.Wo + rem;
    if (img >= a.N || pix >= e.HoWo) return z;
    return *reinterpret_cast<const f32x4*>(a.add[0] + ((size_t)img * a.add_ctot[0] + a.add_coff[0] + co) * e.HoWo + pix);
}
// `bias` is loaded by the caller BEFORE the first store of the epilogue: on CDNA loads and stores share vmcnt, so a load issued
// after a store can only be consumed once that store has completed -- a bias load per tile serialises the tiles' stores at one
// memory round trip each (measured in the bf16 twin of this epilogue: 0.75 us per tile).
__device__ __forceinline__ void store_tile(const ConvArgs& a, const EpiCtx& e, f32x4 v, int q, int co, f32x4 pre, float bias) {
    if (co >= a.Cout) return;
    if (e.vec_ok) {
        if (q >= e.qlimit) return;
        const int gl = fdiv(q, e.inv_RW), rem = q - gl * e.RW;
        const int img = e.g0 + gl, pix = e.y0 * a.Wo + rem;
        if (img >= a.N || pix >= e.HoWo) return;
        v += bias;
        if (e.pre0) v += pre;
#pragma unroll
        for (int k = 0; k < kMaxAdd; ++k) {                  // static indices: ConvArgs may live in registers
            if (k >= a.n_add) break;
            if (k == 0 && e.pre0) continue;
            const int sh = a.add_shift[k];
            if (sh == 0) {
                const float* ap = a.add[k] + ((size_t)img * a.add_ctot[k] + a.add_coff[k] + co) * e.HoWo + pix;
                v += *reinterpret_cast<const f32x4*>(ap);
            } else {
                const int hs = a.Ho >> sh, ws = a.Wo >> sh;
                const float* ap = a.add[k] + ((size_t)img * a.add_ctot[k] + a.add_coff[k] + co) * (hs * ws);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int y = fdiv(pix + r, e.inv_Wo), x = (pix + r) - y * a.Wo;
                    v[r] += ap[(y >> sh) * ws + (x >> sh)];
                }
            }
        }
        if (a.relu && co >= a.relu_from) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        float* op = a.out + ((size_t)img * a.out_ctot + a.out_coff + co) * e.HoWo + pix;
        *reinterpret_cast<f32x4*>(op) = v;
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = q + r;
            if (qq >= e.qlimit) continue;
            const int gl = fdiv(qq, e.inv_RW), rem = qq - gl * e.RW;
            const int img = e.g0 + gl, pix = e.y0 * a.Wo + rem;
            if (img >= a.N || pix >= e.HoWo) continue;
            float o = v[r] + bias;
#pragma unroll
            for (int k = 0; k < kMaxAdd; ++k) {
                if (k >= a.n_add) break;
                const int sh = a.add_shift[k];
                const int hs = a.Ho >> sh, ws = a.Wo >> sh;
                const int y = fdiv(pix, e.inv_Wo), x = pix - y * a.Wo;
                o += a.add[k][((size_t)img * a.add_ctot[k] + a.add_coff[k] + co) * (hs * ws) + (y >> sh) * ws + (x >> sh)];
            }
            if (a.relu && co >= a.relu_from) o = fmaxf(o, 0.f);
            a.out[((size_t)img * a.out_ctot + a.out_coff + co) * e.HoWo + pix] = o;
        }
    }
}

// Input staging modes.  ROWS (a.rows): the tile is whole image rows of ONE image and the plane size is a
// multiple of 4 floats, so each channel's patch is one contiguous, 16-byte-aligned range of the NCHW
// plane: LDS slot i of a channel plane holds global plane float (gal + i) and is filled by 16-byte
// LDS-DMA (1 KiB per wave-instruction instead of 256 B); rows above/below the image come from the
// zero block, and the left/right zero padding is applied when the A operand is read (lanes whose tap
// falls outside the row select 0).  Otherwise (multi-image tiles of the 7x7 maps): a zero-padded patch
// [G][Rin][Wp] gathered float by float through the source-offset table.

// CK = input channels per K-chunk: 8 for 3x3 (72 k-steps of 4 per chunk ... 18 MFMA steps), kConvCK1 = 16 for 1x1 convolutions:
// 8 would hold only 2 MFMA steps between barriers (latency-bound: 34.7 us for 64 -> 256 @56x56, 16 frames), 32 doubles the staging
// buffers to ~100 KB and leaves one workgroup per CU with nobody to overlap its load and store phases (35.0 us); 16: 32.1 us.
constexpr int kConvCK1 = 16;
template <bool ROWS, int KS, int S, int TPS, int TCS, int WP, int WC, int CK = (KS == 1 ? kConvCK1 : kConvCK)>
__global__ __launch_bounds__(WP* WC * 64) void conv_mfma_f32(const ConvArgs a) {
    constexpr int NT = WP * WC * 64, TC = TCS * 16;
    constexpr int PSW = TPS / WP, CSW = TCS / WC, TAPS = KS * KS;
    constexpr int WFLOATS = TAPS * CK * TC;              // weight slab of one chunk
    static_assert(TPS % WP == 0 && TCS % WC == 0, "wave grid must divide the tile");
    static_assert((WFLOATS / 4) % 64 == 0, "weight slab must be whole wave-instructions of 16 B/lane");

    extern __shared__ __align__(16) float smem[];       // ONE LDS object (hipcc: see guide 5 item 4a)
    float* w_lds = smem;                                 // [2][TAPS*CK][TC]   (cout index XOR-swizzled by row parity)
    float* in_lds = smem + 2 * WFLOATS;                  // [2][CK][PSTR]      (PSTR = padded G*Rin*Wp plane)
    int* tab = reinterpret_cast<int*>(in_lds + 2 * CK * a.PSTR);   // [PSTR] source offset of each plane slot, -1 = zero

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave / WC, wc = wave % WC;
    const int l15 = lane & 15, lq = lane >> 4;

    int bx, by;
    if (!xcd_block(a, bx, by)) return;
    GRK_TICK(t_start);
    const int ty = bx % a.tiles_y, grp = bx / a.tiles_y;
    const int y0 = ty * a.R, g0 = grp * a.G, co0 = by * TC;
    const int HW = a.H * a.W, RW = a.R * a.Wo, RinWp = a.Rin * a.Wp;
    constexpr int pad = KS / 2;
    const float* inb = a.in + ((size_t)g0 * a.in_ctot + a.in_coff) * HW;

    const int gal = floor4((y0 * S - pad) * a.W - 1);     // ROWS: global plane index of LDS slot 0
    if constexpr (!ROWS) {
        for (int idx = tid; idx < a.PSTR; idx += NT) {
            const int gl = idx / RinWp, rem = idx - gl * RinWp;
            const int ry = rem / a.Wp, rx = rem - ry * a.Wp;
            const int yin = y0 * S + ry - pad, xin = rx - pad;
            const bool ok = gl < a.G && (g0 + gl) < a.N && yin >= 0 && yin < a.H && xin >= 0 && xin < a.W;
            tab[idx] = ok ? gl * a.in_ctot * HW + yin * a.W + xin : -1;
        }
        __syncthreads();
    }

    // per-thread source offsets of the DMA units repeat every chunk (only the channel base moves): computed once
    constexpr int UW = WFLOATS / 4, UPRW = TC / 4, NWI = (UW + NT - 1) / NT;   // weight units (16 B) per thread
    int woff[NWI];
#pragma unroll
    for (int it = 0; it < NWI; ++it) {
        const int u = it * NT + tid;
        const int row = u / UPRW, j = u - row * UPRW;
        const int tap = row / CK, c = row - tap * CK;
        woff[it] = u < UW ? (tap * a.CinPad + c) * a.CoutPad + co0 + 4 * (j ^ ((row & 1) << 2)) : -1;
    }
    constexpr int NII = 4;                                  // row-mode input units per thread kept in registers
    const int upc_i = a.PSTR >> 2;
    const bool fast_in = ROWS && CK * upc_i <= NII * NT && (a.Cin % CK) == 0;
    int ioff[NII];
    if constexpr (ROWS) {
        const float inv_upc = a.inv_upc;
#pragma unroll
        for (int it = 0; it < NII; ++it) {
            const int u = it * NT + tid;
            const int c = fdiv(u, inv_upc), gi = gal + 4 * (u - c * upc_i);
            ioff[it] = (u < CK * upc_i) ? ((gi >= 0 && gi < HW) ? c * HW + gi : -2) : -1;
        }
    }

    auto issue = [&](int chunk, int buf) {
        const int c0 = chunk * CK;
        float* dst_in = in_lds + buf * CK * a.PSTR;
        if constexpr (ROWS) {
            if (fast_in) {
                const float* isrc = inb + (size_t)c0 * HW;
#pragma unroll
                for (int it = 0; it < NII; ++it)
                    if (it * NT + wave * 64 < CK * upc_i && ioff[it] != -1)
                        stage16(ioff[it] >= 0 ? isrc + ioff[it] : a.zeros, dst_in + (it * NT + wave * 64) * 4, lane);
            } else {
                const int upc = a.PSTR >> 2;                   // 16-byte units per channel plane
                for (int ub = wave * 64; ub < CK * upc; ub += NT) {
                    const int u = ub + lane;
                    if (u < CK * upc) {
                        const int c = u / upc, gi = gal + 4 * (u - c * upc);
                        const bool ok = (c0 + c) < a.Cin && gi >= 0 && gi < HW;
                        stage16(ok ? inb + (size_t)(c0 + c) * HW + gi : a.zeros, dst_in + ub * 4, lane);
                    }
                }
            }
        } else {
            for (int c = 0; c < CK; ++c) {
                const bool cvalid = (c0 + c) < a.Cin;
                const float* src_c = inb + (size_t)(c0 + c) * HW;
                for (int base = wave * 64; base < a.PSTR; base += NT) {
                    const int idx = base + lane;
                    if (idx < a.PSTR) {
                        const int off = tab[idx];
                        const float* src = (cvalid && off >= 0) ? src_c + off : a.zeros;
                        stage4(src, dst_in + c * a.PSTR + base, lane);
                    }
                }
            }
        }
        float* dst_w = w_lds + buf * WFLOATS;
        const float* wsrc = a.w + (size_t)c0 * a.CoutPad;   // cout index XOR-swizzled by row parity: same involution as the read side
#pragma unroll
        for (int it = 0; it < NWI; ++it)
            if (it * NT + wave * 64 < UW) stage16(wsrc + woff[it], dst_w + (it * NT + wave * 64) * 4, lane);   // UW % 64 == 0
    };

    f32x4 acc[PSW][CSW];
#pragma unroll
    for (int ps = 0; ps < PSW; ++ps)
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs) acc[ps][cs] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.CinPad / CK;
    GRK_TICK(t_issue);
    issue(0, 0);                                          // first chunk in flight while the lane offsets are computed
    const EpiCtx ec = make_epi_ctx(a, y0, g0);
    f32x4 pre[PSW][CSW];
#pragma unroll
    for (int ps = 0; ps < PSW; ++ps)
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs)
            pre[ps][cs] = prefetch_add0(a, ec, (wp * PSW + ps) * 16 + lq * 4, co0 + (wc * CSW + cs) * 16 + l15);
    float biasv[CSW];                                      // with the residual: nothing is loaded once the epilogue has started storing
#pragma unroll
    for (int cs = 0; cs < CSW; ++cs) {
        const int co = co0 + (wc * CSW + cs) * 16 + l15;
        biasv[cs] = co < a.Cout ? a.bias[co] : 0.f;
    }
    // A operand: lane holds pixel (lane&15) of its sub-tile, channel (lane>>4) of the k-group.
    int abase[PSW];
    unsigned lmask = 0, rmask = 0;                         // ROWS: sub-tiles whose pixel sits on the left / right image edge
    const float inv_RW = a.inv_RW, inv_Wo = a.inv_Wo;
#pragma unroll
    for (int ps = 0; ps < PSW; ++ps) {
        const int q = (wp * PSW + ps) * 16 + l15;
        // rows mode: one image per tile (gl = 0 wherever q < RW, and nothing below reads yl / x of the pixels past it)
        const int gl = ROWS ? 0 : fdiv(q, inv_RW), rem = q - gl * RW;
        const int yl = fdiv(rem, inv_Wo), x = rem - yl * a.Wo;
        int off;
        if constexpr (ROWS) {
            off = (q < RW) ? (y0 * S - pad + yl * S) * a.W + x * S - pad - gal : 0;
            if (x == 0) lmask |= 1u << ps;
            if (x * S + KS - 1 - pad >= a.W) rmask |= 1u << ps;
        } else {
            off = (q < a.G * RW) ? gl * RinWp + yl * S * a.Wp + x * S : 0;   // masked rows read slot 0
        }
        abase[ps] = lq * a.PSTR + off;
    }
    // B operand: lane holds cout (lane&15) of its sub-tile, row (lane>>4) of the k-group.
    int bbase[CSW];
#pragma unroll
    for (int cs = 0; cs < CSW; ++cs) bbase[cs] = lq * TC + (((wc * CSW + cs) * 16 + l15) ^ ((lq & 1) << 4));

#ifdef GRNET_ABLATION
    unsigned long long t_first = 0;
#endif
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of chunk ch has landed
        __syncthreads();                                     // ... and everybody else's; buf^1 is free
#ifdef GRNET_ABLATION
        if (ch == 0) t_first = __builtin_readcyclecounter();
#endif
        if (ch + 1 < nchunks && !GRK_DBG(a, 2)) issue(ch + 1, buf ^ 1);
        if (GRK_DBG(a, 1)) continue;
        const float* wi = w_lds + buf * WFLOATS;
        const float* xi = in_lds + buf * CK * a.PSTR;
        // k-steps of the chunk = (tap, channel group of 4); software pipeline: the LDS reads of step s+1
        // are in flight under the MFMAs of step s, and a step's operands are taken out of the LDS queue
        // while they are the only pending reads (lgkmcnt is a 4-bit counter).
        constexpr int STEPS = TAPS * (CK / 4);
        float av[2][PSW], bv[2][CSW];
        auto load_step = [&](int st, float* ar, float* br) {
            const int tap = st / (CK / 4), cg = st % (CK / 4);
            const int toff = (tap / KS) * a.Wp + (tap % KS);
#pragma unroll
            for (int cs = 0; cs < CSW; ++cs) br[cs] = wi[(tap * CK + cg * 4) * TC + bbase[cs]];
#pragma unroll
            for (int ps = 0; ps < PSW; ++ps) ar[ps] = xi[abase[ps] + cg * 4 * a.PSTR + toff];
        };
        load_step(0, av[0], bv[0]);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int cur = st & 1, tap = st / (CK / 4);
            float v[PSW], b[CSW];
#pragma unroll
            for (int ps = 0; ps < PSW; ++ps) {
                v[ps] = av[cur][ps];
                if constexpr (ROWS && KS == 3) {
                    if (tap % KS == 0) v[ps] = (lmask >> ps & 1) ? 0.f : v[ps];
                    if (tap % KS == 2 && S == 1) v[ps] = (rmask >> ps & 1) ? 0.f : v[ps];
                }
                asm volatile("" : "+v"(v[ps]));
            }
#pragma unroll
            for (int cs = 0; cs < CSW; ++cs) { b[cs] = bv[cur][cs]; asm volatile("" : "+v"(b[cs])); }
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < STEPS) load_step(st + 1, av[cur ^ 1], bv[cur ^ 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ps = 0; ps < PSW; ++ps)
#pragma unroll
                for (int cs = 0; cs < CSW; ++cs)
                    acc[ps][cs] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[ps], b[cs], acc[ps][cs], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue.  D: column (lane&15) = cout, rows (lane>>4)*4 + r = 4 consecutive pixels.
    if (GRK_DBG(a, 4)) return;
    GRK_TICK(t_loop);
#pragma unroll
    for (int cs = 0; cs < CSW; ++cs) {
        const int co = co0 + (wc * CSW + cs) * 16 + l15;
#pragma unroll
        for (int ps = 0; ps < PSW; ++ps) {
            if (GRK_DBG(a, 8) && (ps & 1)) continue;       // ablation: half of the tile stores
            if (GRK_DBG(a, 16)) { if (acc[ps][cs][0] == 12345.678f) a.out[0] = 1.f; continue; }   // ablation: no stores at all (results kept live)
            store_tile(a, ec, acc[ps][cs], (wp * PSW + ps) * 16 + lq * 4, co, pre[ps][cs], biasv[cs]);
        }
    }
#ifdef GRNET_ABLATION
    {
        GRK_TICK(t_end);
        GRK_PHASE_WK(0, t_start, t_issue);
        GRK_PHASE_WK(1, t_issue, t_first);
        GRK_PHASE_WK(2, t_first, t_loop);
        GRK_PHASE_WK(3, t_loop, t_end);
        if (threadIdx.x == 0) atomicAdd(&g_phase_wk[4], 1ull);
    }
#endif
}


// ---------------------------------------------------------------------------------------------
// Split-K variant for layers whose output is too small to fill 1024 SIMDs with whole-K tiles
// (the 7x7 / 14x14 / 28x28 branches and the 32-channel 56x56 branch at 16 frames per call).
// A workgroup owns ONE small output tile (PSW*16 pixels x CSW*16 channels); its NW (4 or 8) waves split the
// input channels in groups of 4 (= one MFMA k-step per filter tap), round-robin.  The waves share
// nothing but the source-offset table: each stages its OWN channel group (input patch + weight
// slab) by LDS-DMA into its own stage buffer and runs its own wait -> MFMA -> refill loop with
// no workgroup barrier; partial accumulators are summed through LDS in a fixed order at the end.

// MODE: 0 = gather (source-offset table), 1 = rows (contiguous image rows), 2 = planes (the 4 channel planes of a
// k-group of ONE whole small image are contiguous in NCHW: one 16-byte LDS-DMA per k-group; all zero padding is
// applied through a per-lane 9-bit tap-validity mask when the A operand is read).
// tid_in: the thread index as the caller wants the body to see it (the persistent dataflow kernel passes it through an opaque
// register copy per task, so that no lane-derived value of any body variant is loop-invariant across tasks and kept live in registers)
// WPT > 0 (rows mode, 3x3): the image width as a compile-time constant.  The row pitch of the patch then folds into the ds_read
// immediates, which frees registers for per-lane LEFT and RIGHT tap pointers: a lane whose pixel sits on the image's left (right) edge
// reads its kx = 0 (kx = 2) taps from a zero block of its wave instead of masking the value afterwards.  42 v_cndmask per stage of 63
// MFMAs are gone from the loop -- on gfx950 vector-ALU instructions inside an fp32 MFMA loop are matrix-pipe time.
constexpr int kEdgeZeros = 128;                          // floats of the per-wave zero block: offsets 0 .. 2 * WPT are read
template <int MODE, int KS, int S, int PSW, int CSW, int NW, int WPT = 0>
__device__ __forceinline__ void splitk_body(const ConvArgs& a, const int bx, const int by, float* smem, const int tid_in = -1) {
    constexpr int TC = CSW * 16, TAPS = KS * KS, WFL = TAPS * 4 * TC, NT = PSW * CSW;
    // planes mode (whole small images, all zero padding by tap validity) always reads invalid taps from the zero block: its row pitch is
    // already folded into per-(pixel, filter row) pointers, three of them (centre / left tap / right tap) instead of one.
    constexpr bool EDGEPTR = WPT > 0, ZPLANES = MODE == 2, ZBLOCK = EDGEPTR || ZPLANES;
    static_assert(!EDGEPTR || (MODE == 1 && KS == 3 && 2 * WPT + 3 <= kEdgeZeros), "edge pointers: rows mode, 3x3");
    const int tid = tid_in >= 0 ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    GRK_TICK(t_start);
    const int stage_floats = WFL + 4 * a.PSTR + (ZBLOCK ? kEdgeZeros : 0);   // [weights TAPS*4 x TC | input 4 x PSTR | zeros], ONE stage per wave:
    int* tab = reinterpret_cast<int*>(smem + NW * stage_floats);   // the co-resident waves hide the DMA (a second stage per wave
    float* mine = smem + wave * stage_floats;                        // measured no faster), and a fixed buffer keeps LDS addresses loop-invariant

    const int ty = bx % a.tiles_y, grp = bx / a.tiles_y;
    const int y0 = ty * a.R, g0 = grp * a.G, co0 = by * TC;
    const int HW = a.H * a.W, RW = a.R * a.Wo, RinWp = a.Rin * a.Wp;
    constexpr int pad = KS / 2;
    const float* inb = a.in + ((size_t)g0 * a.in_ctot + a.in_coff) * HW;

    constexpr bool ROWS = MODE == 1, PLANES = MODE == 2;
    const int gal = floor4((y0 * S - pad) * a.W - 1);
    if constexpr (MODE == 0) {
        for (int idx = tid; idx < a.PSTR; idx += NW * 64) {
            const int gl = idx / RinWp, rem = idx - gl * RinWp;
            const int ry = rem / a.Wp, rx = rem - ry * a.Wp;
            const int yin = y0 * S + ry - pad, xin = rx - pad;
            const bool ok = gl < a.G && (g0 + gl) < a.N && yin >= 0 && yin < a.H && xin >= 0 && xin < a.W;
            tab[idx] = ok ? gl * a.in_ctot * HW + yin * a.W + xin : -1;
        }
        __syncthreads();
    }

    // per-lane source offsets of this wave's DMA units are the same for every stage (only the channel group moves):
    // computed once, so a stage costs one 64-bit add per DMA instruction instead of divisions
    constexpr int U = WFL / 4, UPR = TC / 4, NWI = (U + 63) / 64;
    int woff[NWI];
#pragma unroll
    for (int it = 0; it < NWI; ++it) {
        const int u = it * 64 + lane;
        const int row = u / UPR, j = u - row * UPR;
        const int tap = row >> 2, c = row & 3;
        const int js = TC == 32 ? (j ^ ((row & 1) << 2)) : j;
        woff[it] = u < U ? (tap * a.CinPad + c) * a.CoutPad + co0 + 4 * js : -1;
    }
    constexpr int NII = 4;                                  // row-mode input units per lane kept in registers (PSTR <= 256)
    const bool fast_in = ROWS && a.PSTR <= 64 * NII && (a.Cin & 3) == 0;
    int ioff[NII];
    if constexpr (ROWS) {
        const int upc = a.PSTR >> 2;
        const float inv_upc = a.inv_upc;
#pragma unroll
        for (int it = 0; it < NII; ++it) {
            const int u2 = it * 64 + lane;
            const int c = fdiv(u2, inv_upc), gi = gal + 4 * (u2 - c * upc);
            ioff[it] = (u2 < 4 * upc) ? ((gi >= 0 && gi < HW) ? c * HW + gi : -2) : -1;     // -2: zero block, -1: no unit
        }
    }

    auto issue = [&](int grp4, int) {
        const int c0 = grp4 * 4;
        float* dst = mine;
        const float* wsrc = a.w + (size_t)c0 * a.CoutPad;
#pragma unroll
        for (int it = 0; it < NWI; ++it)
            if (woff[it] >= 0) stage16(wsrc + woff[it], dst + it * 256, lane);
        if constexpr (PLANES) {
            for (int ub = 0; ub < HW; ub += 64) {           // 4 planes x HW floats = HW 16-byte units, contiguous in HBM
                const int u2 = ub + lane;
                if (u2 < HW) stage16(inb + (size_t)c0 * HW + 4 * u2, dst + WFL + ub * 4, lane);
            }
        } else if constexpr (ROWS) {
            if (fast_in) {
                const float* isrc = inb + (size_t)c0 * HW;
                const int nin = (a.PSTR + 63) >> 6;
#pragma unroll
                for (int it = 0; it < NII; ++it)
                    if (it < nin && ioff[it] != -1) stage16(ioff[it] >= 0 ? isrc + ioff[it] : a.zeros, dst + WFL + it * 256, lane);
            } else {
                const int upc = a.PSTR >> 2;
                for (int ub = 0; ub < 4 * upc; ub += 64) {
                    const int u2 = ub + lane;
                    if (u2 < 4 * upc) {
                        const int c = u2 / upc, gi = gal + 4 * (u2 - c * upc);
                        const bool ok = (c0 + c) < a.Cin && gi >= 0 && gi < HW;
                        stage16(ok ? inb + (size_t)(c0 + c) * HW + gi : a.zeros, dst + WFL + ub * 4, lane);
                    }
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool cvalid = (c0 + c) < a.Cin;
                const float* src_c = inb + (size_t)(c0 + c) * HW;
                for (int base = 0; base < a.PSTR; base += 64) {
                    const int idx = base + lane;
                    if (idx < a.PSTR) {
                        const int off = tab[idx];
                        const float* src = (cvalid && off >= 0) ? src_c + off : a.zeros;
                        stage4(src, dst + WFL + c * a.PSTR + base, lane);
                    }
                }
            }
        }
    };

    f32x4 acc[PSW][CSW];
#pragma unroll
    for (int ps = 0; ps < PSW; ++ps)
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs) acc[ps][cs] = f32x4{0.f, 0.f, 0.f, 0.f};

    // One stage per wave at a time: issue -> wait -> MFMAs -> issue the next into the same buffer.
    const int ngroups = a.CinPad / 4;
    const int my_stages = ngroups > wave ? (ngroups - wave + NW - 1) / NW : 0;
    GRK_TICK(t_issue);
    if (my_stages > 0) issue(wave, 0);
    const EpiCtx ec = make_epi_ctx(a, y0, g0);
    constexpr int MAXT = (NT + NW - 1) / NW;                 // output tiles this wave finishes after the reduction
    f32x4 pre[MAXT];
    float biasv[MAXT];                                     // with the residual: nothing is loaded once the epilogue has started storing
#pragma unroll
    for (int i2 = 0; i2 < MAXT; ++i2) {
        const int t = wave + i2 * NW, ps = t / CSW, cs = t - ps * CSW;
        pre[i2] = t < NT ? prefetch_add0(a, ec, ps * 16 + lq * 4, co0 + cs * 16 + l15) : f32x4{0.f, 0.f, 0.f, 0.f};
        const int co = co0 + cs * 16 + l15;
        biasv[i2] = (t < NT && co < a.Cout) ? a.bias[co] : 0.f;
    }
    // (the first stages are in flight while the lane offsets and edge masks are computed)
    int abase[PSW];
    unsigned lmask = 0, rmask = 0;
    unsigned vmask[PLANES ? PSW : 1] = {};                  // PLANES: bit tap = that tap of this pixel is inside the image
    const float inv_RW = a.inv_RW, inv_Wo = a.inv_Wo;
#pragma unroll
    for (int ps = 0; ps < PSW; ++ps) {
        const int q = ps * 16 + l15;
        // rows / planes mode: one image per tile (gl = 0 wherever q < RW, and nothing below reads yl / x of the pixels past it)
        const int gl = (ROWS || PLANES) ? 0 : fdiv(q, inv_RW), rem = q - gl * RW;
        const int yl = fdiv(rem, inv_Wo), x = rem - yl * a.Wo;
        int off;
        if constexpr (PLANES) {
            off = (q < RW) ? (yl * S - pad) * a.W + x * S - pad : 0;     // may be negative: lands in the weight slab, masked
            if (q < RW) {                                   // tap (ky, kx) is inside the image iff its row and its column are: KS + KS tests, not KS * KS
                unsigned rb = 0, cb = 0;
#pragma unroll
                for (int k = 0; k < KS; ++k) {
                    const int yy = yl * S + k - pad, xx = x * S + k - pad;
                    if (yy >= 0 && yy < a.H) rb |= 1u << k;
                    if (xx >= 0 && xx < a.W) cb |= 1u << k;
                }
#pragma unroll
                for (int ky = 0; ky < KS; ++ky)
                    if (rb >> ky & 1) vmask[ps] |= cb << (ky * KS);
            }
        } else if constexpr (ROWS) {
            off = (q < RW) ? (y0 * S - pad + yl * S) * a.W + x * S - pad - gal : 0;
            if (x == 0) lmask |= 1u << ps;
            if (x * S + KS - 1 - pad >= a.W) rmask |= 1u << ps;
        } else {
            off = (q < a.G * RW) ? gl * RinWp + yl * S * a.Wp + x * S : 0;
        }
        abase[ps] = WFL + lq * a.PSTR + off;
    }
    int bbase[CSW];
#pragma unroll
    for (int cs = 0; cs < CSW; ++cs) bbase[cs] = lq * TC + ((cs * 16 + l15) ^ (TC == 32 ? ((lq & 1) << 4) : 0));

    // loop-invariant LDS row pointers: the kx offset of a tap becomes the immediate of ds_read_b32
    const float* arow[EDGEPTR ? 1 : PSW][KS];
    const float* pN[EDGEPTR ? PSW : 1], *pL[EDGEPTR ? PSW : 1], *pR[EDGEPTR ? PSW : 1];      // EDGEPTR: centre / left-tap / right-tap pointer of a pixel
    const float* qL[ZPLANES ? PSW : 1][KS], *qR[ZPLANES ? PSW : 1][KS];                      // ZPLANES: left / right-tap pointers per filter row (arow = centre)
    const float* brow[CSW];
    if constexpr (ZPLANES) {
        float* zreg = mine + WFL + 4 * a.PSTR;           // this wave's zero block (never a DMA target); same-wave LDS accesses stay in order
        zreg[lane] = 0.f;
        zreg[64 + lane] = 0.f;
#pragma unroll
        for (int ps = 0; ps < PSW; ++ps)
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const float* P = mine + abase[ps] + ky * a.Wp;
                const unsigned m = vmask[ps] >> (ky * KS);                // validity of the row's KS taps
                if constexpr (KS == 3) {
                    arow[ps][ky] = (m & 2u) ? P : zreg;                   // read at + 1
                    qL[ps][ky] = (m & 1u) ? P : zreg;                     // read at + 0
                    qR[ps][ky] = (m & 4u) ? P : zreg;                     // read at + 2
                } else {
                    arow[ps][ky] = (m & 1u) ? P : zreg;
                }
            }
    } else if constexpr (EDGEPTR) {
        float* zreg = mine + WFL + 4 * a.PSTR;           // this wave's zero block (never a DMA target); same-wave LDS accesses stay in order
        zreg[lane] = 0.f;
        zreg[64 + lane] = 0.f;
#pragma unroll
        for (int ps = 0; ps < PSW; ++ps) {
            const float* P = mine + abase[ps];
            pN[ps] = P;
            pL[ps] = (lmask >> ps & 1) ? zreg : P;                        // read at + ky * WPT + 0
            pR[ps] = (S == 1 && (rmask >> ps & 1)) ? zreg - 2 : P;        // read at + ky * WPT + 2
        }
    } else {
#pragma unroll
        for (int ps = 0; ps < PSW; ++ps)
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) arow[ps][ky] = mine + abase[ps] + ky * a.Wp;
    }
#pragma unroll
    for (int cs = 0; cs < CSW; ++cs) brow[cs] = mine + bbase[cs];
#ifdef GRNET_ABLATION
    unsigned long long t_first = 0;
#endif
    for (int i = 0; i < my_stages; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's stage has landed
#ifdef GRNET_ABLATION
        if (i == 0) t_first = __builtin_readcyclecounter();
#endif
        if (!GRK_DBG(a, 1)) {
            // software pipeline over the filter taps: the LDS reads of tap t+1 are in flight under the
            // MFMAs of tap t (one exposed LDS latency per stage instead of one per tap)
            // LD = taps prefetched ahead: 2 when a tap is only 4 MFMAs (128 cycles < LDS latency), else 1
            constexpr int LD = (PSW * CSW <= 4) ? 2 : 1;
            float av[LD + 1][PSW], bv[LD + 1][CSW];
            auto load_tap = [&](int tap, float* ar, float* br) {
#pragma unroll
                for (int cs = 0; cs < CSW; ++cs) br[cs] = brow[cs][tap * 4 * TC];
#pragma unroll
                for (int ps = 0; ps < PSW; ++ps) {
                    if constexpr (EDGEPTR) ar[ps] = (tap % KS == 0 ? pL[ps] : tap % KS == 2 ? pR[ps] : pN[ps])[(tap / KS) * WPT + tap % KS];
                    else if constexpr (ZPLANES && KS == 3) ar[ps] = (tap % KS == 0 ? qL[ps][tap / KS] : tap % KS == 2 ? qR[ps][tap / KS] : arow[ps][tap / KS])[tap % KS];
                    else ar[ps] = arow[ps][tap / KS][tap % KS];
                }
            };
#pragma unroll
            for (int t = 0; t < LD && t < TAPS; ++t) load_tap(t, av[t], bv[t]);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int cur = tap % (LD + 1);
                // (1) take this tap's operands out of the LDS queue while only the later prefetched taps are pending
                //     (lgkmcnt is a 4-bit counter: more than 15 reads in flight would force a full drain)
                float v[PSW], b[CSW];
#pragma unroll
                for (int ps = 0; ps < PSW; ++ps) {
                    v[ps] = av[cur][ps];
                    if constexpr (ROWS && KS == 3 && !EDGEPTR) {
                        if (tap % KS == 0) v[ps] = (lmask >> ps & 1) ? 0.f : v[ps];
                        if (tap % KS == 2 && S == 1) v[ps] = (rmask >> ps & 1) ? 0.f : v[ps];
                    }
                    if constexpr (PLANES && !ZPLANES) v[ps] = (vmask[ps] >> tap & 1) ? v[ps] : 0.f;
                    asm volatile("" : "+v"(v[ps]));
                }
#pragma unroll
                for (int cs = 0; cs < CSW; ++cs) { b[cs] = bv[cur][cs]; asm volatile("" : "+v"(b[cs])); }
                __builtin_amdgcn_sched_barrier(0);
                // (2) prefetch tap + LD, (3) this tap's MFMAs cover its LDS latency
                if (tap + LD < TAPS) load_tap(tap + LD, av[(tap + LD) % (LD + 1)], bv[(tap + LD) % (LD + 1)]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int ps = 0; ps < PSW; ++ps)
#pragma unroll
                    for (int cs = 0; cs < CSW; ++cs)
                        acc[ps][cs] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[ps], b[cs], acc[ps][cs], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (i + 1 < my_stages && !GRK_DBG(a, 2)) issue(wave + (i + 1) * NW, 0);          // refill the buffer just consumed
    }

    // ---- cross-wave reduction (fixed order -> deterministic), then the shared epilogue
    GRK_TICK(t_loop);
    __syncthreads();                                       // every wave is done with its staging buffers
    f32x4* red = reinterpret_cast<f32x4*>(smem);           // [NW][NT][64]
#pragma unroll
    for (int ps = 0; ps < PSW; ++ps)
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs) red[(wave * NT + ps * CSW + cs) * 64 + lane] = acc[ps][cs];
    __syncthreads();
    if (GRK_DBG(a, 4)) return;
    GRK_TICK(t_red);
#pragma unroll
    for (int i2 = 0; i2 < MAXT; ++i2) {
        const int t = wave + i2 * NW;
        if (t >= NT) break;
        f32x4 v = red[t * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[(w * NT + t) * 64 + lane];
        const int ps = t / CSW, cs = t - ps * CSW;
#ifdef GRNET_ABLATION
        const unsigned long long t_v = __builtin_readcyclecounter();
        if (i2 == 0) GRK_PHASE(6, t_red, t_v);                     // bias loads + partial sums of the first tile
#endif
        store_tile(a, ec, v, ps * 16 + lq * 4, co0 + cs * 16 + l15, pre[i2], biasv[i2]);
#ifdef GRNET_ABLATION
        if (i2 == 0) { const unsigned long long t_s = __builtin_readcyclecounter(); GRK_PHASE(7, t_v, t_s); }   // first tile's store_tile
#endif
    }
#ifdef GRNET_ABLATION
    {
        GRK_TICK(t_end);
        GRK_PHASE(0, t_start, t_issue);
        GRK_PHASE(1, t_issue, t_first);
        GRK_PHASE(2, t_first, t_loop);
        GRK_PHASE(3, t_loop, t_red);
        GRK_PHASE(4, t_red, t_end);
        if (threadIdx.x == 0) atomicAdd(&g_phase_f32[5], 1ull);
    }
#endif
}

template <int MODE, int KS, int S, int PSW, int CSW, int NW, int WPT = 0>
__global__ __launch_bounds__(NW * 64) void conv_splitk_f32(const ConvArgs a) {
    extern __shared__ __align__(16) float smem[];
    int bx, by;
    if (!xcd_block(a, bx, by)) return;
    splitk_body<MODE, KS, S, PSW, CSW, NW, WPT>(a, bx, by, smem);
}

// ---------------------------------------------------------------------------------------------
int conv_pick_tc(int Cout) { return Cout >= 64 ? 64 : 32; }

namespace {

constexpr size_t kMaxLds = 160 * 1024;
constexpr int kSplitWaves = 4;

static void plan_tile_geom(ConvArgs& a, int tps, int family) {
    const int TP = tps * 16, HoWo = a.Ho * a.Wo;
    if (HoWo <= TP) {
        a.G = TP / HoWo;
        if (a.G > a.N) a.G = a.N;
        a.R = a.Ho;
    } else {
        a.G = 1;
        a.R = TP / a.Wo;
        if (a.R > a.Ho) a.R = a.Ho;
    }
    if (a.R < 1) return;
    a.tiles_y = (a.Ho + a.R - 1) / a.R;
    a.groups = (a.N + a.G - 1) / a.G;
    a.Rin = (a.R - 1) * a.stride + a.ks;
    a.rows = (a.G == 1 && (a.H * a.W) % 4 == 0) ? 1 : 0;
    if (family == 1 && !a.rows && a.G == 1 && a.R == a.Ho && a.Cin % 4 == 0 && a.in_ctot % 4 == 0 && a.in_coff % 4 == 0 && a.H * a.W <= 256) {
        a.rows = 2;                                       // planes mode (split-K family only)
        a.Wp = a.W;
        a.PSTR = a.H * a.W;                               // plane stride = plane size: the 4 planes of a k-group stay contiguous
        return;
    }
    if (a.rows) {
        a.Wp = a.W;                                       // row pitch in LDS = image row pitch
        const int need = (a.Rin * a.W + 5 + 3) & ~3;      // phase (<=3) + 1 + rows + right overhang, in whole 16-byte units
        a.PSTR = ((need + 15) / 32) * 32 + 16;            // = 16 (mod 32): see below (stride-2 rows keep a 2-way conflict)
        if (a.PSTR < need) a.PSTR += 32;
        return;
    }
    a.Wp = (a.Wo - 1) * a.stride + a.ks;
    const int need = a.G * a.Rin * a.Wp;
    if (a.stride == 1) {
        // plane stride = 16 (mod 32): channel k and k+1 of a half-wave hit disjoint LDS banks
        a.PSTR = ((need + 15) / 32) * 32 + 16;
        if (a.PSTR < need) a.PSTR += 32;
    } else {
        a.PSTR = need | 1;   // stride-2 rows touch even banks; an odd plane stride moves channel k+1 to the odd ones
    }
}

// The reciprocals the kernels' index math divides by (fdiv): uniform values, but a float division in a kernel is ~10 vector
// instructions in EVERY lane of every wave -- and on gfx950 vector-ALU time is fp32-MFMA time.
void plan_tile(ConvArgs& a, int tps, int family) {
    plan_tile_geom(a, tps, family);
    if (a.R < 1) return;
    a.inv_RW = 1.0f / (float)(a.R * a.Wo);
    a.inv_Wo = 1.0f / (float)a.Wo;
    a.inv_upc = 1.0f / (float)(a.PSTR >> 2);
}

// A launch configuration: family 0 = whole-K tiles with a workgroup barrier per chunk (conv_mfma_f32),
// family 1 = split-K independent waves (conv_splitk_f32).
struct Cfg { int family, tps, tcs; int nw = 4; };   // nw: split-K waves per workgroup (4, or 8 = two per SIMD for launches of about one workgroup per CU)

// The image width the split-K kernel is compiled for (edge-pointer variants: splitk_body), 0 = the generic kernel.
int splitk_width_variant(const ConvArgs& a, const Cfg& c) {
    static const int on = GRNET_AB(EDGEPTR, 1);
    if (!on || c.family != 1 || a.rows != 1 || a.ks != 3 || c.tcs != 1) return 0;
    if (!((c.tps == 7 && (c.nw == 8 || c.nw == 4)) || (c.tps == 4 && c.nw == 8))) return 0;
    return (a.W == 56 || a.W == 28 || a.W == 14) ? a.W : 0;
}

size_t lds_bytes(const ConvArgs& a, const Cfg& c) {
    const int taps = a.ks * a.ks, TC = c.tcs * 16;
    const size_t tab = a.rows ? 0 : a.PSTR;
    const size_t ck = a.ks == 1 ? kConvCK1 : kConvCK;
    if (c.family == 0) return sizeof(float) * (2 * (size_t)taps * ck * TC + 2 * ck * a.PSTR + tab);
    const size_t stage = (size_t)taps * 4 * TC + 4 * (size_t)a.PSTR + ((splitk_width_variant(a, c) || a.rows == 2) ? kEdgeZeros : 0);
    const size_t staging = c.nw * stage + tab, red = (size_t)c.nw * c.tps * c.tcs * 256;
    return sizeof(float) * (staging > red ? staging : red);
}

// MFMA-issue model: the chip has 1024 SIMDs; a wave issues `chain` MFMAs of 32 cycles back to back.
// cost ~ chain x number of rounds the waves need; split-K pays its extra staging traffic as a 15 % penalty.
double cfg_cost(ConvArgs a, const Cfg& c, bool* ok) {
    plan_tile(a, c.tps, c.family);
    *ok = a.R >= 1 && lds_bytes(a, c) <= kMaxLds && a.CoutPad % (c.tcs * 16) == 0 &&
          !(c.family == 0 && a.ks == 1 && a.CinPad % kConvCK1 != 0);       // whole-K 1x1 tiles stage kConvCK1 channels per chunk
    if (!*ok) return 0;
    const double blocks = (double)a.tiles_y * a.groups * (a.CoutPad / (c.tcs * 16));
    const int kgroups = a.CinPad / 4, taps = a.ks * a.ks;
    double waves, chain;
    if (c.family == 0) {
        const int nw = (c.tps == 14) ? 4 : (c.tcs == 4 ? 4 : 2);
        waves = blocks * nw;
        chain = (double)kgroups * taps * 7 * (c.tps == 14 ? c.tcs / 2 : 1);
    } else {
        waves = blocks * c.nw;
        chain = (double)((kgroups + c.nw - 1) / c.nw) * taps * c.tps * c.tcs;
    }
    const double rounds = waves / 1024.0 < 1.0 ? 1.0 : waves / 1024.0;
    double cost = chain * rounds * (c.family == 1 ? 1.15 : 1.0) + 300.0;   // + fixed prologue/epilogue
    // whole-K tiles of a 1x1 convolution do ~14 MFMAs per 8-channel chunk: with less than one workgroup per CU
    // nothing covers the chunk's DMA + barrier latency (measured: 128->25 @56x56 took 108 us instead of 19)
    if (c.family == 0 && a.ks == 1 && blocks < 256) cost += (kgroups / 2.0) * 400.0;
    // bandwidth-leaning layers (1x1 on 56x56 maps, the 3-channel stem) run their load / MFMA / store phases back to back
    // inside a workgroup: twice as many half-size workgroups overlap them better (measured 39.5 -> 34.6 us, 51.5 -> 39.8 us)
    if (c.family == 0 && c.tps == 14 && (a.ks == 1 || a.Cin <= 8)) cost *= 1.05;
    return cost;
}

template <typename K>
hipError_t set_lds(K kern) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds);
}

#define GRK_FOR_KS(M) M(1, 1) M(3, 1) M(3, 2)

template <bool ROWS, int KS, int S>
hipError_t init_ks() {
    hipError_t e;
    if ((e = set_lds(conv_mfma_f32<ROWS, KS, S, 14, 4, 2, 2>)) != hipSuccess) return e;
    if ((e = set_lds(conv_mfma_f32<ROWS, KS, S, 14, 2, 2, 2>)) != hipSuccess) return e;
    if ((e = set_lds(conv_mfma_f32<ROWS, KS, S, 7, 4, 1, 4>)) != hipSuccess) return e;
    if ((e = set_lds(conv_mfma_f32<ROWS, KS, S, 7, 2, 1, 2>)) != hipSuccess) return e;
    if ((e = set_lds(conv_splitk_f32<ROWS ? 1 : 0, KS, S, 7, 1, kSplitWaves>)) != hipSuccess) return e;
    if ((e = set_lds(conv_splitk_f32<ROWS ? 1 : 0, KS, S, 7, 2, kSplitWaves>)) != hipSuccess) return e;
    if ((e = set_lds(conv_splitk_f32<ROWS ? 1 : 0, KS, S, 4, 1, kSplitWaves>)) != hipSuccess) return e;
    if ((e = set_lds(conv_splitk_f32<ROWS ? 1 : 0, KS, S, 4, 2, kSplitWaves>)) != hipSuccess) return e;
    if ((e = set_lds(conv_splitk_f32<ROWS ? 1 : 0, KS, S, 7, 1, 8>)) != hipSuccess) return e;
    if ((e = set_lds(conv_splitk_f32<ROWS ? 1 : 0, KS, S, 4, 1, 8>)) != hipSuccess) return e;
    if constexpr (ROWS && KS == 3) {
#define GRK_SET_W(W) \
        if ((e = set_lds(conv_splitk_f32<1, KS, S, 7, 1, 8, W>)) != hipSuccess) return e; \
        if ((e = set_lds(conv_splitk_f32<1, KS, S, 7, 1, kSplitWaves, W>)) != hipSuccess) return e; \
        if ((e = set_lds(conv_splitk_f32<1, KS, S, 4, 1, 8, W>)) != hipSuccess) return e;
        GRK_SET_W(56) GRK_SET_W(28) GRK_SET_W(14)
#undef GRK_SET_W
    }
    if (!ROWS) {
        if ((e = set_lds(conv_splitk_f32<2, KS, S, 4, 1, kSplitWaves>)) != hipSuccess) return e;
        if ((e = set_lds(conv_splitk_f32<2, KS, S, 4, 2, kSplitWaves>)) != hipSuccess) return e;
        if ((e = set_lds(conv_splitk_f32<2, KS, S, 4, 1, 8>)) != hipSuccess) return e;
    }
    return hipSuccess;
}

template <bool ROWS, int KS, int S>
hipError_t dispatch(const ConvArgs& a_in, const Cfg& c, size_t lds, hipStream_t s) {
    ConvArgs a = a_in;
    a.gx = a.tiles_y * a.groups;
    a.gy = a.CoutPad / (c.tcs * 16);
    static const int xcd_env = GRNET_AB(XCD_ORDER, 3);   // A/B runs: 0 plain (tile, block) order, 1 tile-major only
    const double in_bytes = 4.0 * a.N * a.Cin * a.H * a.W, w_bytes = 4.0 * a.ks * a.ks * a.Cin * a.Cout;
    a.gx8 = (a.gx + 7) / 8;
    a.xcd = 0;
    if ((xcd_env & 2) && a.gy % 8 == 0 && w_bytes > in_bytes) a.xcd = 2;
    else if ((xcd_env & 1) && (a.gy > 1 || a.ks > 1) && a.gx >= 16) a.xcd = 1;
    const dim3 grid((a.xcd == 1 ? a.gx8 * 8 : a.gx) * a.gy);
    if (a.rows == 2) {
        if (c.family == 1 && c.tps == 4 && c.tcs == 1 && c.nw == 8) return launch_k(conv_splitk_f32<2, KS, S, 4, 1, 8>, grid, dim3(512), lds, s, a);
        if (c.family == 1 && c.tps == 4 && c.tcs == 1) return launch_k(conv_splitk_f32<2, KS, S, 4, 1, kSplitWaves>, grid, dim3(kSplitWaves * 64), lds, s, a);
        if (c.family == 1 && c.tps == 4 && c.tcs == 2) return launch_k(conv_splitk_f32<2, KS, S, 4, 2, kSplitWaves>, grid, dim3(kSplitWaves * 64), lds, s, a);
        return hipErrorInvalidValue;
    }
#define GRK_LAUNCH(KERN, THREADS) return launch_k(KERN, grid, dim3(THREADS), lds, s, a)
    if (c.family == 0) {
        if (c.tps == 14 && c.tcs == 4) GRK_LAUNCH((conv_mfma_f32<ROWS, KS, S, 14, 4, 2, 2>), 256);
        if (c.tps == 14 && c.tcs == 2) GRK_LAUNCH((conv_mfma_f32<ROWS, KS, S, 14, 2, 2, 2>), 256);
        if (c.tps == 7 && c.tcs == 4) GRK_LAUNCH((conv_mfma_f32<ROWS, KS, S, 7, 4, 1, 4>), 256);
        if (c.tps == 7 && c.tcs == 2) GRK_LAUNCH((conv_mfma_f32<ROWS, KS, S, 7, 2, 1, 2>), 128);
    } else {
        if constexpr (ROWS && KS == 3) {                   // width-specialised variants without edge masks in the loop (splitk_body)
            const int wv = splitk_width_variant(a, c);
#define GRK_LAUNCH_W(W) \
            if (wv == W) { \
                if (c.tps == 7 && c.nw == 8) GRK_LAUNCH((conv_splitk_f32<1, KS, S, 7, 1, 8, W>), 512); \
                if (c.tps == 7 && c.nw == 4) GRK_LAUNCH((conv_splitk_f32<1, KS, S, 7, 1, kSplitWaves, W>), kSplitWaves * 64); \
                if (c.tps == 4 && c.nw == 8) GRK_LAUNCH((conv_splitk_f32<1, KS, S, 4, 1, 8, W>), 512); \
            }
            GRK_LAUNCH_W(56) GRK_LAUNCH_W(28) GRK_LAUNCH_W(14)
#undef GRK_LAUNCH_W
        }
        if (c.tps == 7 && c.tcs == 1 && c.nw == 8) GRK_LAUNCH((conv_splitk_f32<ROWS ? 1 : 0, KS, S, 7, 1, 8>), 512);
        if (c.tps == 4 && c.tcs == 1 && c.nw == 8) GRK_LAUNCH((conv_splitk_f32<ROWS ? 1 : 0, KS, S, 4, 1, 8>), 512);
        if (c.tps == 7 && c.tcs == 1) GRK_LAUNCH((conv_splitk_f32<ROWS ? 1 : 0, KS, S, 7, 1, kSplitWaves>), kSplitWaves * 64);
        if (c.tps == 7 && c.tcs == 2) GRK_LAUNCH((conv_splitk_f32<ROWS ? 1 : 0, KS, S, 7, 2, kSplitWaves>), kSplitWaves * 64);
        if (c.tps == 4 && c.tcs == 1) GRK_LAUNCH((conv_splitk_f32<ROWS ? 1 : 0, KS, S, 4, 1, kSplitWaves>), kSplitWaves * 64);
        if (c.tps == 4 && c.tcs == 2) GRK_LAUNCH((conv_splitk_f32<ROWS ? 1 : 0, KS, S, 4, 2, kSplitWaves>), kSplitWaves * 64);
    }
#undef GRK_LAUNCH
    return hipErrorInvalidValue;
}

template <bool ROWS>
hipError_t dispatch_ks(const ConvArgs& a, const Cfg& c, size_t lds, hipStream_t s) {
    if (a.ks == 1) return dispatch<ROWS, 1, 1>(a, c, lds, s);
    return a.stride == 1 ? dispatch<ROWS, 3, 1>(a, c, lds, s) : dispatch<ROWS, 3, 2>(a, c, lds, s);
}

}  // namespace

hipError_t conv_init() {
    hipError_t e;
    if ((e = init_ks<false, 1, 1>()) != hipSuccess) return e;
    if ((e = init_ks<false, 3, 1>()) != hipSuccess) return e;
    if ((e = init_ks<false, 3, 2>()) != hipSuccess) return e;
    if ((e = init_ks<true, 1, 1>()) != hipSuccess) return e;
    if ((e = init_ks<true, 3, 1>()) != hipSuccess) return e;
    if ((e = init_ks<true, 3, 2>()) != hipSuccess) return e;
    return hipSuccess;
}

const char* conv_dominant_kernel_name() { return "conv_mfma_f32 / conv_splitk_f32"; }

// tile_hint: 0 = cost model; 7 / 14 = whole-K family with that pixel tile; 1000 + 10*psw + csw = split-K
// family (1071, 1072, 1041, 1042), + 100 = eight split-K waves per workgroup (1171, 1141).  Hints exist for the per-kernel parity tests and for tuning.
hipError_t launch_conv(ConvArgs a, hipStream_t s, int tile_hint) {
    const int TCpack = conv_pick_tc(a.Cout);
    if (a.CoutPad % TCpack != 0 || a.CinPad % kConvCK != 0) return hipErrorInvalidValue;
    if (!((a.ks == 1 && a.stride == 1) || (a.ks == 3 && (a.stride == 1 || a.stride == 2)))) return hipErrorInvalidValue;
    if (a.Wo > 14 * 16) return hipErrorInvalidValue;
    Cfg best{0, 7, TCpack / 16};
    bool found = false;
    if (tile_hint == 7 || tile_hint == 14) {
        best = Cfg{0, tile_hint, TCpack / 16};
        cfg_cost(a, best, &found);
        if (!found && tile_hint == 14) { best.tps = 7; cfg_cost(a, best, &found); }
    } else if (tile_hint >= 1000) {
        const int code = tile_hint - 1000, wide = code >= 100;                   // 11xx: 8 split-K waves (csw 1 only)
        best = Cfg{1, (code % 100) / 10, code % 10, wide ? 8 : 4};
        if (!((best.tps == 7 || best.tps == 4) && (best.tcs == 1 || (best.tcs == 2 && !wide)))) return hipErrorInvalidValue;
        cfg_cost(a, best, &found);
    } else {
        const Cfg cands[] = {{0, 14, TCpack / 16}, {0, 7, TCpack / 16}, {1, 7, 2}, {1, 7, 1}, {1, 4, 2}, {1, 4, 1}};
        double best_cost = 0;
        for (const Cfg& c : cands) {
            if (c.family == 0 && c.tps == 7 && a.Wo > 7 * 16) continue;
            if (c.family == 1 && a.Wo > c.tps * 16) continue;
            bool ok;
            const double cost = cfg_cost(a, c, &ok);
            if (ok && (!found || cost < best_cost)) { best = c; best_cost = cost; found = true; }
        }
        // a split-K launch of at most ~2 workgroups per CU leaves each SIMD one or two waves: eight waves per workgroup
        // (half the channels each) give the MFMA pipe a second wave to switch to (measured: 7x7 25.0 -> 21.3 us,
        // 14x14 17.9 -> 15.7, 28x28 16.0 -> 15.4; the 896-block 56x56 launches lose)
        static const int nw8_env = GRNET_AB(NW8, 1);
        if (found && nw8_env && best.family == 1 && best.tcs == 1) {
            ConvArgs t = a;
            plan_tile(t, best.tps, 1);
            const long blocks = (long)t.tiles_y * t.groups * (a.CoutPad / 16);
            Cfg wide = best;
            wide.nw = 8;
            bool ok;
            cfg_cost(a, wide, &ok);
            if (ok && blocks <= 512 && a.CinPad >= 64) best = wide;
        }
    }
    if (!found) return hipErrorInvalidValue;
    plan_tile(a, best.tps, best.family);
    a.TC = best.tcs * 16;
    const size_t lds = lds_bytes(a, best);
    const hipError_t e = a.rows == 1 ? dispatch_ks<true>(a, best, lds, s) : dispatch_ks<false>(a, best, lds, s);
#ifdef GRNET_ABLATION
    static const bool phases = GRNET_AB_SET(F32_PHASES);
    if (phases && e == hipSuccess && best.family == 0) {
        unsigned long long h[8] = {}, z[8] = {};
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_wk), sizeof(h));
        hipMemcpyToSymbol(HIP_SYMBOL(g_phase_wk), z, sizeof(z));
        const double n = h[4] ? (double)h[4] : 1.0;
        fprintf(stderr, "[f32 whole-K phases] %d->%d k%d s%d %dx%d N%d tps %d tcs %d wgs %llu: per WG ticks  index math %.0f  first wait %.0f  chunk loop %.0f "
                "(%d chunks)  epilogue %.0f\n", a.Cin, a.Cout, a.ks, a.stride, a.H, a.W, a.N, best.tps, best.tcs, h[4], h[0] / n, h[1] / n, h[2] / n,
                a.CinPad / (a.ks == 1 ? kConvCK1 : kConvCK), h[3] / n);
    }
    if (phases && e == hipSuccess && best.family == 1) {
        unsigned long long h[8] = {}, z[8] = {};
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_f32), sizeof(h));
        hipMemcpyToSymbol(HIP_SYMBOL(g_phase_f32), z, sizeof(z));
        const double n = h[5] ? (double)h[5] : 1.0;
        fprintf(stderr, "[f32 split-K phases] %d->%d k%d s%d %dx%d N%d psw %d csw %d nw %d wgs %llu: per WG ticks  index math %.0f  first wait %.0f  "
                "stage loop %.0f  reduction %.0f  epilogue %.0f (first tile: sums %.0f, store_tile %.0f)\n", a.Cin, a.Cout, a.ks, a.stride, a.H, a.W, a.N,
                best.tps, best.tcs, best.nw, h[5], h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[6] / n, h[7] / n);
    }
#endif
    return e;
}

}  // namespace grk
