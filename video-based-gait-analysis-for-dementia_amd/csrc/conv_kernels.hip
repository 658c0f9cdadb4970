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

namespace grk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
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

template <int KS, int S, int TPS, int TCS, int WP, int WC>
__global__ __launch_bounds__(WP* WC * 64) void conv_mfma_f32(const ConvArgs a) {
    constexpr int NT = WP * WC * 64, CK = kConvCK, TC = TCS * 16;
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

    const int ty = blockIdx.x % a.tiles_y, grp = blockIdx.x / a.tiles_y;
    const int y0 = ty * a.R, g0 = grp * a.G, co0 = blockIdx.y * TC;
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo, RW = a.R * a.Wo, RinWp = a.Rin * a.Wp;
    constexpr int pad = KS / 2;
    const float* inb = a.in + ((size_t)g0 * a.in_ctot + a.in_coff) * HW;

    for (int idx = tid; idx < a.PSTR; idx += NT) {
        const int gl = idx / RinWp, rem = idx - gl * RinWp;
        const int ry = rem / a.Wp, rx = rem - ry * a.Wp;
        const int yin = y0 * S + ry - pad, xin = rx - pad;
        const bool ok = gl < a.G && (g0 + gl) < a.N && yin >= 0 && yin < a.H && xin >= 0 && xin < a.W;
        tab[idx] = ok ? gl * a.in_ctot * HW + yin * a.W + xin : -1;
    }
    __syncthreads();

    // A operand: lane holds pixel (lane&15) of its sub-tile, channel (lane>>4) of the k-group.
    int abase[PSW];
#pragma unroll
    for (int ps = 0; ps < PSW; ++ps) {
        const int q = (wp * PSW + ps) * 16 + l15;
        const int gl = q / RW, rem = q - gl * RW;
        const int yl = rem / a.Wo, x = rem - yl * a.Wo;
        const int off = (q < a.G * RW) ? gl * RinWp + yl * S * a.Wp + x * S : 0;   // masked rows read slot 0
        abase[ps] = lq * a.PSTR + off;
    }
    // B operand: lane holds cout (lane&15) of its sub-tile, row (lane>>4) of the k-group.
    int bbase[CSW];
#pragma unroll
    for (int cs = 0; cs < CSW; ++cs) bbase[cs] = lq * TC + (((wc * CSW + cs) * 16 + l15) ^ ((lq & 1) << 4));

    auto issue = [&](int chunk, int buf) {
        const int c0 = chunk * CK;
        float* dst_in = in_lds + buf * CK * a.PSTR;
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
        float* dst_w = w_lds + buf * WFLOATS;
        constexpr int U = WFLOATS / 4, UPR = TC / 4;     // 16-byte units, units per weight row
#pragma unroll
        for (int ub = wave * 64; ub < U; ub += NT) {
            const int u = ub + lane;
            const int row = u / UPR, j = u - row * UPR;
            const int tap = row / CK, c = row - tap * CK;
            const int js = j ^ ((row & 1) << 2);         // same involution as the read side (rule 21)
            const float* src = a.w + ((size_t)(tap * a.CinPad + c0 + c) * a.CoutPad + co0 + 4 * js);
            stage16(src, dst_w + ub * 4, lane);
        }
    };

    f32x4 acc[PSW][CSW];
#pragma unroll
    for (int ps = 0; ps < PSW; ++ps)
#pragma unroll
        for (int cs = 0; cs < CSW; ++cs) acc[ps][cs] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.CinPad / CK;
    issue(0, 0);
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of chunk ch has landed
        __syncthreads();                                     // ... and everybody else's; buf^1 is free
        if (ch + 1 < nchunks) issue(ch + 1, buf ^ 1);
        const float* wi = w_lds + buf * WFLOATS;
        const float* xi = in_lds + buf * CK * a.PSTR;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int toff = (tap / KS) * a.Wp + (tap % KS);
#pragma unroll
            for (int cg = 0; cg < CK / 4; ++cg) {
                float av[PSW], bv[CSW];
#pragma unroll
                for (int cs = 0; cs < CSW; ++cs) bv[cs] = wi[(tap * CK + cg * 4) * TC + bbase[cs]];
#pragma unroll
                for (int ps = 0; ps < PSW; ++ps) av[ps] = xi[abase[ps] + cg * 4 * a.PSTR + toff];
#pragma unroll
                for (int ps = 0; ps < PSW; ++ps)
#pragma unroll
                    for (int cs = 0; cs < CSW; ++cs)
                        acc[ps][cs] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ps], bv[cs], acc[ps][cs], 0, 0, 0);
            }
        }
    }

    // ---- epilogue.  D: column (lane&15) = cout, rows (lane>>4)*4 + r = 4 consecutive pixels.
    const bool vec_ok = (RW % 4 == 0) && (HoWo % 4 == 0) && ((y0 * a.Wo) % 4 == 0);
    const int qlimit = a.G * RW;
#pragma unroll
    for (int cs = 0; cs < CSW; ++cs) {
        const int co = co0 + (wc * CSW + cs) * 16 + l15;
        if (co >= a.Cout) continue;
        const float bias = a.bias[co];
#pragma unroll
        for (int ps = 0; ps < PSW; ++ps) {
            const int q = (wp * PSW + ps) * 16 + lq * 4;
            f32x4 v = acc[ps][cs];
            if (vec_ok) {
                if (q >= qlimit) continue;
                const int gl = q / RW, rem = q - gl * RW;
                const int img = g0 + gl, pix = y0 * a.Wo + rem;
                if (img >= a.N || pix >= HoWo) continue;
                v += bias;
                for (int k = 0; k < a.n_add; ++k) {
                    const int sh = a.add_shift[k];
                    if (sh == 0) {
                        const float* ap = a.add[k] + ((size_t)img * a.add_ctot[k] + a.add_coff[k] + co) * HoWo + pix;
                        v += *reinterpret_cast<const f32x4*>(ap);
                    } else {
                        const int hs = a.Ho >> sh, ws = a.Wo >> sh;
                        const float* ap = a.add[k] + ((size_t)img * a.add_ctot[k] + a.add_coff[k] + co) * (hs * ws);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int y = (pix + r) / a.Wo, x = (pix + r) - y * a.Wo;
                            v[r] += ap[(y >> sh) * ws + (x >> sh)];
                        }
                    }
                }
                if (a.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                }
                float* op = a.out + ((size_t)img * a.out_ctot + a.out_coff + co) * HoWo + pix;
                *reinterpret_cast<f32x4*>(op) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = q + r;
                    if (qq >= qlimit) continue;
                    const int gl = qq / RW, rem = qq - gl * RW;
                    const int img = g0 + gl, pix = y0 * a.Wo + rem;
                    if (img >= a.N || pix >= HoWo) continue;
                    float o = v[r] + bias;
                    for (int k = 0; k < a.n_add; ++k) {
                        const int sh = a.add_shift[k];
                        const int hs = a.Ho >> sh, ws = a.Wo >> sh;
                        const int y = pix / a.Wo, x = pix - y * a.Wo;
                        o += a.add[k][((size_t)img * a.add_ctot[k] + a.add_coff[k] + co) * (hs * ws) + (y >> sh) * ws + (x >> sh)];
                    }
                    if (a.relu) o = fmaxf(o, 0.f);
                    a.out[((size_t)img * a.out_ctot + a.out_coff + co) * HoWo + pix] = o;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
int conv_pick_tc(int Cout) { return Cout >= 64 ? 64 : 32; }

namespace {

template <int KS, int S, int TPS, int TCS, int WP, int WC>
hipError_t launch_one(const ConvArgs& a, size_t lds_bytes, hipStream_t s) {
    dim3 grid(a.tiles_y * a.groups, a.CoutPad / (TCS * 16));
    dim3 block(WP * WC * 64);
    hipLaunchKernelGGL((conv_mfma_f32<KS, S, TPS, TCS, WP, WC>), grid, block, lds_bytes, s, a);
    return hipGetLastError();
}

template <int KS, int S, int TPS, int TCS, int WP, int WC>
hipError_t set_lds_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_f32<KS, S, TPS, TCS, WP, WC>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <int KS, int S>
hipError_t init_ks() {
    hipError_t e;
    if ((e = set_lds_attr<KS, S, 14, 4, 2, 2>()) != hipSuccess) return e;
    if ((e = set_lds_attr<KS, S, 14, 2, 2, 2>()) != hipSuccess) return e;
    if ((e = set_lds_attr<KS, S, 7, 4, 1, 4>()) != hipSuccess) return e;
    if ((e = set_lds_attr<KS, S, 7, 2, 1, 2>()) != hipSuccess) return e;
    return hipSuccess;
}

template <int KS, int S>
hipError_t dispatch_tile(const ConvArgs& a, int tps, size_t lds, hipStream_t s) {
    if (tps == 14) {
        return a.TC == 64 ? launch_one<KS, S, 14, 4, 2, 2>(a, lds, s) : launch_one<KS, S, 14, 2, 2, 2>(a, lds, s);
    }
    return a.TC == 64 ? launch_one<KS, S, 7, 4, 1, 4>(a, lds, s) : launch_one<KS, S, 7, 2, 1, 2>(a, lds, s);
}

void plan_tile(ConvArgs& a, int tps) {
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
    a.tiles_y = (a.Ho + a.R - 1) / a.R;
    a.groups = (a.N + a.G - 1) / a.G;
    a.Rin = (a.R - 1) * a.stride + a.ks;
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

}  // namespace

hipError_t conv_init() {
    hipError_t e;
    if ((e = init_ks<1, 1>()) != hipSuccess) return e;
    if ((e = init_ks<3, 1>()) != hipSuccess) return e;
    if ((e = init_ks<3, 2>()) != hipSuccess) return e;
    return hipSuccess;
}

const char* conv_dominant_kernel_name() { return "conv_mfma_f32"; }

hipError_t launch_conv(ConvArgs a, hipStream_t s, int tile_hint) {
    a.TC = conv_pick_tc(a.Cout);
    if (a.CoutPad % a.TC != 0 || a.CinPad % kConvCK != 0) return hipErrorInvalidValue;
    if (!((a.ks == 1 && a.stride == 1) || (a.ks == 3 && (a.stride == 1 || a.stride == 2)))) return hipErrorInvalidValue;
    if (a.Wo > 14 * 16) return hipErrorInvalidValue;
    int tps = tile_hint;
    if (tps != 7 && tps != 14) {
        ConvArgs t = a;
        plan_tile(t, 14);
        const long blocks14 = (long)t.tiles_y * t.groups * (a.CoutPad / a.TC);
        tps = (blocks14 >= 512 && a.Wo <= 14 * 16) ? 14 : 7;
        if (a.Wo > 7 * 16) tps = 14;
    }
    plan_tile(a, tps);
    if (a.R < 1) return hipErrorInvalidValue;
    const int taps = a.ks * a.ks;
    const size_t lds = sizeof(float) * (2 * (size_t)taps * kConvCK * a.TC + 2 * (size_t)kConvCK * a.PSTR + a.PSTR);
    if (lds > 160 * 1024) {
        if (tps == 14) {   // fall back to the smaller pixel tile
            plan_tile(a, 7);
            tps = 7;
            const size_t lds7 = sizeof(float) * (2 * (size_t)taps * kConvCK * a.TC + 2 * (size_t)kConvCK * a.PSTR + a.PSTR);
            if (lds7 > 160 * 1024 || a.R < 1) return hipErrorInvalidValue;
            if (a.ks == 1) return dispatch_tile<1, 1>(a, tps, lds7, s);
            return a.stride == 1 ? dispatch_tile<3, 1>(a, tps, lds7, s) : dispatch_tile<3, 2>(a, tps, lds7, s);
        }
        return hipErrorInvalidValue;
    }
    if (a.ks == 1) return dispatch_tile<1, 1>(a, tps, lds, s);
    return a.stride == 1 ? dispatch_tile<3, 1>(a, tps, lds, s) : dispatch_tile<3, 2>(a, tps, lds, s);
}

}  // namespace grk
