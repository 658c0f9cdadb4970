// The "up" half of an HR module's fuse layer in ONE launch (round 4).  Reference: HighResolutionModule.forward, hrnet.py:249-267, with
// the fuse layers built at hrnet.py:189-244: for output i, every source branch j > i contributes nearest_up_{2^(j-i)}(BN(conv1x1(x_j)))
// and branch i contributes x_i itself.  Until round 3 every (i, j) term was its own 1x1 convolution launch (31 per forward, 6-9 us each
// for < 0.1 us of arithmetic, all between dependent branch kernels) and output 0 a further elementwise launch.  Here one workgroup
// owns (output i, frame, band of 8 >> i output rows) -- the band is exactly one row of the coarsest 7x7 source in stage 4 -- and does
//   phase 1: T_j[C_i][pixels of the band at resolution j] = W_ij (BN folded, fp64 at load) . x_j  on the fp32 matrix cores, both
//            operands straight from global memory (everything is L2-resident: the sources are the branch outputs just written), ALL
//            loads of a workgroup in flight at once (one exposed latency), partial tiles to LDS;
//   phase 2: out[c][y][x] = act(x_i[c][y][x] + bias_i[c] + sum_j T_j[c][y >> (j-i)][x >> (j-i)]), 16-byte loads and stores.
// Outputs 1 .. nb-2 also add their finished stride-2 chains D_ij, j < i (plain convolutions launched earlier, on the streams of the
// branches they start from: grnet.cpp, hr_fuse_grouped), so every output but the last leaves this launch final, ReLU applied.
//
// MFMA operand maps (v_mfma_f32_16x16x4_f32: A[m = lane & 15][k = lane >> 4], B[k][n = lane & 15], D[4 (lane >> 4) + r][lane & 15]):
// M = pixels, N = output channels, K = input channels.  A lane loads LW = 4 / 2 / 1 CONSECUTIVE pixels of its channel with one load
// (bands of 112 / 28 / 7 pixels); component r of that load is the lane's A value of M-tile r, i.e. row m of tile (q, r) is pixel
// 16 LW q + LW m + r -- any bijection between rows and pixels is a valid GEMM, and the tiles leave through LDS by pixel index anyway.
// Waves split the 16-channel N tiles first and K with what is left of the four (C_i = 32: two N tiles x two K halves, reduced in phase 2).
#include "kernels.h"

namespace grk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

template <int LW> struct VecOf;
template <> struct VecOf<1> { typedef float type; };
template <> struct VecOf<2> { typedef f32x2 type; };
template <> struct VecOf<4> { typedef f32x4 type; };
__device__ __forceinline__ float comp(float v, int) { return v; }
__device__ __forceinline__ float comp(f32x2 v, int r) { return v[r]; }
__device__ __forceinline__ float comp(f32x4 v, int r) { return v[r]; }

template <int NB, int I>
struct OutGeom {
    static constexpr int CI = 32 << I, WI = 56 >> I, BR = 8 >> I, NSRC = NB - 1 - I, NT = CI / 16;
    static constexpr int NSPLIT = NT < 4 ? NT : 4, KS = 4 / NSPLIT, NPW = NT / NSPLIT;
    static constexpr int VW = WI % 4 == 0 ? 4 : 2, WV = WI / VW, NV = CI * BR * WV;
};
template <int NB, int I, int S>
struct SrcGeom {
    typedef OutGeom<NB, I> G;
    static constexpr int J = I + 1 + S, CJ = 32 << J, SH = S + 1, WJ = G::WI >> SH, ROWS = G::BR >> SH, PJ = ROWS * WJ, PJP = PJ | 1;
    static constexpr int LW = PJ > 32 ? 4 : (PJ > 16 ? 2 : 1), NLOAD = (PJ + 16 * LW - 1) / (16 * LW), MT = NLOAD * LW;
    static constexpr int KSW = CJ / 4 / G::KS;            // k-steps of one wave
    static_assert(ROWS >= 1 && KSW % 4 == 0, "band / K split geometry");
};
template <int NB, int I, int S> struct TOff { static constexpr int v = TOff<NB, I, S - 1>::v + OutGeom<NB, I>::CI * SrcGeom<NB, I, S - 1>::PJP; };
template <int NB, int I> struct TOff<NB, I, 0> { static constexpr int v = 0; };

template <int NB, int I, int S>
struct SrcRegs {
    typedef SrcGeom<NB, I, S> Q;
    typename VecOf<Q::LW>::type a[Q::KSW][Q::NLOAD];
    f32x4 b[Q::KSW / 4][OutGeom<NB, I>::NPW];
};

template <int NB, int I, int S>
__device__ __forceinline__ void src_load(SrcRegs<NB, I, S>& R, const FuseUpSrc& src, int n, int band, int lane, int nsel, int ksel) {
    typedef SrcGeom<NB, I, S> Q;
    typedef OutGeom<NB, I> G;
    typedef typename VecOf<Q::LW>::type vec;
    const int l15 = lane & 15, lq = lane >> 4;
    const float* xb = src.x + ((size_t)n * src.ctot + src.coff) * (Q::WJ * Q::WJ) + band * Q::PJ;
#pragma unroll
    for (int q = 0; q < Q::NLOAD; ++q) {
        const int pl = q * 16 * Q::LW + l15 * Q::LW;
        const float* xp = xb + (pl < Q::PJ ? pl : 0) + (size_t)(4 * ksel * Q::KSW + lq) * (Q::WJ * Q::WJ);     // rows past the band re-read pixel 0 and are never stored
#pragma unroll
        for (int ks = 0; ks < Q::KSW; ++ks) R.a[ks][q] = *reinterpret_cast<const vec*>(xp + (size_t)(4 * ks) * (Q::WJ * Q::WJ));
    }
#pragma unroll
    for (int g = 0; g < Q::KSW / 4; ++g)
#pragma unroll
        for (int u = 0; u < G::NPW; ++u)
            R.b[g][u] = *reinterpret_cast<const f32x4*>(src.w + ((size_t)((ksel * (Q::KSW / 4) + g) * G::NT + nsel * G::NPW + u) * 64 + lane) * 4);
}

template <int NB, int I, int S>
__device__ __forceinline__ void src_compute(const SrcRegs<NB, I, S>& R, float* T, int lane, int nsel, int ksel) {
    typedef SrcGeom<NB, I, S> Q;
    typedef OutGeom<NB, I> G;
    const int l15 = lane & 15, lq = lane >> 4;
    f32x4 acc[Q::MT][G::NPW];
#pragma unroll
    for (int t = 0; t < Q::MT; ++t)
#pragma unroll
        for (int u = 0; u < G::NPW; ++u) acc[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < Q::KSW; ++ks)
#pragma unroll
        for (int q = 0; q < Q::NLOAD; ++q)
#pragma unroll
            for (int r = 0; r < Q::LW; ++r)
#pragma unroll
                for (int u = 0; u < G::NPW; ++u)
                    acc[q * Q::LW + r][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(R.a[ks][q], r), R.b[ks >> 2][u][ks & 3], acc[q * Q::LW + r][u], 0, 0, 0);
    float* Ts = T + ksel * (TOff<NB, I, G::NSRC>::v) + TOff<NB, I, S>::v;
#pragma unroll
    for (int q = 0; q < Q::NLOAD; ++q)
#pragma unroll
        for (int r = 0; r < Q::LW; ++r)
#pragma unroll
            for (int u = 0; u < G::NPW; ++u) {
                const int c = (nsel * G::NPW + u) * 16 + l15;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int pix = q * 16 * Q::LW + (4 * lq + e) * Q::LW + r;
                    if (pix < Q::PJ) Ts[c * Q::PJP + pix] = acc[q * Q::LW + r][u][e];
                }
            }
}

template <int NB, int I, int S>
__device__ __forceinline__ float src_term(const float* T, int c, int yl, int x) {
    typedef SrcGeom<NB, I, S> Q;
    typedef OutGeom<NB, I> G;
    const float* Ts = T + TOff<NB, I, S>::v + c * Q::PJP + (yl >> Q::SH) * Q::WJ + (x >> Q::SH);
    float v = Ts[0];
    if constexpr (G::KS == 2) v += Ts[TOff<NB, I, G::NSRC>::v];
    return v;
}

template <int NB, int I>
__device__ __forceinline__ void fuse_up_body(const FuseUpOut& o, int n, int band, float* T) {
    typedef OutGeom<NB, I> G;
    static_assert(G::NSRC >= 1 && G::NSRC <= 3 && G::NV % 256 == 0, "geometry");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nsel = wave % G::NSPLIT, ksel = wave / G::NSPLIT;
    {
        SrcRegs<NB, I, 0> r0;
        src_load<NB, I, 0>(r0, o.src[0], n, band, lane, nsel, ksel);
        if constexpr (G::NSRC == 1) {
            src_compute<NB, I, 0>(r0, T, lane, nsel, ksel);
        } else {
            SrcRegs<NB, I, 1> r1;
            src_load<NB, I, 1>(r1, o.src[1], n, band, lane, nsel, ksel);
            if constexpr (G::NSRC == 2) {
                src_compute<NB, I, 0>(r0, T, lane, nsel, ksel);
                src_compute<NB, I, 1>(r1, T, lane, nsel, ksel);
            } else {
                SrcRegs<NB, I, 2> r2;
                src_load<NB, I, 2>(r2, o.src[2], n, band, lane, nsel, ksel);
                src_compute<NB, I, 0>(r0, T, lane, nsel, ksel);
                src_compute<NB, I, 1>(r1, T, lane, nsel, ksel);
                src_compute<NB, I, 2>(r2, T, lane, nsel, ksel);
            }
        }
    }
    typedef typename VecOf<G::VW>::type vec;
    // the identity term and the bias are requested before the barrier: their round trip overlaps the tiles' way through LDS
    vec basev[G::NV / 256];
    float biasv[G::NV / 256];
    const size_t plane = (size_t)G::WI * G::WI;
#pragma unroll
    for (int it = 0; it < G::NV / 256; ++it) {
        const int e = it * 256 + tid, c = e / (G::BR * G::WV), rem = e - c * (G::BR * G::WV), yl = rem / G::WV, xv = rem - yl * G::WV;
        const int pos = (band * G::BR + yl) * G::WI + xv * G::VW;
        basev[it] = *reinterpret_cast<const vec*>(o.base + ((size_t)n * o.base_ctot + o.base_coff + c) * plane + pos);
        if constexpr (I >= 1) {                          // the finished down chains D_ij (I of them): x_i + D_i0 + ..., in the order they are listed
            if (o.n_extra >= 1) basev[it] += *reinterpret_cast<const vec*>(o.extra[0] + ((size_t)n * o.extra_ctot[0] + o.extra_coff[0] + c) * plane + pos);
            if constexpr (I >= 2) { if (o.n_extra >= 2) basev[it] += *reinterpret_cast<const vec*>(o.extra[1] + ((size_t)n * o.extra_ctot[1] + o.extra_coff[1] + c) * plane + pos); }
        }
        biasv[it] = o.bias[c];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < G::NV / 256; ++it) {
        const int e = it * 256 + tid, c = e / (G::BR * G::WV), rem = e - c * (G::BR * G::WV), yl = rem / G::WV, xv = rem - yl * G::WV;
        vec y = basev[it];
#pragma unroll
        for (int r = 0; r < G::VW; ++r) {
            const int x = xv * G::VW + r;
            float v = y[r] + biasv[it];
            v += src_term<NB, I, 0>(T, c, yl, x);
            if constexpr (G::NSRC >= 2) v += src_term<NB, I, 1>(T, c, yl, x);
            if constexpr (G::NSRC >= 3) v += src_term<NB, I, 2>(T, c, yl, x);
            y[r] = o.relu ? fmaxf(v, 0.f) : v;
        }
        *reinterpret_cast<vec*>(o.out + ((size_t)n * o.out_ctot + o.out_coff + c) * plane + (band * G::BR + yl) * G::WI + xv * G::VW) = y;
    }
}

template <int NB, int I> constexpr int lds_floats() { return OutGeom<NB, I>::KS * TOff<NB, I, OutGeom<NB, I>::NSRC>::v; }
template <int NB> constexpr int lds_floats_max() {
    int m = lds_floats<NB, 0>();
    if constexpr (NB >= 3) m = lds_floats<NB, 1>() > m ? lds_floats<NB, 1>() : m;
    if constexpr (NB >= 4) m = lds_floats<NB, 2>() > m ? lds_floats<NB, 2>() : m;
    return m;
}

// grid: outputs 0 .. NB-2, each N frames x 7 bands
template <int NB>
__global__ __launch_bounds__(256) void hr_fuse_up_f32(const FuseUpArgs a) {
    __shared__ __align__(16) float T[lds_floats_max<NB>()];
    const int per = a.N * 7, i = blockIdx.x / per, rem = blockIdx.x - i * per, n = rem / 7, band = rem - n * 7;
    if (i == 0) fuse_up_body<NB, 0>(a.o[0], n, band, T);
    if constexpr (NB >= 3) { if (i == 1) fuse_up_body<NB, 1>(a.o[1], n, band, T); }
    if constexpr (NB >= 4) { if (i == 2) fuse_up_body<NB, 2>(a.o[2], n, band, T); }
}

}  // namespace

// folded 1x1 weights (cout, cin) fp64 -> [cin/16][cout/16][64 lanes][4 k-steps]: lane l's B values (channel cin = 16 g + 4 kk + (l >> 4),
// output channel 16 nt + (l & 15)) of four consecutive k-steps are one 16-byte load
void pack_fuse_up_weights(const double* w, int cout, int cin, float* out) {
    const int nt_n = cout / 16;
    for (int g = 0; g < cin / 16; ++g)
        for (int nt = 0; nt < nt_n; ++nt)
            for (int l = 0; l < 64; ++l)
                for (int kk = 0; kk < 4; ++kk)
                    out[(((size_t)g * nt_n + nt) * 64 + l) * 4 + kk] = (float)w[(size_t)(nt * 16 + (l & 15)) * cin + 16 * g + 4 * kk + (l >> 4)];
}

hipError_t launch_hr_fuse_up(const FuseUpArgs& a, hipStream_t s) {
    if (a.N < 1 || a.nb < 2 || a.nb > 4) return hipErrorInvalidValue;
    const dim3 grid(a.N * 7 * (a.nb - 1));
    if (a.nb == 2) return launch_k(hr_fuse_up_f32<2>, grid, dim3(256), 0, s, a);
    if (a.nb == 3) return launch_k(hr_fuse_up_f32<3>, grid, dim3(256), 0, s, a);
    return launch_k(hr_fuse_up_f32<4>, grid, dim3(256), 0, s, a);
}


// ================================================================================================================================================
// The same launch on the bf16 path (round 5): NHWC bf16 activations, bf16 matrix cores, fp32 sums.  Until now the bf16 path kept round 3's fuse layer
// (hr_fuse_separate: 31 1x1 launches of 8-10 us + 8 elementwise launches per forward at 256 frames, ~0.5 ms of 10.7).  Channels-last makes this kernel
// simpler than the fp32 one: a source pixel's 32 k-values of a chunk are 64 contiguous bytes, so the B operand of v_mfma_f32_16x16x32_bf16 (lane: pixel
// l & 15, k-group l >> 4) is ONE 16-byte global load, the A operand (W_ij packed [C_j / 32][C_i][32], BatchNorm folded) likewise, and phase 2 reads and
// writes whole 16-byte channel groups.  Workgroup = (output i, frame, band of 8 >> i rows), 4 waves:
//   phase 1: every (source j, 16-pixel tile of the band at resolution j, 16-channel block of C_i) is one MFMA column tile; the tiles are dealt to the
//            waves round-robin, a tile's K = C_j / 32 steps are loaded at once; T_j = bf16(W_ij x_j) -> LDS [pixel][C_i] (rounded where the separate
//            launches stored it);
//   phase 2: out = relu(x_i + sum_k D_ik + sum_j T_j[y >> (j-i)][x >> (j-i)] + bias_i), fp32, 16-byte loads / stores; x_i and the D's are requested
//            before the barrier.
namespace {

typedef __bf16 bf16x8_f __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4_f __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_f __attribute__((ext_vector_type(2)));
typedef unsigned short u16_f;
typedef __bf16 bf16x2_f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2_f(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2_f)); }
__device__ __forceinline__ void add8_f(float (&acc)[8], u32x4_f v) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { acc[2 * k] += __uint_as_float(v[k] << 16); acc[2 * k + 1] += __uint_as_float(v[k] & 0xffff0000u); }
}

template <int NB, int I>
struct OutGeomB {
    static constexpr int CI = 32 << I, WI = 56 >> I, BR = 8 >> I, NSRC = NB - 1 - I, NCB = CI / 16, UPP = CI / 8;
    static constexpr int NU = BR * WI * UPP, NIT = (NU + 255) / 256;
};
template <int NB, int I, int S>
struct SrcGeomB {
    typedef OutGeomB<NB, I> G;
    static constexpr int J = I + 1 + S, CJ = 32 << J, SH = S + 1, WJ = G::WI >> SH, ROWS = G::BR >> SH, PJ = ROWS * WJ, NPT = (PJ + 15) / 16, NK = CJ / 32;
    static constexpr int TILES = NPT * G::NCB;
    static_assert(ROWS >= 1, "band geometry");
};
template <int NB, int I, int S> struct TOffB { static constexpr int v = TOffB<NB, I, S - 1>::v + SrcGeomB<NB, I, S - 1>::PJ * OutGeomB<NB, I>::CI; };      // bf16 elements
template <int NB, int I> struct TOffB<NB, I, 0> { static constexpr int v = 0; };

// tiles t0, t0 + 4, ... of source S for this wave (t0 = the wave's first tile of this source, continuing the round-robin over all sources)
template <int NB, int I, int S>
__device__ __forceinline__ void src_tiles_bf16(const FuseUpSrc& src, int n, int band, int lane, int first, u16_f* T) {
    typedef SrcGeomB<NB, I, S> Q;
    typedef OutGeomB<NB, I> G;
    const int l15 = lane & 15, lq = lane >> 4;
    const u16_f* xb = reinterpret_cast<const u16_f*>(src.x) + (size_t)n * Q::WJ * Q::WJ * src.ctot + src.coff + (size_t)band * Q::PJ * src.ctot;
    const u16_f* wb = reinterpret_cast<const u16_f*>(src.w);
    u16_f* Ts = T + TOffB<NB, I, S>::v;
#pragma unroll
    for (int t = 0; t < (Q::TILES + 3) / 4; ++t) {
        const int tile = first + 4 * t;
        if (tile >= Q::TILES) break;                           // wave-uniform
        const int pt = tile / G::NCB, cb = tile - pt * G::NCB, px = pt * 16 + l15;
        const u16_f* xp = xb + (size_t)(px < Q::PJ ? px : 0) * src.ctot + lq * 8;      // columns past the band re-read pixel 0 and are never stored
        const u16_f* wp = wb + ((size_t)cb * 16 + l15) * 32 + lq * 8;
        bf16x8_f bq[Q::NK], aq[Q::NK];
#pragma unroll
        for (int k = 0; k < Q::NK; ++k) {
            bq[k] = *reinterpret_cast<const bf16x8_f*>(xp + k * 32);
            aq[k] = *reinterpret_cast<const bf16x8_f*>(wp + (size_t)k * G::CI * 32);
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < Q::NK; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[k], bq[k], acc, 0, 0, 0);
        if (px < Q::PJ) *reinterpret_cast<u32x2_f*>(Ts + px * G::CI + cb * 16 + lq * 4) = u32x2_f{pack2_f(acc[0], acc[1]), pack2_f(acc[2], acc[3])};
    }
}

template <int NB, int I>
__device__ __forceinline__ void fuse_up_body_bf16(const FuseUpOut& o, int n, int band, u16_f* T) {
    typedef OutGeomB<NB, I> G;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the waves take the tiles of all sources round-robin: tile g of the concatenated list goes to wave g % 4
    {
        typedef SrcGeomB<NB, I, 0> Q0;
        src_tiles_bf16<NB, I, 0>(o.src[0], n, band, lane, wave, T);
        if constexpr (G::NSRC >= 2) {
            typedef SrcGeomB<NB, I, 1> Q1;
            constexpr int base1 = Q0::TILES;
            src_tiles_bf16<NB, I, 1>(o.src[1], n, band, lane, (wave + 4 - base1 % 4) % 4, T);
            if constexpr (G::NSRC >= 3) {
                constexpr int base2 = base1 + Q1::TILES;
                src_tiles_bf16<NB, I, 2>(o.src[2], n, band, lane, (wave + 4 - base2 % 4) % 4, T);
            }
        }
    }
    // phase 2 operands that do not depend on phase 1: requested before the barrier
    const u16_f* xb = reinterpret_cast<const u16_f*>(o.base) + (size_t)n * G::WI * G::WI * o.base_ctot + o.base_coff;
    u32x4_f bv[G::NIT], e0[G::NIT], e1[G::NIT];
#pragma unroll
    for (int it = 0; it < G::NIT; ++it) {
        const int u = it * 256 + tid, px = u / G::UPP, part = u - px * G::UPP, yl = px / G::WI, x = px - yl * G::WI;
        const size_t pix = (size_t)(band * G::BR + yl) * G::WI + x;
        bv[it] = e0[it] = e1[it] = u32x4_f{0u, 0u, 0u, 0u};
        if (u < G::NU) {
            bv[it] = *reinterpret_cast<const u32x4_f*>(xb + pix * o.base_ctot + part * 8);
            if constexpr (I >= 1) {
                if (o.n_extra >= 1) e0[it] = *reinterpret_cast<const u32x4_f*>(reinterpret_cast<const u16_f*>(o.extra[0]) + ((size_t)n * G::WI * G::WI + pix) * o.extra_ctot[0] + o.extra_coff[0] + part * 8);
                if constexpr (I >= 2) { if (o.n_extra >= 2) e1[it] = *reinterpret_cast<const u32x4_f*>(reinterpret_cast<const u16_f*>(o.extra[1]) + ((size_t)n * G::WI * G::WI + pix) * o.extra_ctot[1] + o.extra_coff[1] + part * 8); }
            }
        }
    }
    __syncthreads();
    u16_f* ob = reinterpret_cast<u16_f*>(o.out) + (size_t)n * G::WI * G::WI * o.out_ctot + o.out_coff;
#pragma unroll
    for (int it = 0; it < G::NIT; ++it) {
        const int u = it * 256 + tid, px = u / G::UPP, part = u - px * G::UPP, yl = px / G::WI, x = px - yl * G::WI;
        if (u >= G::NU) continue;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        add8_f(acc, bv[it]);
        if constexpr (I >= 1) { add8_f(acc, e0[it]); if constexpr (I >= 2) add8_f(acc, e1[it]); }
        {
            typedef SrcGeomB<NB, I, 0> Q;
            add8_f(acc, *reinterpret_cast<const u32x4_f*>(T + TOffB<NB, I, 0>::v + ((yl >> Q::SH) * Q::WJ + (x >> Q::SH)) * G::CI + part * 8));
        }
        if constexpr (G::NSRC >= 2) {
            typedef SrcGeomB<NB, I, 1> Q;
            add8_f(acc, *reinterpret_cast<const u32x4_f*>(T + TOffB<NB, I, 1>::v + ((yl >> Q::SH) * Q::WJ + (x >> Q::SH)) * G::CI + part * 8));
        }
        if constexpr (G::NSRC >= 3) {
            typedef SrcGeomB<NB, I, 2> Q;
            add8_f(acc, *reinterpret_cast<const u32x4_f*>(T + TOffB<NB, I, 2>::v + ((yl >> Q::SH) * Q::WJ + (x >> Q::SH)) * G::CI + part * 8));
        }
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(o.bias + part * 8), b1 = *reinterpret_cast<const f32x4*>(o.bias + part * 8 + 4);
        float v[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = acc[k] + b0[k]; v[4 + k] = acc[4 + k] + b1[k]; }
        if (o.relu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        *reinterpret_cast<u32x4_f*>(ob + ((size_t)(band * G::BR + yl) * G::WI + x) * o.out_ctot + part * 8) =
            u32x4_f{pack2_f(v[0], v[1]), pack2_f(v[2], v[3]), pack2_f(v[4], v[5]), pack2_f(v[6], v[7])};
    }
}

template <int NB, int I> constexpr int lds_elems_b() { return TOffB<NB, I, OutGeomB<NB, I>::NSRC>::v; }
template <int NB> constexpr int lds_elems_b_max() {
    int m = lds_elems_b<NB, 0>();
    if constexpr (NB >= 3) m = lds_elems_b<NB, 1>() > m ? lds_elems_b<NB, 1>() : m;
    if constexpr (NB >= 4) m = lds_elems_b<NB, 2>() > m ? lds_elems_b<NB, 2>() : m;
    return (m + 7) & ~7;
}

template <int NB>
__global__ __launch_bounds__(256) void hr_fuse_up_bf16(const FuseUpArgs a) {
    __shared__ __align__(16) u16_f T[lds_elems_b_max<NB>()];
    const int per = a.N * 7, i = a.only >= 0 ? a.only : blockIdx.x / per, rem = a.only >= 0 ? blockIdx.x : blockIdx.x - i * per, n = rem / 7, band = rem - n * 7;
    if (i == 0) fuse_up_body_bf16<NB, 0>(a.o[0], n, band, T);
    if constexpr (NB >= 3) { if (i == 1) fuse_up_body_bf16<NB, 1>(a.o[1], n, band, T); }
    if constexpr (NB >= 4) { if (i == 2) fuse_up_body_bf16<NB, 2>(a.o[2], n, band, T); }
}

}  // namespace

// folded 1x1 weights (cout, cin) fp64 -> [cin / 32][cout][32] bf16 (round to nearest even): the layout of every bf16 convolution (grnet.cpp: pack_conv)
void pack_fuse_up_weights_bf16(const double* w, int cout, int cin, unsigned short* out) {
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            float f = (float)w[(size_t)co * cin + ci];
            unsigned u;
            __builtin_memcpy(&u, &f, 4);
            u += 0x7fffu + ((u >> 16) & 1u);
            out[((size_t)(ci / 32) * cout + co) * 32 + ci % 32] = (unsigned short)(u >> 16);
        }
}

hipError_t launch_hr_fuse_up_bf16(const FuseUpArgs& a, hipStream_t s) {
    if (a.N < 1 || a.nb < 2 || a.nb > 4) return hipErrorInvalidValue;
    if (a.only >= a.nb - 1) return hipErrorInvalidValue;
    const dim3 grid(a.N * 7 * (a.only >= 0 ? 1 : a.nb - 1));
    if (a.nb == 2) return launch_k(hr_fuse_up_bf16<2>, grid, dim3(256), 0, s, a);
    if (a.nb == 3) return launch_k(hr_fuse_up_bf16<3>, grid, dim3(256), 0, s, a);
    return launch_k(hr_fuse_up_bf16<4>, grid, dim3(256), 0, s, a);
}

}  // namespace grk
