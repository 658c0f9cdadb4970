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

}  // namespace grk
