// Winograd F(4x4, 3x3) for the 3x3 stride-1 convolutions on the SMALL maps of the path: the 128-channel 14x14 and the 256-channel 7x7
// HR branches (hrnet.py:43-57 as instantiated by hrnet.py:141-187: 80 launches per forward) and the 256 -> 256 @14x14 layer of the
// last upsample head (hrnet.py:444-451).  Until round 3 these ran as direct implicit GEMMs (conv_splitk_f32) at 0.3-0.4 of the fp32
// matrix peak; a kernel trace of the HR section (profiles/r03_hr_section_timeline.txt) shows its four concurrent branch kernels
// together keeping the FP32 pipe (matrix + vector instructions share it on gfx950) busy all the time, and the two small-map branches
// asking for 80 % of that time -- 9 multiplies per output where F(4x4,3x3) needs 2.25 (x 1.31 for padding 14 -> 16 / 7 -> 8).
//
// Register-resident: nothing of the main loop goes through the LDS and there is no workgroup barrier in it.  A 14x14 map is 4x4 tiles of 4x4 outputs (16 tiles = the 16 rows of ONE MFMA
// row tile = one image), a 7x7 map 2x2 tiles (an MFMA row tile = 4 images).  A wave owns (row tile, 16 output channels, all 36
// points, the k-steps of its K slice):
//   * lane (tile, channel k of the k-step) loads its own 6 patch rows straight into registers (rows / images outside are out-of-range
//     buffer offsets = zeros; the columns past the map's right edge are an out-of-range second half on 14-wide maps, one select on
//     7-wide ones), column pass on its own 4 columns, halo columns TRANSFORMED from the x-neighbour tiles by DPP row shifts of 4 / 8
//     lanes (tiles are laid out x-major in the 16 lanes so that a shift past the row's end IS the zero padding), row pass, 36 MFMAs;
//   * B fragments: one 8-byte load per point and pair of k-steps, packed per (channel block, k-pair, point) as 64 lanes x 2 floats
//     (pack_wino4r_weights), re-requested for the next pair right behind the point's last MFMA;
//   * KS waves of a workgroup split the input channels (K = 128 / 256 needs it: 8 k-steps per wave).  Every wave inverse-transforms
//     its partial sums in registers (the transform is linear) and writes them to ITS slot of an LDS tile laid out like the output
//     (16 channels x H x W are one contiguous run of the NCHW tensor per image); after one barrier the waves share the read-out:
//     slot 0 + slot 1 + ... in that order (deterministic), + residual (requested before the inverse transform), ReLU, 16-byte stores.
#include "kernels.h"

#include <type_traits>

namespace grk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

// B^T of F(4,3), rows 0..2 / 3..5 (Lavin & Gray; as in conv_wino4.hip)
__device__ __forceinline__ void bt_lo(const float* d, float& r0, float& r1, float& r2) {
    const float t1 = fmaf(-4.f, d[2], d[4]), t2 = fmaf(-4.f, d[1], d[3]);
    r0 = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
    r1 = t1 + t2;
    r2 = t1 - t2;
}
__device__ __forceinline__ void bt_hi(const float* d, float& r3, float& r4, float& r5) {
    const float u1 = d[4] - d[2], u2 = 2.f * (d[3] - d[1]);
    r3 = u1 + u2;
    r4 = u1 - u2;
    r5 = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
}

template <int WD, int KS>
struct GeoS {
    static constexpr int IPW = WD == 14 ? 1 : 4;        // images per MFMA row tile
    static constexpr int HW = WD * WD;
    static constexpr int SLAB = 16 * HW;                // floats of one image's 16-channel output slab (3136 / 784): contiguous in NCHW
    static constexpr int SLABP = SLAB + 4;              // in LDS
    static constexpr int OSLOT = IPW * SLABP;           // floats per wave slot
    static constexpr int UNITS = IPW * SLAB / 4;        // 16-byte units of a workgroup's output: 784
    static constexpr int ITERS = (UNITS + 64 * KS - 1) / (64 * KS);
    static constexpr size_t lds_bytes = sizeof(float) * KS * OSLOT;
};

template <int WD, int C, int KS>
__global__ __launch_bounds__(64 * KS) void conv_wino4s_f32(const ConvArgs a) {
    typedef GeoS<WD, KS> G;
    constexpr int HW = G::HW, IPW = G::IPW, NBK = C / 16, NKP = C / 8, NK = C / 4 / KS, SH = WD == 14 ? 4 : 8;
    static_assert(NK % 2 == 0 && NBK % 8 == 0, "k-steps come in pairs; channel blocks are dealt to the 8 XCDs");
    constexpr int kOOB = 0x7fffffff;
    extern __shared__ __align__(16) float O[];           // [KS][IPW][SLABP]
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
    const int kw = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // workgroup -> (row tile, channel block); blocks b and b + 8 share an XCD (round-robin dispatch): a channel block's weights stay in
    // ONE XCD's L2 for all row tiles (speed only)
    const int id = blockIdx.x, jd = id >> 3;
    const int nb = (id & 7) + 8 * (jd % (NBK / 8)), grp = jd / (NBK / 8);
    const int img0 = grp * IPW;
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio >= 2) __builtin_amdgcn_s_setprio(3);

    // tile of this lane as A-operand row: x-major, so that the x neighbours are SH lanes away inside the 16-lane row
    const int tx = WD == 14 ? (l15 >> 2) : (l15 >> 3), ty = WD == 14 ? (l15 & 3) : (l15 & 1), ii = WD == 14 ? 0 : ((l15 >> 1) & 3);
    const bool img_ok = img0 + ii < a.N;
    const float* inb = a.in + ((size_t)img0 * a.in_ctot + a.in_coff) * HW;
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, (((IPW - 1) * a.in_ctot + C) * HW + 4) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, 36 * C * C * 4, 0x00020000);
    int voff[6], voff2[WD == 14 ? 6 : 1];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = 4 * ty - 1 + i;
        const bool ok = img_ok && row >= 0 && row < WD;
        const int off = ((ii * a.in_ctot + lq) * HW + row * WD + 4 * tx) * 4;
        voff[i] = ok ? off : kOOB;
        if constexpr (WD == 14) voff2[i] = ok && tx < 3 ? off + 8 : kOOB;      // columns 4tx+2, 4tx+3 exist for tx <= 2
    }
    const int ks0 = kw * NK;
    f32x4 pd[6];
    auto load_patch = [&](int ks) {
        const int soff = (ks0 + ks) * (4 * HW * 4);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if constexpr (WD == 14) {
                const f32x2 lo = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_rsrc, voff[i], soff, 0));
                const f32x2 hi = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r_rsrc, voff2[i], soff, 0));
                pd[i] = f32x4{lo[0], lo[1], hi[0], hi[1]};
            } else {
                pd[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, voff[i], soff, 0));
            }
        }
    };
    const int ub = lane * 8;
    f32x2 bq[36];
    const int ubase = (nb * NKP + (ks0 >> 1)) * (36 * 512);
    auto load_b = [&](int kp, int p) {
        bq[p] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(u_rsrc, ub, ubase + (kp * 36 + p) * 512, 0));
    };

    f32x4 acc[36];
#pragma unroll
    for (int p = 0; p < 36; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    load_patch(0);
#pragma unroll
    for (int p = 0; p < 36; ++p) load_b(0, p);

    const bool last_col = WD == 7 && tx == 1;            // 7-wide maps: column 7 does not exist
    auto kstep = [&](int ks, auto sel_c, auto reload_c, auto last_c, int reload_kp) {
        constexpr int sel = decltype(sel_c)::value;
        constexpr bool RELOAD = decltype(reload_c)::value, LAST = decltype(last_c)::value;
        float e[6][6];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float col[6] = {pd[0][c], pd[1][c], pd[2][c], pd[3][c], pd[4][c], pd[5][c]};
            if (WD == 7 && c == 3) {
#pragma unroll
                for (int i = 0; i < 6; ++i) col[i] = last_col ? 0.f : col[i];
            }
            bt_lo(col, e[0][c + 1], e[1][c + 1], e[2][c + 1]);
            bt_hi(col, e[3][c + 1], e[4][c + 1], e[5][c + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LAST) load_patch(ks + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {                    // halo columns, transformed, from the tiles tx - 1 / tx + 1; past the row's end: zero
            e[i][0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e[i][4]), 0x110 + SH, 0xf, 0xf, true));   // row_shr:SH
            e[i][5] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e[i][1]), 0x100 + SH, 0xf, 0xf, true));   // row_shl:SH
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float v[6];
            bt_lo(e[i], v[0], v[1], v[2]);
            bt_hi(e[i], v[3], v[4], v[5]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 6; ++j)
                acc[i * 6 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j], bq[i * 6 + j][sel], acc[i * 6 + j], 0, 0, 0);
            if constexpr (RELOAD) {
#pragma unroll
                for (int j = 0; j < 6; ++j) load_b(reload_kp, i * 6 + j);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using std::integral_constant;
    constexpr integral_constant<int, 0> c0{};
    constexpr integral_constant<int, 1> c1{};
    constexpr integral_constant<bool, true> yes{};
    constexpr integral_constant<bool, false> no{};
#pragma unroll 1
    for (int kp = 0; kp + 1 < NK / 2; ++kp) {
        kstep(2 * kp, c0, no, no, 0);
        kstep(2 * kp + 1, c1, yes, no, kp + 1);
    }
    kstep(NK - 2, c0, no, no, 0);
    kstep(NK - 1, c1, no, yes, 0);

    // ---- epilogue.  The residual of this wave's share of the read-out is requested now.
    const size_t slab0 = ((size_t)img0 * a.out_ctot + a.out_coff + nb * 16) * HW;          // first float of image img0's slab
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + slab0), (short)0, ((IPW - 1) * a.out_ctot * HW + G::SLAB) * 4, 0x00020000);
    const bool has_add = a.n_add == 1;
    int uoff[G::ITERS], goff[G::ITERS];                  // LDS float offset inside a slot / global byte offset from image img0's slab
    f32x4 res[G::ITERS];
#pragma unroll
    for (int it = 0; it < G::ITERS; ++it) {
        const int u = it * (64 * KS) + kw * 64 + lane;
        const int iu = WD == 14 ? 0 : u / (G::SLAB / 4), w = u - iu * (G::SLAB / 4);
        const bool ok = u < G::UNITS && img0 + iu < a.N;
        uoff[it] = ok ? iu * G::SLABP + w * 4 : 0;
        goff[it] = ok ? (iu * a.out_ctot * HW + w * 4) * 4 : kOOB;
        res[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (has_add) {
        const size_t aslab0 = ((size_t)img0 * a.add_ctot[0] + a.add_coff[0] + nb * 16) * HW;
        const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.add[0] + aslab0), (short)0, ((IPW - 1) * a.add_ctot[0] * HW + G::SLAB) * 4, 0x00020000);
#pragma unroll
        for (int it = 0; it < G::ITERS; ++it) {
            const int u = it * (64 * KS) + kw * 64 + lane;
            const int iu = WD == 14 ? 0 : u / (G::SLAB / 4), w = u - iu * (G::SLAB / 4);
            const int ao = u < G::UNITS && img0 + iu < a.N ? (iu * a.add_ctot[0] * HW + w * 4) * 4 : kOOB;
            res[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, ao, 0, 0));
        }
    }
    // inverse transform A^T M A of the 4 tiles of this lane (D layout: tile slots 4*lq .. 4*lq+3, output channel l15), partial over the K slice
    const float bias = kw == 0 ? a.bias[nb * 16 + l15] : 0.f;
    float* Ow = O + kw * G::OSLOT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float s[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float m0 = acc[j][i], m1 = acc[6 + j][i], m2 = acc[12 + j][i], m3 = acc[18 + j][i], m4 = acc[24 + j][i], m5 = acc[30 + j][i];
            const float p12 = m1 + m2, m12 = m1 - m2, p34 = m3 + m4, m34 = m3 - m4;
            s[0][j] = m0 + p12 + p34;
            s[1][j] = fmaf(2.f, m34, m12);
            s[2][j] = fmaf(4.f, p34, p12);
            s[3][j] = fmaf(8.f, m34, m12) + m5;
        }
        const int slot = 4 * lq + i;
        const int stx = WD == 14 ? (slot >> 2) : (slot >> 3), sty = WD == 14 ? (slot & 3) : (slot & 1), sii = WD == 14 ? 0 : ((slot >> 1) & 3);
        float* op = Ow + sii * G::SLABP + l15 * HW + (4 * sty) * WD + 4 * stx;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* q = s[r];
            const float p12 = q[1] + q[2], m12 = q[1] - q[2], p34 = q[3] + q[4], m34 = q[3] - q[4];
            const float y0 = q[0] + p12 + p34 + bias, y1 = fmaf(2.f, m34, m12) + bias, y2 = fmaf(4.f, p34, p12) + bias, y3 = fmaf(8.f, m34, m12) + q[5] + bias;
            if (4 * sty + r < WD) {
                if constexpr (WD == 14) {
                    *reinterpret_cast<f32x2*>(op + r * WD) = f32x2{y0, y1};
                    if (stx < 3) *reinterpret_cast<f32x2*>(op + r * WD + 2) = f32x2{y2, y3};
                } else {
                    op[r * WD] = y0; op[r * WD + 1] = y1; op[r * WD + 2] = y2;
                    if (stx == 0) op[r * WD + 3] = y3;
                }
            }
        }
    }
    __syncthreads();
    // read-out: the KS waves share the workgroup's 784 units; partial sums added in slot order
#pragma unroll
    for (int it = 0; it < G::ITERS; ++it) {
        f32x4 v = *reinterpret_cast<const f32x4*>(O + uoff[it]);
#pragma unroll
        for (int k = 1; k < KS; ++k) v += *reinterpret_cast<const f32x4*>(O + k * G::OSLOT + uoff[it]);
        v += res[it];
        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), o_rsrc, goff[it], 0, 0);
    }
}

template <int WD, int C, int KS>
hipError_t launch_s(const ConvArgs& a, hipStream_t s) {
    typedef GeoS<WD, KS> G;
    static PerDeviceOnce attr;                           // one per instantiation
    int dev = 0;
    if (hipError_t e = current_device(&dev); e != hipSuccess) return e;
    if (hipError_t e = once_per_device(attr, dev, [](int*) {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino4s_f32<WD, C, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::lds_bytes);
        }); e != hipSuccess) return e;
    const int groups = (a.N + G::IPW - 1) / G::IPW;
    return launch_k(conv_wino4s_f32<WD, C, KS>, dim3(groups * (C / 16)), dim3(64 * KS), G::lds_bytes, s, a);
}

}  // namespace

bool conv_wino4s_eligible(int cin, int cout, int ks, int stride, int h, int w, int n_add) {
    return ks == 3 && stride == 1 && cin == cout && n_add <= 1 && ((h == 14 && w == 14 && (cin == 128 || cin == 256)) || (h == 7 && w == 7 && cin == 256));
}

// a.w: pack_wino4r_weights; ksplit: waves per workgroup splitting the input channels (0: the default of the shape)
hipError_t launch_conv_wino4s(ConvArgs a, hipStream_t s, int ksplit) {
    if (!conv_wino4s_eligible(a.Cin, a.Cout, a.ks, a.stride, a.H, a.W, a.n_add) || a.N < 1) return hipErrorInvalidValue;
    if (a.n_add == 1 && a.add_shift[0] != 0) return hipErrorInvalidValue;
    // more than 4 waves per workgroup would need two waves per SIMD, i.e. <= 256 registers per wave: the kernel holds 144 accumulation
    // registers + 72 of B fragments + the patch in flight
    if (a.W == 14 && a.Cin == 128) return ksplit == 2 ? launch_s<14, 128, 2>(a, s) : launch_s<14, 128, 4>(a, s);
    if (a.W == 14) return launch_s<14, 256, 4>(a, s);
    return ksplit == 2 ? launch_s<7, 256, 2>(a, s) : launch_s<7, 256, 4>(a, s);
}

// U = G g G^T per (cout, cin) in fp64 -> [cout/16][cin/8][36][lane = (cin%4)*16 + cout%16][k-step of the pair]; w: (cout, cin, 3, 3) folded weights
void pack_wino4r_weights(const double* w, int cout, int cin, float* out) {
    const int nkp = cin / 8;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            double u[36];
            wino4_transform_filter(w + ((size_t)co * cin + ci) * 9, u);
            const int nb = co / 16, l15 = co % 16, ks = ci / 4, lq = ci % 4;
            for (int p = 0; p < 36; ++p)
                out[((((size_t)nb * nkp + ks / 2) * 36 + p) * 64 + lq * 16 + l15) * 2 + (ks & 1)] = (float)u[p];
        }
}

}  // namespace grk
