// Winograd F(4x4, 3x3) for the 3x3 stride-1 convolutions of the two narrow HR branches (hrnet.py:43-57 as instantiated by
// hrnet.py:141-187: 32 -> 32 on 56x56 maps, 64 -> 64 on 28x28 maps; 129 of the path's launches), register-resident:
// nothing of the main loop goes through the LDS and there is no workgroup barrier in it.
//
// Why another kernel for these layers.  conv_wino4.hip splits the 36 transform points of a tile row over the 4 waves of a workgroup,
// so the transformed input V and the accumulators M both cross waves through the LDS (2 barriers + 2 LDS round trips per 8 channels
// / per 16 output channels).  On a 32- or 64-channel layer that is most of the time: phase stamps of the same loop inside the fused
// BasicBlock kernel (profiles/r03_fused_block_phases.txt) show a chunk taking 2 400 cycles for 1 152 cycles of MFMAs, and a launch
// spends about as long in prologue + epilogue as in the loop.  Here a WAVE owns a whole MFMA row tile (14 tiles: one tile row of a
// 56-wide map or two of a 28-wide one) x ALL 36 points x 16 output channels (x the k-steps of its K slice):
//   * lane (tile t, channel k of the k-step) loads its own 6 patch rows straight from HBM/L2 into registers (16 bytes per row; rows
//     and tiles outside the image are out-of-range buffer offsets, which read as zero = the convolution's padding), transforms the
//     patch B^T d B in registers -- column pass on its own 4 columns, the two halo columns come TRANSFORMED from the neighbour lanes
//     by DPP -- and the 36 results ARE its A operands of the 36 MFMAs of that k-step (A[row = tile][k]);
//   * B fragments: one 8-byte load per point and pair of k-steps, packed per (channel block, k-pair, point) as 64 lanes x 2 floats
//     (pack_wino4r_weights), double-buffered two pairs ahead;
//   * the accumulators of all 36 points of a (tile, channel) sit in ONE lane, so the inverse transform A^T M A is lane-local too;
//     the 4x4 outputs go through a 14 KB LDS tile only to leave as whole 896-byte runs per channel (+ bias, + residual, ReLU).
// KS > 1: the workgroup's KS waves split the input channels; partial OUTPUTS (the inverse transform is linear) are added through the
// same LDS tile in a fixed order, so results do not depend on timing.
#include "kernels.h"

#include <type_traits>

namespace grk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

#ifdef GRNET_ABLATION
__device__ unsigned long long g_phase_w4r[8];   // [0] start -> first k-step entered, [1] k loop, [2] epilogue, [3] waves, [4] column pass + patch issue (incl. the wait for the patch), [5] halo + row pass + MFMAs
#define GRK_W4R_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#define GRK_W4R_PHASE(i, t0, t1) do { if (lane == 0) atomicAdd(&g_phase_w4r[i], (t1) - (t0)); } while (0)
#define GRK_W4R_ACC(var, t0, t1) var += (t1) - (t0)
#else
#define GRK_W4R_ACC(var, t0, t1) do {} while (0)
#define GRK_W4R_STAMP(var) do {} while (0)
#define GRK_W4R_PHASE(i, t0, t1) do {} while (0)
#endif
constexpr int kOStride = 228;                // floats per channel of the output tile in LDS (224 + 4: the 16 channels of a store hit distinct banks)

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

template <int WD, int C, int KS>
__global__ __launch_bounds__(64 * KS) void conv_wino4r_f32(const ConvArgs a) {
    constexpr int H = WD, HW = WD * WD, TPR = WD / 4, TRG = WD == 56 ? 1 : 2, GROUPS = WD == 56 ? 14 : 4;
    constexpr int NBK = C / 16, NKP = C / 8, NK = C / 4 / KS;      // channel blocks, k-pairs of the layer, k-steps of a wave
    static_assert(NK % 2 == 0, "the k loop walks pairs of k-steps");
    constexpr int kOOB = 0x7fffffff;
    __shared__ __align__(16) float O[16 * kOStride];
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
    const int kw = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // wave-uniform: it goes into scalar load offsets
    GRK_W4R_STAMP(tk0);
#ifdef GRNET_ABLATION
    unsigned long long acc_t4 = 0, acc_t5 = 0;
#endif

    int bx, nb;
    {
        const int id = blockIdx.x;
        if (a.xcd) {                                     // an XCD owns a contiguous range of (image, row tile), all channel blocks of it
            const int x = id & 7, j = id >> 3;
            bx = x * (a.gx >> 3) + j / NBK;
            nb = j - (j / NBK) * NBK;
        } else {
            bx = id / NBK;
            nb = id - bx * NBK;
        }
    }
    const int img = bx / GROUPS, g = bx - img * GROUPS;
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio >= 2) __builtin_amdgcn_s_setprio(3);
    const float* inb = a.in + ((size_t)img * a.in_ctot + a.in_coff) * HW;
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, C * HW * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, 36 * C * C * 4, 0x00020000);

    // ---- A side: lane (tile l15, channel lq of the k-step)
    const int tx = WD == 56 ? l15 : (l15 & 7), trl = WD == 56 ? 0 : (l15 >> 3), tr = g * TRG + trl;
    const bool real = tx < TPR && tr < H / 4;
    int voff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int row = 4 * tr - 1 + i;
        voff[i] = real && row >= 0 && row < H ? (lq * HW + row * WD + 4 * tx) * 4 : kOOB;
    }
    const int ks0 = kw * NK;                             // first k-step of this wave
    f32x4 pd[6];
    auto load_patch = [&](int ks) {
        const int soff = (ks0 + ks) * (4 * HW * 4);
#pragma unroll
        for (int i = 0; i < 6; ++i) pd[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_rsrc, voff[i], soff, 0));
    };
    // ---- B side: lane (k = lq, n = l15): [channel block][k-pair][point][lane][2 k-steps]
    const int ub = lane * 8;
    f32x2 bq[36];                                        // one k-pair of B fragments; each point's is re-requested for the next pair behind its last MFMA
    const int ubase = (nb * NKP + (ks0 >> 1)) * (36 * 512);
    auto load_b = [&](int kp, int p) {                   // kp: k-pair of this wave
        bq[p] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(u_rsrc, ub, ubase + (kp * 36 + p) * 512, 0));
    };

    f32x4 acc[36];
#pragma unroll
    for (int p = 0; p < 36; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_patch(0);
#pragma unroll
    for (int p = 0; p < 36; ++p) load_b(0, p);

    // sel: k-step of the pair; RELOAD: re-request every point's fragments of k-pair reload_kp behind its MFMA; LAST: no further patch
    auto kstep = [&](int ks, auto sel_c, auto reload_c, auto last_c, int reload_kp) {
        constexpr int sel = decltype(sel_c)::value;
        constexpr bool RELOAD = decltype(reload_c)::value, LAST = decltype(last_c)::value;
        float e[6][6];
        GRK_W4R_STAMP(ta);
        // column pass (B^T over the 6 rows) on the lane's own 4 columns
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float col[6] = {pd[0][c], pd[1][c], pd[2][c], pd[3][c], pd[4][c], pd[5][c]};
            bt_lo(col, e[0][c + 1], e[1][c + 1], e[2][c + 1]);
            bt_hi(col, e[3][c + 1], e[4][c + 1], e[5][c + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LAST) load_patch(ks + 1);         // pd is free again: the next k-step's rows travel under this one's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        GRK_W4R_STAMP(tb);
        GRK_W4R_ACC(acc_t4, ta, tb);
        // halo columns 4t-1 / 4t+4, already transformed, from the neighbour lanes (an idle lane / the row's end supplies the zero padding)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            e[i][0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e[i][4]), 0x111, 0xf, 0xf, true));   // row_shr:1
            e[i][5] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(e[i][1]), 0x101, 0xf, 0xf, true));   // row_shl:1
        }
        // row pass ((B^T d) B) and the k-step's 36 MFMAs: A = the transformed patch, B = this wave's 16 output channels
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
        GRK_W4R_STAMP(tc);
        GRK_W4R_ACC(acc_t5, tb, tc);
    };
    using std::integral_constant;
    constexpr integral_constant<int, 0> c0{};
    constexpr integral_constant<int, 1> c1{};
    constexpr integral_constant<bool, true> yes{};
    constexpr integral_constant<bool, false> no{};
    GRK_W4R_STAMP(tk1);
#pragma unroll 1
    for (int kp = 0; kp + 1 < NK / 2; ++kp) {
        kstep(2 * kp, c0, no, no, 0);
        kstep(2 * kp + 1, c1, yes, no, kp + 1);
    }
    kstep(NK - 2, c0, no, no, 0);
    kstep(NK - 1, c1, no, yes, 0);
    GRK_W4R_STAMP(tk2);

    // ---- epilogue.  D layout: lane holds tiles 4*lq .. 4*lq+3 (acc[p][i]) of output channel l15 of the block
    const int rows_out = WD == 56 ? 4 : (g * TRG + 1 < H / 4 ? 8 : 4);      // output rows of this row tile inside the image
    const float bias = a.bias[nb * 16 + l15];
    // the tile leaves as whole channel runs: rows_out * WD floats per channel are contiguous in the NCHW plane; iteration c of the
    // store loop = channel c, lane = 16-byte unit of the run (56 or 28 of the 64 lanes).  The residual is requested NOW, so that it
    // travels under the inverse transform.
    const int upc = rows_out * WD / 4;
    const int rvoff = lane < upc ? lane * 16 : kOOB;
    const size_t plane0 = (size_t)(nb * 16) * HW + (size_t)(g * TRG * 4) * WD;
    const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + ((size_t)img * a.out_ctot + a.out_coff) * HW + plane0), (short)0, 16 * HW * 4, 0x00020000);
    f32x4 res[16];
    const bool has_add = a.n_add == 1;
    if (has_add && kw == 0) {
        const __amdgpu_buffer_rsrc_t a_rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)(a.add[0] + ((size_t)img * a.add_ctot[0] + a.add_coff[0]) * HW + plane0), (short)0, 16 * HW * 4, 0x00020000);
#pragma unroll
        for (int c = 0; c < 16; ++c) res[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, rvoff, c * (HW * 4), 0));
    } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) res[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 y[4][4];                                       // [tile i][output row]
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
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* q = s[r];
            const float p12 = q[1] + q[2], m12 = q[1] - q[2], p34 = q[3] + q[4], m34 = q[3] - q[4];
            y[i][r] = f32x4{q[0] + p12 + p34, fmaf(2.f, m34, m12), fmaf(4.f, p34, p12), fmaf(8.f, m34, m12) + q[5]};
        }
    }
    int opos[4];                                         // LDS position of tile i's first row, -1: a padding slot
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = 4 * lq + i, stx = WD == 56 ? slot : (slot & 7), strl = WD == 56 ? 0 : (slot >> 3);
        opos[i] = stx < TPR && 4 * strl < rows_out ? l15 * kOStride + (4 * strl) * WD + 4 * stx : -1;
    }
    if constexpr (KS > 1) {                              // waves 1 .. KS-1 hand their partial outputs over, one after another
#pragma unroll 1
        for (int w = KS - 1; w >= 1; --w) {
            if (kw == w) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (opos[i] >= 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            f32x4 v = y[i][r];
                            if (w != KS - 1) v += *reinterpret_cast<const f32x4*>(O + opos[i] + r * WD);
                            *reinterpret_cast<f32x4*>(O + opos[i] + r * WD) = v;
                        }
                    }
            }
            __syncthreads();
        }
        if (kw != 0) return;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (opos[i] >= 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y[i][r] += *reinterpret_cast<const f32x4*>(O + opos[i] + r * WD);
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (opos[i] >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(O + opos[i] + r * WD) = y[i][r] + bias;
        }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        f32x4 v = *reinterpret_cast<const f32x4*>(O + c * kOStride + (lane < upc ? lane : 0) * 4) + res[c];
        if (a.relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), o_rsrc, rvoff, c * (HW * 4), 0);
    }
    GRK_W4R_STAMP(tk3);
    GRK_W4R_PHASE(0, tk0, tk1);
    GRK_W4R_PHASE(1, tk1, tk2);
    GRK_W4R_PHASE(2, tk2, tk3);
#ifdef GRNET_ABLATION
    if (lane == 0) { atomicAdd(&g_phase_w4r[3], 1ull); atomicAdd(&g_phase_w4r[4], acc_t4); atomicAdd(&g_phase_w4r[5], acc_t5); }
#endif
}

}  // namespace

bool conv_wino4r_eligible(int cin, int cout, int ks, int stride, int h, int w, int n_add) {
    return ks == 3 && stride == 1 && cin == cout && n_add <= 1 && ((cin == 32 && h == 56 && w == 56) || (cin == 64 && h == 28 && w == 28));
}

// a.w: pack_wino4r_weights; ksplit: waves per workgroup that split the input channels (1 or 2)
hipError_t launch_conv_wino4r(ConvArgs a, hipStream_t s, int ksplit) {
    if (!conv_wino4r_eligible(a.Cin, a.Cout, a.ks, a.stride, a.H, a.W, a.n_add) || a.N < 1) return hipErrorInvalidValue;
    if (a.n_add == 1 && a.add_shift[0] != 0) return hipErrorInvalidValue;
    a.gx = a.N * (a.W == 56 ? 14 : 4);
    a.xcd = a.gx % 8 == 0 && a.gx >= 16 ? 1 : 0;
    const int total = a.gx * (a.Cout / 16);
    hipError_t e;
    if (a.W == 56) e = ksplit == 2 ? launch_k(conv_wino4r_f32<56, 32, 2>, dim3(total), dim3(128), 0, s, a) : launch_k(conv_wino4r_f32<56, 32, 1>, dim3(total), dim3(64), 0, s, a);
    else if (ksplit == 1) e = launch_k(conv_wino4r_f32<28, 64, 1>, dim3(total), dim3(64), 0, s, a);
    else if (ksplit == 4) e = launch_k(conv_wino4r_f32<28, 64, 4>, dim3(total), dim3(256), 0, s, a);
    else e = launch_k(conv_wino4r_f32<28, 64, 2>, dim3(total), dim3(128), 0, s, a);
#ifdef GRNET_ABLATION
    static const bool phases = getenv("GRNET_BB_PHASES") != nullptr;
    if (phases && e == hipSuccess) {
        unsigned long long h[8] = {}, z[8] = {};
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_w4r), sizeof(h));
        hipMemcpyToSymbol(HIP_SYMBOL(g_phase_w4r), z, sizeof(z));
        const double n = h[3] ? (double)h[3] : 1.0;
        fprintf(stderr, "[wino4r phases] c %d w %d N %d ksplit %d waves %llu: per wave ticks  start->loop %.0f  k loop %.0f  epilogue %.0f | per wave, summed over k-steps: "
                "column pass + patch issue %.0f  halo + row pass + MFMAs %.0f\n", a.Cin, a.W, a.N, ksplit, h[3], h[0] / n, h[1] / n, h[2] / n, h[4] / n, h[5] / n);
    }
#endif
    return e;
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
