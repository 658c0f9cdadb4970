// Winograd F(4x4, 3x3) on the fp32 matrix cores for the widest 3x3 stride-1 layers on 56x56 maps (upsample head 256 -> 256, PARE head
// 480 -> 256: hrnet.py:444-451, pare.py:197-210 -- Conv2d 3x3 pad 1 + BatchNorm2d(eval) + ReLU).  A 4x4 output tile is 36 independent
// products instead of 144 multiplies: 2.25 per output against 4 for F(2x2, 3x3) (round 2; source in the history: commit 8d3a931) and 9 for the direct kernel.  The
// transforms have non-trivial coefficients (4, 5, 2, 8 and sixths in the filter transform, which is applied in fp64 at load):
// measured on a 256-channel layer in fp32 the result is 7.8e-6 of the output rms away from the exact sum (F(2x2,3x3): 8e-7, the
// direct fma chain: 1.9e-6) -- two orders of magnitude inside the 1e-3 bar, covered by the same parity tests.
//
// Loop discipline (worked out on the round-2 F(2x2,3x3) kernel): on gfx950 every vector instruction between fp32 MFMAs is matrix-pipe time: few of them,
// clustered).  One workgroup (4 waves): one image, ONE tile row = 14 tiles (output rows 4r .. 4r+3; one MFMA row tile, 2 rows
// idle), 64 output channels, all 36 points -- wave w owns points 9w .. 9w+8 = 9 x 4 accumulator tiles (144 registers).  Per chunk of
// 8 input channels: the 6 input rows of the chunk by LDS-DMA (rows contiguous in the NCHW plane, 16-byte units); thread (channel, tile, half)
// transforms HALF of a 6x6 patch -- 6 16-byte LDS reads + DPP for the two edge columns, three rows of B^T d and their 18 products
// with B, 18 LDS writes into V[point][channel][16 tiles]; 72 MFMAs per wave, A fragment = one LDS dword, the 4 B fragments = one
// 16-byte load straight from L2, requested a chunk ahead.
#include "kernels.h"

#include <cstdio>
#include <cstdlib>

namespace grk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GRNET_LDS_AS __attribute__((address_space(3)))

namespace {

#ifdef GRNET_ABLATION
__device__ unsigned long long g_w4phase[8];      // diagnostic build, dbg bit 3: ticks summed over workgroups (tools/wino_phases.py)
#define W4_TICK(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define W4_TICK(var) do { } while (0)
#endif

constexpr int kCK = 8;                       // input channels per chunk
constexpr int kRaw = 6 * 56;                 // raw floats per channel: 6 input rows
constexpr int kV = 36 * kCK * 16;            // V[point][channel][16 tile slots]
constexpr int kMrow = 20;                    // epilogue: [point][channel][16 MFMA rows + 4]
constexpr size_t kLdsB = sizeof(float) * (2 * kCK * kRaw + 2 * kV);     // 58 368 B
static_assert(sizeof(float) * 36 * 16 * kMrow <= kLdsB, "the epilogue tile reuses the staging area");

template <typename T>
__device__ __forceinline__ void landed(T& x) { asm volatile("" : "+v"(x)); }

// 1-D transforms.  B^T rows 0..2 / 3..5 of F(4,3) (Lavin & Gray):  [4 0 -5 0 1 0] [0 -4 -4 1 1 0] [0 4 -4 -1 1 0] /
// [0 -2 -1 2 1 0] [0 2 -1 -2 1 0] [0 4 0 -5 0 1]
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

// NB: 16-channel blocks per workgroup: 4, or 2 = HALF a 64-channel block of the same packed weights (the half-size workgroups of a
// layer's last round, ConvArgs::wsplit)
// WD: map width, 56 or 28.  A workgroup's 14 tiles are one tile row of a 56-wide map (6 input rows) or two tile rows of 7 of a 28-wide
// one (10 input rows; 7 tile rows per image = 3.5 groups: the last group's lower half reads zeros and stores nothing).
// WSPLIT: the 32-channel kernel on HALF a 64-channel block of weights packed for the 64-channel kernel (last-round workgroups).  The
// standalone 32-channel layers (NB = 2, !WSPLIT) have their own packing with the two k-steps of a chunk interleaved (pack_wino4_weights):
// one 16-byte load brings a point's B fragments of BOTH k-steps, 9 weight loads per chunk and wave instead of 18.
template <int NB, int WD, int ABL, bool WSPLIT>
__device__ __forceinline__ void conv_wino4_body(const ConvArgs& a) {
    constexpr bool PAIR = NB == 2 && !WSPLIT;
    constexpr int NLD = PAIR ? 9 : 18;                   // weight loads per chunk and wave
    constexpr int TPR = WD / 4, TRG = 14 / TPR, kRawW = (4 * TRG + 2) * WD, UPC = kRawW / 4;   // tiles per tile row, tile rows per workgroup, raw floats / units per channel
    static_assert(WD == 56 || WD == 28, "tile geometry");
    extern __shared__ __align__(16) float smem[];
    float* raw = smem;                                  // [2][8][kRawW]
    float* V = raw + 2 * kCK * kRaw;                    // [2][36][8][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;

    const int id = a.blk0 + (WSPLIT ? (int)(blockIdx.x >> 1) : (int)blockIdx.x), nb0 = WSPLIT ? 2 * (int)(blockIdx.x & 1) : 0;
    int bx, by;
    if (a.xcd) {
        const int j = id >> 3, x = id & 7, q = j / a.gy;
        by = j - q * a.gy;
        bx = x * (a.gx >> 3) + q;
    } else {
        bx = id / a.gy;
        by = id - bx * a.gy;
    }
    const int groups = ((a.H >> 2) + TRG - 1) / TRG;     // tile-row groups per image (14 or 4)
    const int img = bx / groups, r = bx - img * groups;
    const int co0 = WSPLIT ? by * 64 : by * (NB * 16);      // first channel of the weight block; channel n*16 + l sits at l*cstr + n
    constexpr int cstr = WSPLIT ? 4 : NB;
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);         // critical-chain layers
    else if (a.prio >= 2) __builtin_amdgcn_s_setprio(3);
    const int HW = a.H * a.W;
    const float* inb = a.in + ((size_t)img * a.in_ctot + a.in_coff) * HW;
    const int g0 = (4 * TRG * r - 1) * WD;               // plane index of raw[.][0]

    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, 36 * a.CinPad * a.CoutPad * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, a.Cin * HW * 4, 0x00020000);
    typedef float bfrag __attribute__((ext_vector_type(PAIR ? 4 : NB)));   // PAIR: {k-step 0: n0 n1, k-step 1: n0 n1}
    const int nchunks_u = a.CinPad / kCK;
    // byte offsets: a lane's part (VGPR) + point / chunk part (scalar)
    const int ub = PAIR ? ((lq * a.CoutPad * 2 + co0 * 2 + l15 * 4) * 4 + wave * 9 * nchunks_u * 4 * a.CoutPad * 2 * 4)
                        : ((wave * 9 * a.CinPad + lq) * a.CoutPad + co0 + l15 * cstr + nb0) * 4;
    const int u_point = PAIR ? nchunks_u * 4 * a.CoutPad * 2 * 4 : a.CinPad * a.CoutPad * 4;
    const int u_kstep = 4 * a.CoutPad * 4, u_chunk = PAIR ? 4 * a.CoutPad * 2 * 4 : kCK * a.CoutPad * 4;
    auto load_u = [&](int chunk, int g) -> bfrag {       // !PAIR: group g = (point wave*9 + g/2, k-step g%2); PAIR: g = point
        const int soff = PAIR ? chunk * u_chunk + g * u_point : chunk * u_chunk + (g >> 1) * u_point + (g & 1) * u_kstep;
        if constexpr (PAIR || NB == 4) return __builtin_bit_cast(bfrag, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, ub, soff, 0));
        else return __builtin_bit_cast(bfrag, __builtin_amdgcn_raw_buffer_load_b64(u_rsrc, ub, soff, 0));
    };
    int roff[3];                                         // raw rows: 672 units per chunk; -1 = no unit or a row outside the image
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = i * 256 + tid, ch = u / UPC, k = u - ch * UPC, gi = g0 + 4 * k;
        const bool unit = u < kCK * UPC, inside = gi >= 0 && gi < HW;
        roff[i] = unit && inside ? (ch * HW + gi) * 4 : -1;
        if (unit && !inside) {                           // rows above / below the image: zero once in both buffers, the DMA never writes there
            *reinterpret_cast<f32x4*>(raw + u * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(raw + kCK * kRawW + u * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    auto issue_raw = [&](int chunk) {
        const int soff = chunk * (kCK * 4) * HW;
        float* dst = raw + (chunk & 1) * (kCK * kRawW);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (roff[i] >= 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (GRNET_LDS_AS void*)(dst + (i * 256 + wave * 64) * 4), 16, roff[i], soff, 0, 0);
        asm volatile("" ::: "memory");                   // later loads stay behind these requests: the vmcnt waits below count on the order
    };

    // ---- input transform.  16 rows of 16 lanes: row = (channel 0..7, half 0..1), lane = tile 0..13 (lanes 14, 15 idle: their zeros are the
    // right neighbour of tile 13 and they write into the two padding tile slots).  A thread reads all 6 rows of its tile's 6x6 patch (own
    // columns 4t .. 4t+3 as one 16-byte read per row, 4t-1 / 4t+4 from the neighbour lanes by DPP, whose out-of-row zero is the image's
    // left padding) and produces rows 3*half .. 3*half+2 of B^T d B.
    const int row16 = tid >> 4, px = tid & 15, chn = row16 & 7;
    const int half = __builtin_amdgcn_readfirstlane(row16 >> 3);                         // wave-uniform: waves 0, 1 / 2, 3
    // WD = 28: the 16 lanes are two tile rows of 7 tiles + 1 idle lane each; idle lanes supply the zero on BOTH sides there
    const int pc = WD == 56 ? px : (px & 7), trl = WD == 56 ? 0 : (px >> 3);
    const bool real = pc < TPR;
    const int rpos = chn * kRawW + (4 * trl) * WD + 4 * (real ? pc : TPR - 1);
    const int slot = real ? trl * TPR + pc : 14 + (WD == 56 ? px - 14 : trl);            // tile slot in V (14, 15: padding)
    const int vpos = (half * 18) * (kCK * 16) + chn * 16 + slot;                         // + (rr * 6 + c) * 128 for row rr of the half, column c
    struct Tf { float d[6][6]; float e[3][6]; };
    auto tf_read = [&](Tf& t, const float* rp) {        // 6 LDS reads
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rp + i * WD);
            t.d[i][1] = v[0]; t.d[i][2] = v[1]; t.d[i][3] = v[2]; t.d[i][4] = v[3];
        }
    };
    auto tf_halo = [&](Tf& t) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            t.d[i][0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(WD == 56 || real ? t.d[i][4] : 0.f), 0x111, 0xf, 0xf, true));   // row_shr:1
            t.d[i][5] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(real ? t.d[i][1] : 0.f), 0x101, 0xf, 0xf, true));    // row_shl:1
        }
    };
    auto tf_rows = [&](Tf& t) {                         // three rows of B^T d, per column
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float col[6] = {t.d[0][j], t.d[1][j], t.d[2][j], t.d[3][j], t.d[4][j], t.d[5][j]};
            if (half == 0) bt_lo(col, t.e[0][j], t.e[1][j], t.e[2][j]);
            else bt_hi(col, t.e[0][j], t.e[1][j], t.e[2][j]);
        }
    };
    auto tf_cols = [&](Tf& t, float* vp) {              // (B^T d) B: all 6 columns of the three rows, 18 LDS writes
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            float o[6];
            bt_lo(t.e[rr], o[0], o[1], o[2]);
            bt_hi(t.e[rr], o[3], o[4], o[5]);
#pragma unroll
            for (int c = 0; c < 6; ++c) vp[(rr * 6 + c) * (kCK * 16)] = o[c];
        }
    };

    f32x4 acc[9][NB];                                    // [point of this wave][channel block]
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.CinPad / kCK;
    bfrag bq[NLD];                                       // B fragments of the 18 MFMA groups (PAIR: of the 9 points); each is re-requested for the next chunk behind its group
    W4_TICK(t_start);
    issue_raw(0);
    if (nchunks > 1) issue_raw(1);
#pragma unroll
    for (int g = 0; g < NLD; ++g) bq[g] = load_u(0, g);
    if constexpr (PAIR) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");   // the raw rows (requested before the weight loads) have landed
    else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    __syncthreads();
    W4_TICK(t_first);
    {
        Tf t;
        tf_read(t, raw + rpos);
        tf_halo(t);
        tf_rows(t);
        tf_cols(t, V + vpos);
    }
    W4_TICK(t_tf0);
    // ---- the chunk loop: three clusters of 24 MFMAs (three points x two k-steps x four channel blocks)
    float av[2][6];
    auto load_a = [&](int buf, int c, int set) {         // A fragments of cluster c: points 3c .. 3c+2 of this wave, both k-steps
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int p = wave * 9 + 3 * c + (k >> 1), ks = k & 1;
            av[set][k] = V[buf * kV + p * (kCK * 16) + (ks * 4 + lq) * 16 + l15];
        }
    };
    auto chunk = [&](int buf, bool with_transform, int next) {
        Tf t;
        const float* rp = raw + (next & 1) * (kCK * kRawW) + rpos;
        float* vp = V + (next & 1) * kV + vpos;
        if (ABL != 3) load_a(buf, 0, 0);
        if (with_transform && ABL != 1 && next + 1 < nchunks) issue_raw(next + 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // cluster c: (1) everything requested one cluster ago has landed
#pragma unroll
            for (int k = 0; k < 6; ++k) landed(av[c & 1][k]);
            if (with_transform && ABL != 2) {
                if (c == 1) { tf_halo(t); tf_rows(t); }
                if (c == 2) tf_cols(t, vp);
            }
            __builtin_amdgcn_sched_barrier(0);
            // (2) requests: the next cluster's A fragments, the next chunk's B fragments of the previous cluster, the next chunk's input rows
            if (c < 2 && ABL != 3) load_a(buf, c + 1, (c + 1) & 1);
            if (with_transform && ABL != 1 && c > 0) {
#pragma unroll
                for (int k = 0; k < NLD / 3; ++k) bq[(NLD / 3) * (c - 1) + k] = load_u(next, (NLD / 3) * (c - 1) + k);
            }
            if (with_transform && ABL != 2 && c == 0) tf_read(t, rp);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const int pi = 3 * c + (k >> 1);
                    const float bfr = PAIR ? bq[pi][(k & 1) * 2 + n] : bq[6 * c + k][n];
                    acc[pi][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c & 1][k], bfr, acc[pi][n], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (with_transform && ABL != 1) {
#pragma unroll
            for (int k = 0; k < NLD / 3; ++k) bq[2 * (NLD / 3) + k] = load_u(next, 2 * (NLD / 3) + k);
        }
    };
    auto meet = [&](bool last) {
        // this wave's share of raw(ch+1) has landed: it was requested before the 18 weight loads of the previous iteration, which may stay
        // in flight (loads return in order)
        if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (PAIR) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        __syncthreads();
    };
    for (int ch = 0; ch + 1 < nchunks; ++ch) {
        meet(false);
        chunk(ch & 1, true, ch + 1);
    }
    const bool has_add = a.n_add == 1;
    meet(nchunks > 1);
    // thread (channel c of the pass's 16, tile t): 4 output rows of 4 pixels.  The residual rows of a pass are requested a pass ahead
    // (pass 0's before the accumulators go to LDS): their round trip used to sit between the inverse transform and the stores of every
    // pass -- twice per launch of the 32-channel branch layers, which are the longest dependency chain of stages 2-4.
    const int ec = tid / 14, et = tid - ec * 14, etro = et / TPR, etx = et - etro * TPR, eorow = 4 * (TRG * r + etro);
    const bool ethread = tid < 14 * 16 && eorow < a.H;
    f32x4 radd[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    auto fetch_res = [&](int nt) {
        const int co = co0 + (nb0 + nt) * 16 + ec;
        if (has_add && ethread && co < a.Cout) {
            const float* ap = a.add[0] + ((size_t)img * a.add_ctot[0] + a.add_coff[0] + co) * HW + eorow * WD + 4 * etx;
#pragma unroll
            for (int i = 0; i < 4; ++i) radd[i] = *reinterpret_cast<const f32x4*>(ap + i * WD);
        }
    };
    fetch_res(0);                                       // under the last chunk's MFMAs
    chunk((nchunks - 1) & 1, false, 0);

    W4_TICK(t_loop);
    // ---- epilogue: inverse transform A^T M A (A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]), + bias, + residual, ReLU
    float* Mx = smem;                                    // [36 points][16 channels][20]
    for (int nt = 0; nt < NB; ++nt) {
        __syncthreads();
#pragma unroll
        for (int pi = 0; pi < 9; ++pi) {
            f32x4 v;
            if constexpr (NB == 4) v = nt == 0 ? acc[pi][0] : nt == 1 ? acc[pi][1] : nt == 2 ? acc[pi][2] : acc[pi][3];
            else v = nt == 0 ? acc[pi][0] : acc[pi][1];
            *reinterpret_cast<f32x4*>(Mx + ((wave * 9 + pi) * 16 + l15) * kMrow + lq * 4) = v;
        }
        __syncthreads();
        const f32x4 rcur[4] = {radd[0], radd[1], radd[2], radd[3]};
        if (nt + 1 < NB) fetch_res(nt + 1);
        if (ethread) {
            const int c = ec, t = et, tx = etx;
            const int co = co0 + (nb0 + nt) * 16 + c, orow = eorow;
            if (co < a.Cout) {
                float s[4][6];                           // A^T M: rows of the 4x6 intermediate
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float m[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) m[i] = Mx[((i * 6 + j) * 16 + c) * kMrow + t];
                    const float p12 = m[1] + m[2], m12 = m[1] - m[2], p34 = m[3] + m[4], m34 = m[3] - m[4];
                    s[0][j] = m[0] + p12 + p34;
                    s[1][j] = fmaf(2.f, m34, m12);
                    s[2][j] = fmaf(4.f, p34, p12);
                    s[3][j] = fmaf(8.f, m34, m12) + m[5];
                }
                const float b = a.bias[co];
                const size_t obase = ((size_t)img * a.out_ctot + a.out_coff + co) * HW + orow * WD + 4 * tx;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float* q = s[i];
                    const float p12 = q[1] + q[2], m12 = q[1] - q[2], p34 = q[3] + q[4], m34 = q[3] - q[4];
                    f32x4 y = f32x4{q[0] + p12 + p34 + b, fmaf(2.f, m34, m12) + b, fmaf(4.f, p34, p12) + b, fmaf(8.f, m34, m12) + q[5] + b};
                    if (has_add) y += rcur[i];
                    if (a.relu) { y[0] = fmaxf(y[0], 0.f); y[1] = fmaxf(y[1], 0.f); y[2] = fmaxf(y[2], 0.f); y[3] = fmaxf(y[3], 0.f); }
                    *reinterpret_cast<f32x4*>(a.out + obase + i * WD) = y;
                }
            }
        }
    }
#ifdef GRNET_ABLATION
    if ((a.dbg & 8) && tid == 0) {
        W4_TICK(t_end);
        atomicAdd(&g_w4phase[0], t_first - t_start); atomicAdd(&g_w4phase[1], t_tf0 - t_first); atomicAdd(&g_w4phase[2], t_loop - t_tf0);
        atomicAdd(&g_w4phase[3], t_end - t_loop); atomicAdd(&g_w4phase[4], 1ull);
    }
#endif
}

template <int NB, int WD = 56, int ABL = 0, bool WSPLIT = false>
__global__ __launch_bounds__(256) void conv_wino4_f32(const ConvArgs a) { conv_wino4_body<NB, WD, ABL, WSPLIT>(a); }

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_wino4w_f32: the same F(4x4,3x3) convolution for the WIDE layers on 56x56 maps (output channels in 128s: 480 -> 256, 256 -> 256,
// 128 -> 128 -- the upsample heads and the PARE head, 0.9 ms of a 3.7 ms step), round 4.  What the 4-wave kernel above loses on them is
// not the matrix pipe (its chunk loop is ~53 % MFMA) but everything a workgroup does per 64 output channels: the input transform, the
// raw-row DMA and the chunk skeleton are paid again by every 64-channel workgroup of the same tile row, and with one wave per SIMD every
// LDS / L2 round trip is dead time.  Here a workgroup is EIGHT waves = 128 output channels on one tile row:
//   * V (the transformed chunk) is built once and shared by both 64-channel halves: transform and DMA per output channel halve;
//   * two waves per SIMD: wave w and wave w+4 share a SIMD (waves go to SIMDs cyclically) and run the two halves of an iteration in
//     OPPOSITE order -- waves 0-3 transform chunk c+1, then multiply chunk c; waves 4-7 multiply first -- so a SIMD's two waves are in
//     different phases and each one's LDS / memory latencies are covered by the other's MFMAs;
//   * chunks of 16 input channels (all 512 threads transform: thread = (channel, tile, half of the patch)), one barrier per 16 channels;
//   * at most 256 registers per wave (two waves per SIMD): hipcc then keeps the 144 accumulators in ordinary VGPRs (no AGPR split), the
//     B fragments live in a ring of two clusters (48 registers) instead of a whole chunk (72), re-requested right behind their MFMAs.
// Wave w: points 9 (w & 3) .. + 8, output channels 64 (w >> 2) .. + 63 of the workgroup's 128 (4 accumulator tiles per point).
// LDS: raw rows [2][16][336] + V [2][36][16][16] = 116.7 KB (one workgroup per CU); the epilogue's [2 halves][36][16][20] reuses it.
constexpr int kCKW = 16;                              // input channels per chunk
constexpr int kVW = 36 * kCKW * 16;                   // V[point][channel][16 tile slots]
constexpr size_t kLdsW = sizeof(float) * (2 * kCKW * kRaw + 2 * kVW);      // 116 736 B (56-wide maps; 28-wide ones need less and take the same)
static_assert(sizeof(float) * 2 * 36 * 16 * kMrow <= kLdsW, "the epilogue tiles reuse the staging area");

// WD: map width, 56 (one tile row of 14 tiles) or 28 (two tile rows of 7, as in the 4-wave kernel).  NPW: 16-channel blocks per wave:
// 4 = 128 output channels per workgroup, 2 = 64 (the 64 -> 64 layers: HR branch 1 on 28x28 maps -- until round 4 two 32-channel
// workgroups, each transforming the same input -- and layer1 / the first upsample head on 56x56 maps).  Weights are packed as for the
// 4-wave kernel's 64-channel blocks: a lane's 16 bytes hold its channel of the four 16-channel blocks; with NPW = 2 a wave reads its 8.
template <int WD, int NPW>
__device__ __forceinline__ void conv_wino4w_body(const ConvArgs& a) {
    constexpr int TPR = WD / 4, TRG = 14 / TPR, kRawW = (4 * TRG + 2) * WD, UPC = kRawW / 4;   // tiles per tile row, tile rows per workgroup, raw floats / units per channel
    constexpr int COUTW = 2 * NPW * 16;                  // output channels per workgroup
    static_assert((WD == 56 || WD == 28) && (NPW == 4 || NPW == 2), "geometry");
    typedef float bfrag __attribute__((ext_vector_type(NPW)));
    extern __shared__ __align__(16) float smem[];
    float* raw = smem;                                  // [2][16][kRawW]
    float* V = raw + 2 * kCKW * kRaw;                   // [2][36][16][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, lq = lane >> 4;
    const int pg = wave & 3, hf = wave >> 2;            // point group; half of the workgroup's output channels = half of the patch it transforms

    const int id = blockIdx.x;
    int bx, by;
    if (a.xcd) {
        const int j = id >> 3, x = id & 7, q = j / a.gy;
        by = j - q * a.gy;
        bx = x * (a.gx >> 3) + q;
    } else {
        bx = id / a.gy;
        by = id - bx * a.gy;
    }
    const int groups = ((a.H >> 2) + TRG - 1) / TRG;     // tile-row groups per image (14 or 4)
    const int img = bx / groups, r = bx - img * groups;
    // first output channel of this wave and where it sits in the packed weights: channel n*16 + l of a 64-block is at l*4 + n
    const int co0 = by * COUTW + hf * (NPW * 16);
    const int wpos = NPW == 4 ? co0 : (co0 & ~63) + ((co0 >> 5) & 1) * 2;
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio >= 2) __builtin_amdgcn_s_setprio(3);
    const int HW = a.H * a.W;
    const float* inb = a.in + ((size_t)img * a.in_ctot + a.in_coff) * HW;
    const int g0 = (4 * TRG * r - 1) * WD;               // plane index of raw[.][0]

    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, 36 * a.CinPad * a.CoutPad * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, a.Cin * HW * 4, 0x00020000);
    const int ub = ((pg * 9 * a.CinPad + lq) * a.CoutPad + wpos + l15 * 4) * 4;       // a lane's part of a B-fragment address (bytes)
    const int u_point = a.CinPad * a.CoutPad * 4, u_kstep = 4 * a.CoutPad * 4, u_chunk = kCKW * a.CoutPad * 4;
    // B fragments of cluster q of a chunk: q = 2 * (point third t) + (k-step pair kp); entry e = 2 * (point in third) + (k-step in pair)
    auto load_u = [&](int chunk, int q, int e) -> bfrag {
        const int soff = chunk * u_chunk + (3 * (q >> 1) + (e >> 1)) * u_point + (2 * (q & 1) + (e & 1)) * u_kstep;
        if constexpr (NPW == 4) return __builtin_bit_cast(bfrag, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, ub, soff, 0));
        else return __builtin_bit_cast(bfrag, __builtin_amdgcn_raw_buffer_load_b64(u_rsrc, ub, soff, 0));
    };
    constexpr int NRU = (kCKW * UPC + 511) / 512;        // 16-byte units of a chunk's raw rows per thread (3)
    int roff[NRU];                                       // -1 = no unit or a row outside the image
#pragma unroll
    for (int i = 0; i < NRU; ++i) {
        const int u = i * 512 + tid, ch = u / UPC, k = u - ch * UPC, gi = g0 + 4 * k;
        const bool unit = u < kCKW * UPC, inside = gi >= 0 && gi < HW;
        roff[i] = unit && inside ? (ch * HW + gi) * 4 : -1;
        if (unit && !inside) {                           // rows above / below the image: zero once in both buffers, the DMA never writes there
            *reinterpret_cast<f32x4*>(raw + u * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(raw + kCKW * kRawW + u * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    auto issue_raw = [&](int chunk) {
        const int soff = chunk * (kCKW * 4) * HW;
        float* dst = raw + (chunk & 1) * (kCKW * kRawW);
#pragma unroll
        for (int i = 0; i < NRU; ++i)
            if (roff[i] >= 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (GRNET_LDS_AS void*)(dst + (i * 512 + wave * 64) * 4), 16, roff[i], soff, 0, 0);
        asm volatile("" ::: "memory");
    };

    // ---- input transform: thread = (channel 0..15, tile slot 0..15, half), as in the 4-wave kernel (WD = 28: the 16 lanes are two tile rows of
    // 7 tiles + 1 idle lane each; idle lanes supply the zero on both sides there)
    const int row16 = tid >> 4, px = tid & 15, chn = row16 & 15;
    const int pc = WD == 56 ? px : (px & 7), trl = WD == 56 ? 0 : (px >> 3);
    const bool real = pc < TPR;
    const int rpos = chn * kRawW + (4 * trl) * WD + 4 * (real ? pc : TPR - 1);
    const int slot = real ? trl * TPR + pc : 14 + (WD == 56 ? px - 14 : trl);
    const int vpos = (hf * 18) * (kCKW * 16) + chn * 16 + slot;            // + (rr * 6 + c) * 256 for row rr of the half, column c
    auto transform = [&](const float* rp, float* vp) {
        float e[3][6];
        float d[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rp + i * WD);
            d[i][1] = v[0]; d[i][2] = v[1]; d[i][3] = v[2]; d[i][4] = v[3];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            d[i][0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(WD == 56 || real ? d[i][4] : 0.f), 0x111, 0xf, 0xf, true));   // row_shr:1
            d[i][5] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(real ? d[i][1] : 0.f), 0x101, 0xf, 0xf, true));               // row_shl:1
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
            if (hf == 0) bt_lo(col, e[0][j], e[1][j], e[2][j]);
            else bt_hi(col, e[0][j], e[1][j], e[2][j]);
        }
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            float o[6];
            bt_lo(e[rr], o[0], o[1], o[2]);
            bt_hi(e[rr], o[3], o[4], o[5]);
#pragma unroll
            for (int c = 0; c < 6; ++c) vp[(rr * 6 + c) * (kCKW * 16)] = o[c];
        }
    };

    f32x4 acc[9][NPW];                                   // [point of this wave][16-channel block of its half]
#pragma unroll
    for (int p = 0; p < 9; ++p)
#pragma unroll
        for (int n = 0; n < NPW; ++n) acc[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.CinPad / kCKW;
    bfrag bq[6];                                         // B fragments of ONE cluster; every entry is re-requested for the next cluster right behind its MFMAs
    float av[2][6];
    auto load_a = [&](int buf, int q, int set) {         // A fragments of cluster q: points 3 (q >> 1) .. + 2 of this wave, k-steps 2 (q & 1), + 1
#pragma unroll
        for (int e = 0; e < 6; ++e) {
            const int p = pg * 9 + 3 * (q >> 1) + (e >> 1), ks = 2 * (q & 1) + (e & 1);
            av[set][e] = V[buf * kVW + p * (kCKW * 16) + (ks * 4 + lq) * 16 + l15];
        }
    };
    // the six clusters of a chunk: 6 x NPW MFMAs each (three points x two k-steps x NPW channel blocks).  Entry e of bq is consumed by NPW
    // MFMAs and re-requested at once for the next cluster (of the next chunk after cluster 5): the other entries' MFMAs and the SIMD's
    // other wave cover the L2 round trip.
    auto multiply = [&](int buf, int ch, bool more) {
        load_a(buf, 0, 0);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#pragma unroll
            for (int e = 0; e < 6; ++e) landed(av[q & 1][e]);
            __builtin_amdgcn_sched_barrier(0);
            if (q < 5) load_a(buf, q + 1, (q + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 6; ++e) {
                const int pi = 3 * (q >> 1) + (e >> 1);
#pragma unroll
                for (int n = 0; n < NPW; ++n) acc[pi][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q & 1][e], bq[e][n], acc[pi][n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (q < 5) bq[e] = load_u(ch, q + 1, e);
                else if (more) bq[e] = load_u(ch + 1, 0, e);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // thread (half h, channel c of the pass's 16, tile t) of the epilogue: 4 output rows of 4 pixels; the residual rows of a pass are requested a pass ahead
    const bool has_add = a.n_add == 1;
    const int eh = tid >> 8, et2 = tid & 255, ec = et2 / 14, et = et2 - ec * 14, etro = et / TPR, etx = et - etro * TPR, eorow = 4 * (TRG * r + etro);
    const bool ethread = et2 < 14 * 16 && eorow < a.H;
    f32x4 radd[4] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    auto fetch_res = [&](int nt) {
        const int co = by * COUTW + eh * (NPW * 16) + nt * 16 + ec;
        if (has_add && ethread && co < a.Cout) {
            const float* ap = a.add[0] + ((size_t)img * a.add_ctot[0] + a.add_coff[0] + co) * HW + eorow * WD + 4 * etx;
#pragma unroll
            for (int i = 0; i < 4; ++i) radd[i] = *reinterpret_cast<const f32x4*>(ap + i * WD);
        }
    };

    issue_raw(0);
    if (nchunks > 1) issue_raw(1);
#pragma unroll
    for (int e = 0; e < 6; ++e) bq[e] = load_u(0, 0, e);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // this wave's shares of raw(0), raw(1) (requested before the 6 weight loads) have landed
    __syncthreads();
    // Barrier #k separates "every wave has transformed chunk k" from "any wave multiplies chunk k".  Between two barriers waves 0-3 run
    // [transform k+1, multiply k] and waves 4-7 [multiply k, transform k+1]: the two waves of a SIMD are never both in their transform, and
    // each one's LDS / memory latencies are covered by the other's MFMAs.  Written as ONE loop with ONE multiply -- waves 0-3 take the
    // barrier in front of their transform, waves 4-7 behind it -- because with the two orders spelled out hipcc copied the 144 accumulators
    // at every merge point and spilled 127 registers.  raw(k + 2) is requested right behind barrier #k (its buffer held chunk k, which every
    // wave transformed before that barrier) and has landed when its requester reaches barrier #(k + 1).
    if (hf == 0) transform(raw + rpos, V + vpos);
    for (int k = 0; k < nchunks; ++k) {
        if (hf == 0) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __syncthreads();
            if (k + 2 < nchunks) issue_raw(k + 2);
        }
        const int tk = k + 1 - hf;                            // the chunk this wave transforms in this iteration
        if (tk < nchunks) transform(raw + (tk & 1) * (kCKW * kRawW) + rpos, V + (tk & 1) * kVW + vpos);
        if (hf != 0) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __syncthreads();
            if (k + 2 < nchunks) issue_raw(k + 2);
        }
        if (k + 1 == nchunks) fetch_res(0);                  // under the last chunk's MFMAs
        __builtin_amdgcn_sched_barrier(0);
        multiply(k & 1, k, k + 1 < nchunks);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: inverse transform A^T M A, + bias, + residual, ReLU; both halves go through LDS together, 16 channels each per pass
    float* Mx = smem + hf * (36 * 16 * kMrow);           // this wave's half: [36 points][16 channels][20]
    const float* Mr = smem + eh * (36 * 16 * kMrow);     // the half this thread reads
    for (int nt = 0; nt < NPW; ++nt) {
        __syncthreads();
#pragma unroll
        for (int pi = 0; pi < 9; ++pi) {
            f32x4 v;
            if constexpr (NPW == 4) v = nt == 0 ? acc[pi][0] : nt == 1 ? acc[pi][1] : nt == 2 ? acc[pi][2] : acc[pi][3];
            else v = nt == 0 ? acc[pi][0] : acc[pi][1];
            *reinterpret_cast<f32x4*>(Mx + ((pg * 9 + pi) * 16 + l15) * kMrow + lq * 4) = v;
        }
        __syncthreads();
        const f32x4 rcur[4] = {radd[0], radd[1], radd[2], radd[3]};
        if (nt + 1 < NPW) fetch_res(nt + 1);
        if (ethread) {
            const int co = by * COUTW + eh * (NPW * 16) + nt * 16 + ec;
            if (co < a.Cout) {
                float s4[4][6];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    float m[6];
#pragma unroll
                    for (int i = 0; i < 6; ++i) m[i] = Mr[((i * 6 + j) * 16 + ec) * kMrow + et];
                    const float p12 = m[1] + m[2], m12 = m[1] - m[2], p34 = m[3] + m[4], m34 = m[3] - m[4];
                    s4[0][j] = m[0] + p12 + p34;
                    s4[1][j] = fmaf(2.f, m34, m12);
                    s4[2][j] = fmaf(4.f, p34, p12);
                    s4[3][j] = fmaf(8.f, m34, m12) + m[5];
                }
                const float b = a.bias[co];
                const size_t obase = ((size_t)img * a.out_ctot + a.out_coff + co) * HW + eorow * WD + 4 * etx;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float* q = s4[i];
                    const float p12 = q[1] + q[2], m12 = q[1] - q[2], p34 = q[3] + q[4], m34 = q[3] - q[4];
                    f32x4 y = f32x4{q[0] + p12 + p34 + b, fmaf(2.f, m34, m12) + b, fmaf(4.f, p34, p12) + b, fmaf(8.f, m34, m12) + q[5] + b};
                    if (has_add) y += rcur[i];
                    if (a.relu) { y[0] = fmaxf(y[0], 0.f); y[1] = fmaxf(y[1], 0.f); y[2] = fmaxf(y[2], 0.f); y[3] = fmaxf(y[3], 0.f); }
                    *reinterpret_cast<f32x4*>(a.out + obase + i * WD) = y;
                }
            }
        }
    }
}

template <int WD, int NPW>
__global__ __launch_bounds__(512) void conv_wino4w_f32(const ConvArgs a) { conv_wino4w_body<WD, NPW>(a); }

}  // namespace

// 16-channel blocks per workgroup: 4 where the output channels come in 64s -- except for exactly 64 channels on a 28x28 map, where
// 32-channel workgroups double a grid that would otherwise be 64 workgroups for 256 CUs (in context, 16 frames: 3 693 frames/s
// against 3 310 with 64-channel workgroups there)
// Eight-wave workgroups (conv_wino4w_f32): 16-channel blocks per wave, 0 = the 4-wave kernel.  GRNET_WINO_WIDE is a mask (default 1):
// bit 0 the 56x56 layers whose output channels come in 128s (upsample heads, PARE head): 480 -> 256 459 -> 357 us, 256 -> 256 264 -> 202,
// 128 -> 128 78 -> 57 at 16 frames, the step 3.71 -> 3.52 ms.  The other shapes the kernel is instantiated for were measured and stay off:
// bit 1 the 64 -> 64 layers of HR branch 1 on 28x28 maps (64 workgroups instead of 128: 18.7 us alone against 15.6, the step 3.53 -> 3.74 ms --
// the branch chains need the short launches), bit 2 the 28x28 layers with output channels in 128s (half the CUs: 95 / 51 us against 74 / 39,
// step unchanged), bit 3 the 64-channel layers on 56x56 maps (21.6 us alone against 24.2, but 3.530 -> 3.549 ms in the step).
int conv_wino4_wide(int cout, int w) {
    static const int wide_env = GRNET_AB(WINO_WIDE, 1);
    if (w == 56) return cout % 128 == 0 ? ((wide_env & 1) ? 4 : 0) : cout == 64 ? ((wide_env & 8) ? 2 : 0) : 0;
    if (w == 28) return cout % 128 == 0 ? ((wide_env & 4) ? 4 : 0) : cout == 64 ? ((wide_env & 2) ? 2 : 0) : 0;
    return 0;
}
int conv_wino4_blocks(int cout, int w) { return (cout % 64 == 0 && !(w == 28 && cout == 64 && !conv_wino4_wide(cout, w))) ? 4 : 2; }

bool conv_wino4_eligible(int cin, int cout, int ks, int stride, int h, int w, int n_add) {
    return ks == 3 && stride == 1 && ((h == 56 && w == 56) || (h == 28 && w == 28)) && n_add <= 1 && cin % kCK == 0 && cout % 32 == 0 && cin >= 32;
}

// GRNET_WINO_LDS_PAD (diagnostic): extra bytes of LDS the 4-wave launches ask for -- how sensitive the step is to the workgroups per CU
static size_t wino4_lds() {
    static const size_t pad = (size_t)GRNET_AB(WINO_LDS_PAD, 0);
    return kLdsB + pad;
}

template <int WD>
static hipError_t launch_wino4_w(ConvArgs a, hipStream_t s, int nb, int* n_launches) {
    constexpr int TRG = 14 / (WD / 4);
    const size_t kLdsB = wino4_lds();
    a.gx = a.N * (((a.H >> 2) + TRG - 1) / TRG);
    a.gy = a.CoutPad / (nb * 16);
    a.xcd = a.gx % 8 == 0 && a.gx >= 16 ? 1 : 0;
    a.blk0 = 0;
    a.wsplit = 0;
    if (n_launches) *n_launches = 1;
    const int total = a.gx * a.gy;
#ifdef GRNET_ABLATION
    if constexpr (WD == 56) {
        if (a.dbg >= 1 && a.dbg <= 3) {
            auto go = [&](auto kern) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsB); return launch_k(kern, dim3(total), dim3(256), kLdsB, s, a); };
            if (a.dbg == 1) return go(conv_wino4_f32<4, 56, 1>);
            if (a.dbg == 2) return go(conv_wino4_f32<4, 56, 2>);
            return go(conv_wino4_f32<4, 56, 3>);
        }
    }
#endif
    if (nb == 2) return launch_k(conv_wino4_f32<2, WD>, dim3(total), dim3(256), kLdsB, s, a);
    // a last round of workgroups that is at most half full runs as twice as many half-size workgroups
    static const int split_env = GRNET_AB(WINO_SPLIT, 1);
    int kCUs = 0;                                        // workgroups per round = CUs of this device (one workgroup fits a CU)
    if (hipError_t e = device_cu_count(&kCUs); e != hipSuccess) return e;
    const int full = total / kCUs * kCUs, rest = total - full;
    if (split_env && full > 0 && rest > 0 && 2 * rest <= kCUs && (!a.xcd || full % 8 == 0)) {
        hipError_t e = launch_k(conv_wino4_f32<4, WD>, dim3(full), dim3(256), kLdsB, s, a);
        if (e != hipSuccess) return e;
        a.blk0 = full;
        a.wsplit = 1;
        if (n_launches) *n_launches = 2;
        return launch_k(conv_wino4_f32<2, WD, 0, true>, dim3(2 * rest), dim3(256), kLdsB, s, a);
    }
    return launch_k(conv_wino4_f32<4, WD>, dim3(total), dim3(256), kLdsB, s, a);
}

// a.w: transformed weights [36][CinPad][CoutPad] (pack_wino4_weights), CoutPad % 64 == 0 (or 32: the 32-channel variant)
hipError_t launch_conv_wino4(ConvArgs a, hipStream_t s, int* n_launches) {
    static PerDeviceOnce attr;
    int dev = 0;
    if (hipError_t e = current_device(&dev); e != hipSuccess) return e;
    if (hipError_t e = once_per_device(attr, dev, [](int*) {
            hipError_t e = hipSuccess;
            auto set = [&](auto kern) { if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino4_lds()); };
            set(conv_wino4_f32<4, 56>); set(conv_wino4_f32<2, 56>); set(conv_wino4_f32<4, 28>); set(conv_wino4_f32<2, 28>);
            set(conv_wino4_f32<2, 56, 0, true>); set(conv_wino4_f32<2, 28, 0, true>);
            auto setw = [&](auto kern) { if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsW); };
            setw(conv_wino4w_f32<56, 4>); setw(conv_wino4w_f32<56, 2>); setw(conv_wino4w_f32<28, 4>); setw(conv_wino4w_f32<28, 2>);
            return e;
        }); e != hipSuccess) return e;
    const int nb = conv_wino4_blocks(a.Cout, a.W);
    if (!conv_wino4_eligible(a.Cin, a.Cout, a.ks, a.stride, a.H, a.W, a.n_add) || a.CinPad % kCK != 0 || a.CoutPad % (nb * 16) != 0) return hipErrorInvalidValue;
    if (a.n_add == 1 && a.add_shift[0] != 0) return hipErrorInvalidValue;
    // eight-wave workgroups (conv_wino4w_f32) where the layer's shape has them; test hint 2003 (dbg bit 5) forces the 4-wave kernel
    const int npw = conv_wino4_wide(a.Cout, a.W);
    if (npw && a.CinPad % kCKW == 0 && a.CoutPad % (npw * 32) == 0 && !(a.dbg & 32)) {
        const int trg = a.W == 56 ? 1 : 2;
        a.gx = a.N * (((a.H >> 2) + trg - 1) / trg);
        a.gy = a.CoutPad / (npw * 32);
        a.xcd = a.gx % 8 == 0 && a.gx >= 16 ? 1 : 0;
        a.blk0 = 0;
        a.wsplit = 0;
        if (n_launches) *n_launches = 1;
        const dim3 grid(a.gx * a.gy);
        if (a.W == 56) return npw == 4 ? launch_k(conv_wino4w_f32<56, 4>, grid, dim3(512), kLdsW, s, a) : launch_k(conv_wino4w_f32<56, 2>, grid, dim3(512), kLdsW, s, a);
        return npw == 4 ? launch_k(conv_wino4w_f32<28, 4>, grid, dim3(512), kLdsW, s, a) : launch_k(conv_wino4w_f32<28, 2>, grid, dim3(512), kLdsW, s, a);
    }
#ifdef GRNET_ABLATION
    if (GRNET_AB_SET(W4_PHASES)) {
        a.dbg |= 8;
        const hipError_t e = a.W == 56 ? launch_wino4_w<56>(a, s, nb, n_launches) : launch_wino4_w<28>(a, s, nb, n_launches);
        unsigned long long h[8] = {}, z[8] = {};
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_w4phase), sizeof(h));
        hipMemcpyToSymbol(HIP_SYMBOL(g_w4phase), z, sizeof(z));
        const double n = h[4] ? (double)h[4] : 1.0;
        fprintf(stderr, "[wino4 phases] %d->%d @%d N%d add %d nb %d wgs %llu: per WG ticks  prologue %.0f  first transform %.0f  chunk loop %.0f (%d chunks)  epilogue %.0f\n",
                a.Cin, a.Cout, a.W, a.N, a.n_add, nb, h[4], h[0] / n, h[1] / n, h[2] / n, a.CinPad / kCK, h[3] / n);
        return e;
    }
#endif
    return a.W == 56 ? launch_wino4_w<56>(a, s, nb, n_launches) : launch_wino4_w<28>(a, s, nb, n_launches);
}

// the filter transform of F(4x4,3x3), U = G g G^T, in fp64: g (3,3) row-major -> u[i * 6 + j]
void wino4_transform_filter(const double* g, double* u) {
    static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    double t[6][3];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[j] + G[i][1] * g[3 + j] + G[i][2] * g[6 + j];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) u[i * 6 + j] = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
}

// U = G g G^T per (cout, cin) in fp64 -> [36][cin_pad][cout_pad] fp32; w: (cout, cin, 3, 3) folded weights (double)
void pack_wino4_weights(const double* w, int cout, int cin, int cin_pad, int cout_pad, float* out, int wid) {
    static const double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
    const int nb = conv_wino4_blocks(cout, wid), tc = nb * 16;              // the kernel variant launch_conv_wino4 picks for this layer
    for (size_t i = 0; i < (size_t)36 * cin_pad * cout_pad; ++i) out[i] = 0.f;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            // within a workgroup's tc channels, channel n*16 + l sits at l*nb + n: lane l's nb MFMA B fragments are one load
            const int cpos = (co / tc) * tc + (co % 16) * nb + (co % tc) / 16;
            const double* g = w + ((size_t)co * cin + ci) * 9;
            double t[6][3];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[j] + G[i][1] * g[3 + j] + G[i][2] * g[6 + j];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) {
                    const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                    const int p = i * 6 + j;
                    if (nb == 4) out[((size_t)p * cin_pad + ci) * cout_pad + cpos] = (float)u;
                    else {                               // 32-channel layers: [point][chunk][k % 4][32-channel block][channel l][k-step][n]
                        const int chunk = ci / kCK, ks = (ci % kCK) / 4, kq = ci % 4;
                        out[(((size_t)p * (cin_pad / kCK) + chunk) * 4 + kq) * ((size_t)cout_pad * 2) + (co / 32) * 64 + (co % 16) * 4 + ks * 2 + (co % 32) / 16] = (float)u;
                    }
                }
        }
}

}  // namespace grk
