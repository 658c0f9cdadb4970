// One launch per BasicBlock of the two narrow HR branches (hrnet.py:30-59: conv3x3 - BN - ReLU - conv3x3 - BN - (+x) - ReLU, as
// instantiated by hrnet.py:141-187 for 32 channels on 56x56 maps and 64 channels on 28x28 maps): both convolutions as Winograd
// F(4x4, 3x3) on the fp32 matrix cores (the loop of conv_wino4.hip), the intermediate tensor never leaving the LDS.
//
// A workgroup (4 waves) owns ALL channels of one image's output rows 4r .. 4r+3.
//   phase 1  conv1 + bias + ReLU on the two tile rows 4r-1 .. 4r+6 (the six rows 4r-1 .. 4r+4 are what conv2 reads; the tile grid of
//            conv1 is shifted by one row for that, so the halo costs one extra tile row, not two): input rows 4r-2 .. 4r+7 by LDS-DMA
//            in chunks of 8 channels, 28 tiles = two MFMA row tiles on a 56-wide map (every B fragment feeds both), 14 tiles = one
//            on a 28-wide map; the result goes through the inverse transform into Y[channel][6 rows][W] in LDS -- rows outside the
//            image as zeros, which is conv2's zero padding.
//   phase 2  conv2 on the one tile row 4r .. 4r+3 with Y as its input rows (no global reads but the weights), + bias + x + ReLU.
// The first B fragments of conv2 are requested before the epilogue of phase 1.  Both weight tensors are the ones pack_wino4_weights
// lays out for the 32-channel kernel (k-steps of a chunk interleaved: one 16-byte load = a point's fragments of both k-steps for a
// block of 32 output channels), so the fused and the per-convolution launches share them.
#include "kernels.h"

namespace grk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define GRNET_LDS_AS __attribute__((address_space(3)))

namespace {

constexpr int kCK = 8;                       // input channels per chunk
constexpr int kMrow = 20;                    // epilogue: [point][channel][16 MFMA rows + 4]
constexpr int kMx = 36 * 16 * kMrow;         // floats of one epilogue pass

template <typename T>
__device__ __forceinline__ void landed(T& x) { asm volatile("" : "+v"(x)); }

// B^T of F(4,3), rows 0..2 / 3..5 (as in conv_wino4.hip)
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

template <int WD>
struct Geo {
    static constexpr int C = WD == 56 ? 32 : 64;        // channels of the block
    static constexpr int NB = C / 16;                   // MFMA column blocks
    static constexpr int NBLK = C / 32;                 // 32-channel weight blocks
    static constexpr int NLD = 9 * NBLK;                // weight loads per chunk and wave
    static constexpr int TPR = WD / 4;                  // tiles per tile row
    static constexpr int MT1 = WD == 56 ? 2 : 1;        // MFMA row tiles of phase 1 (28 / 14 tiles)
    static constexpr int H = WD, HW = WD * WD;
    static constexpr int kRaw1 = 10 * WD;               // phase 1: raw floats per channel (10 input rows)
    static constexpr int UPC1 = kRaw1 / 4;              // 16-byte units per channel
    static constexpr int NIT = (kCK * UPC1 + 255) / 256;
    static constexpr int kY = 6 * WD;                   // Y floats per channel
    static constexpr int kV1 = 36 * kCK * 16 * MT1, kV2 = 36 * kCK * 16;
    static constexpr int nchunks = C / kCK;
    static constexpr int raw_floats = 2 * kCK * kRaw1, v_floats = 2 * kV1, y_floats = C * kY;
    static constexpr size_t lds_bytes = sizeof(float) * (raw_floats + v_floats + y_floats);
    static_assert(raw_floats + v_floats >= kMx, "the epilogue tile reuses the staging area");
    static constexpr int u_point = nchunks * 4 * C * 2 * 4, u_chunk = 4 * C * 2 * 4;     // bytes (pack_wino4_weights, 32-channel layout)
};

struct Tf { float d[6][6]; float e[3][6]; };

#ifdef GRNET_ABLATION
__device__ unsigned long long g_phase_bb[8];   // [0] start -> raw landed, [1] first transform, [2] phase-1 loop, [3] epilogue 1, [4] first transform of phase 2, [5] phase-2 loop, [6] epilogue 2, [7] workgroups
#define GRK_BB_STAMP(var) const unsigned long long var = __builtin_readcyclecounter()
#define GRK_BB_PHASE(i, t0, t1) do { if (threadIdx.x == 0) atomicAdd(&g_phase_bb[i], (t1) - (t0)); } while (0)
#else
#define GRK_BB_STAMP(var) do {} while (0)
#define GRK_BB_PHASE(i, t0, t1) do {} while (0)
#endif

template <int WD>
__device__ __forceinline__ void bblock_body(const BlockArgs& a) {
    typedef Geo<WD> G;
    constexpr int C = G::C, NB = G::NB, NBLK = G::NBLK, NLD = G::NLD, TPR = G::TPR, MT1 = G::MT1, H = G::H, HW = G::HW;
    constexpr int kRaw1 = G::kRaw1, UPC1 = G::UPC1, NIT = G::NIT, kY = G::kY, kV1 = G::kV1, kV2 = G::kV2, nchunks = G::nchunks;
    extern __shared__ __align__(16) float smem[];
    float* raw = smem;                                   // [2][8][10 rows][WD]
    float* V = raw + G::raw_floats;                      // phase 1: [2][36][8][16 * MT1], phase 2: [2][36][8][16]
    float* Y = V + G::v_floats;                          // [C][6][WD]: relu(conv1 + b1), rows 4r-1 .. 4r+4
    float* Mx = smem;                                    // epilogue passes: [36][16][20]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    GRK_BB_STAMP(tk0);

    int bx = blockIdx.x;
    if (a.xcd) bx = (int)(blockIdx.x & 7) * (a.gx >> 3) + (int)(blockIdx.x >> 3);      // an XCD owns a contiguous range of (image, tile row)
    const int img = bx / (H / 4), r = bx - img * (H / 4);
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio >= 2) __builtin_amdgcn_s_setprio(3);
    const float* inb = a.in + ((size_t)img * a.in_ctot + a.in_coff) * HW;
    const int g0 = (4 * r - 2) * WD;                     // plane index of raw row 0

    const __amdgpu_buffer_rsrc_t u1_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w1, (short)0, 36 * C * C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t u2_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2, (short)0, 36 * C * C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, C * HW * 4, 0x00020000);
    // weights: lane part (VGPR) + (point, block, chunk) part (scalar); fragment = {k-step 0: n0 n1, k-step 1: n0 n1} of a 32-channel block
    const int ub = (lq * C * 2 + l15 * 4) * 4 + wave * 9 * G::u_point;
    auto load_u = [&](const __amdgpu_buffer_rsrc_t& rs, int chunk, int g) -> f32x4 {       // g = point * NBLK + block
        const int soff = chunk * G::u_chunk + (g / NBLK) * G::u_point + (g % NBLK) * 256;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ub, soff, 0));
    };

    // ---- phase 1 staging: 10 input rows per channel, 8 channels per chunk
    int roff[NIT];                                       // -1 = no unit or a row outside the image
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int u = i * 256 + tid, ch = u / UPC1, k = u - ch * UPC1, gi = g0 + 4 * k;
        const bool unit = u < kCK * UPC1, inside = gi >= 0 && gi < HW;
        roff[i] = unit && inside ? (ch * HW + gi) * 4 : -1;
        if (unit && !inside) {                           // rows above / below the image: zero once in both buffers, the DMA never writes there
            *reinterpret_cast<f32x4*>(raw + u * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(raw + kCK * kRaw1 + u * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    auto issue_raw = [&](int chunk) {
        const int soff = chunk * (kCK * 4) * HW;
        float* dst = raw + (chunk & 1) * (kCK * kRaw1);
#pragma unroll
        for (int i = 0; i < NIT; ++i)
            if (roff[i] >= 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (GRNET_LDS_AS void*)(dst + (i * 256 + wave * 64) * 4), 16, roff[i], soff, 0, 0);
        asm volatile("" ::: "memory");                   // later loads stay behind these requests: the vmcnt waits below count on the order
    };

    // ---- input transforms.  Rows of 16 lanes = the tiles of one tile row (idle lanes supply the zero padding to their DPP neighbours
    // and write into padding tile slots).  WD = 56, phase 1: thread (channel, tile row, tile) transforms a whole 6x6 patch in two halves;
    // otherwise thread (channel, half, tile) produces three of the six rows of B^T d B (conv_wino4.hip).
    const int row16 = tid >> 4, px = tid & 15;
    auto tf_read = [&](Tf& t, const float* rp) {        // 6 LDS reads: own columns 4t .. 4t+3 of the six patch rows
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rp + i * WD);
            t.d[i][1] = v[0]; t.d[i][2] = v[1]; t.d[i][3] = v[2]; t.d[i][4] = v[3];
        }
    };
    auto tf_halo = [&](Tf& t, bool real) {              // columns 4t-1 / 4t+4 from the neighbour lanes; out-of-row = the image's zero padding
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            t.d[i][0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(WD == 56 || real ? t.d[i][4] : 0.f), 0x111, 0xf, 0xf, true));   // row_shr:1
            t.d[i][5] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(real ? t.d[i][1] : 0.f), 0x101, 0xf, 0xf, true));    // row_shl:1
        }
    };
    auto tf_rows = [&](Tf& t, int half) {               // three rows of B^T d, per column
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float col[6] = {t.d[0][j], t.d[1][j], t.d[2][j], t.d[3][j], t.d[4][j], t.d[5][j]};
            if (half == 0) bt_lo(col, t.e[0][j], t.e[1][j], t.e[2][j]);
            else bt_hi(col, t.e[0][j], t.e[1][j], t.e[2][j]);
        }
    };
    auto tf_cols = [&](Tf& t, float* vp, int pstride) { // (B^T d) B: all 6 columns of the three rows, 18 LDS writes
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            float o[6];
            bt_lo(t.e[rr], o[0], o[1], o[2]);
            bt_hi(t.e[rr], o[3], o[4], o[5]);
#pragma unroll
            for (int c = 0; c < 6; ++c) vp[(rr * 6 + c) * pstride] = o[c];
        }
    };
    // phase 1 positions
    int rpos1, vpos1, half1 = 0;
    bool real1;
    if constexpr (WD == 56) {                            // (channel 0..7, tile row 0..1) x 16 lanes
        const int chn = row16 >> 1, tr = row16 & 1;
        real1 = px < TPR;
        rpos1 = chn * kRaw1 + (4 * tr) * WD + 4 * (real1 ? px : TPR - 1);
        vpos1 = chn * 32 + tr * 16 + px;
    } else {                                             // (channel 0..7, half 0..1) x (two tile rows of 7 tiles + 1 idle lane each)
        const int chn = row16 & 7, pc = px & 7, trl = px >> 3;
        half1 = __builtin_amdgcn_readfirstlane(row16 >> 3);
        real1 = pc < TPR;
        rpos1 = chn * kRaw1 + (4 * trl) * WD + 4 * (real1 ? pc : TPR - 1);
        vpos1 = (half1 * 18) * (kCK * 16) + chn * 16 + (real1 ? trl * TPR + pc : 14 + trl);
    }
    // phase 2 positions (source: Y, one tile row)
    int rpos2, vpos2, half2;
    bool real2, act2 = true;
    if constexpr (WD == 56) {                            // (channel 0..7, half 0..1) x 14 tiles
        const int chn = row16 & 7;
        half2 = __builtin_amdgcn_readfirstlane(row16 >> 3);
        real2 = px < TPR;
        rpos2 = chn * kY + 4 * (real2 ? px : TPR - 1);
        vpos2 = (half2 * 18) * (kCK * 16) + chn * 16 + px;
    } else {                                             // waves 0, 1: (channel pair 0..3, half) x (two channels of 7 tiles + 1 idle lane each)
        const int cp = row16 & 3, pc = px & 7, chn = cp * 2 + (px >> 3);
        half2 = __builtin_amdgcn_readfirstlane((row16 >> 2) & 1);
        act2 = __builtin_amdgcn_readfirstlane(row16 >> 3) == 0;
        real2 = pc < TPR;
        rpos2 = chn * kY + 4 * (real2 ? pc : TPR - 1);
        vpos2 = (half2 * 18) * (kCK * 16) + chn * 16 + (real2 ? pc : 8);
    }

    // =========================================================================================== phase 1: conv1 on two tile rows
    f32x4 bq[NLD];                                       // B fragments of the 9 points x NBLK blocks; each is re-requested for the next chunk behind its cluster
    {
        f32x4 acc[9][NB][MT1];
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int m = 0; m < MT1; ++m) acc[p][n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        issue_raw(0);
        issue_raw(1);
#pragma unroll
        for (int g = 0; g < NLD; ++g) bq[g] = load_u(u1_rsrc, 0, g);
        if constexpr (NLD == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");       // the raw rows (requested before the weight loads) have landed
        else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        __syncthreads();
        GRK_BB_STAMP(tk1);
        GRK_BB_PHASE(0, tk0, tk1);
        {
            Tf t;
            tf_read(t, raw + rpos1);
            tf_halo(t, real1);
            if constexpr (WD == 56) {
                tf_rows(t, 0); tf_cols(t, V + vpos1, kCK * 32);
                tf_rows(t, 1); tf_cols(t, V + vpos1 + 18 * (kCK * 32), kCK * 32);
            } else {
                tf_rows(t, half1); tf_cols(t, V + vpos1, kCK * 16);
            }
        }
        GRK_BB_STAMP(tk2);
        GRK_BB_PHASE(1, tk1, tk2);
        float av[2][6 * MT1];
        auto load_a = [&](int buf, int c, int set) {     // A fragments of cluster c: points 3c .. 3c+2 of this wave, both k-steps, MT1 row tiles
#pragma unroll
            for (int k = 0; k < 6; ++k)
#pragma unroll
                for (int m = 0; m < MT1; ++m) {
                    const int p = wave * 9 + 3 * c + (k >> 1), ks = k & 1;
                    av[set][k * MT1 + m] = V[buf * kV1 + p * (kCK * 16 * MT1) + (ks * 4 + lq) * (16 * MT1) + m * 16 + l15];
                }
        };
        // mode 0: transform of chunk `next` + its B fragments + the raw rows of next + 1; mode 1: last chunk, request conv2's first B fragments
        auto chunk = [&](int buf, int mode, int next) {
            Tf t;
            const float* rp = raw + (next & 1) * (kCK * kRaw1) + rpos1;
            float* vp = V + (next & 1) * kV1 + vpos1;
            load_a(buf, 0, 0);
            if (mode == 0 && next + 1 < nchunks) issue_raw(next + 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int k = 0; k < 6 * MT1; ++k) landed(av[c & 1][k]);
                if (mode == 0) {
                    if constexpr (WD == 56) {
                        if (c == 1) { tf_halo(t, real1); tf_rows(t, 0); tf_cols(t, vp, kCK * 32); }
                        if (c == 2) { tf_rows(t, 1); tf_cols(t, vp + 18 * (kCK * 32), kCK * 32); }
                    } else {
                        if (c == 1) { tf_halo(t, real1); tf_rows(t, half1); }
                        if (c == 2) tf_cols(t, vp, kCK * 16);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (c < 2) load_a(buf, c + 1, (c + 1) & 1);
                if (c > 0) {
#pragma unroll
                    for (int k = 0; k < NLD / 3; ++k) {
                        const int g = (NLD / 3) * (c - 1) + k;
                        bq[g] = mode == 0 ? load_u(u1_rsrc, next, g) : load_u(u2_rsrc, 0, g);
                    }
                }
                if (mode == 0 && c == 0) tf_read(t, rp);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 6; ++k)
#pragma unroll
                    for (int m = 0; m < MT1; ++m)
#pragma unroll
                        for (int n = 0; n < NB; ++n) {
                            const int pi = 3 * c + (k >> 1);
                            const float bfr = bq[pi * NBLK + (n >> 1)][(k & 1) * 2 + (n & 1)];
                            acc[pi][n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c & 1][k * MT1 + m], bfr, acc[pi][n][m], 0, 0, 0);
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int k = 0; k < NLD / 3; ++k) {
                const int g = 2 * (NLD / 3) + k;
                bq[g] = mode == 0 ? load_u(u1_rsrc, next, g) : load_u(u2_rsrc, 0, g);
            }
        };
        for (int ch = 0; ch + 1 < nchunks; ++ch) {
            // this wave's share of raw(ch+1) has landed: it was requested before the weight loads of the previous iteration, which may stay
            // in flight (loads return in order)
            if constexpr (NLD == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
            __syncthreads();
            chunk(ch & 1, 0, ch + 1);
        }
        __syncthreads();
        chunk((nchunks - 1) & 1, 1, 0);
        GRK_BB_STAMP(tk3);
        GRK_BB_PHASE(2, tk2, tk3);

        // ---- epilogue 1: A^T M A + b1, ReLU -> Y (zeros for rows outside the image: conv2's padding)
#pragma unroll
        for (int ps = 0; ps < NB * MT1; ++ps) {
            const int nt = ps / MT1, mt = ps - nt * MT1;
            __syncthreads();
#pragma unroll
            for (int pi = 0; pi < 9; ++pi) {
                f32x4 v = acc[pi][0][0];
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int m = 0; m < MT1; ++m)
                        if (n * MT1 + m == ps) v = acc[pi][n][m];
                *reinterpret_cast<f32x4*>(Mx + ((wave * 9 + pi) * 16 + l15) * kMrow + lq * 4) = v;
            }
            __syncthreads();
            if (tid < 14 * 16) {
                const int c = tid / 14, t = tid - c * 14;
                const int tro = WD == 56 ? mt : t / TPR, tx = WD == 56 ? t : t - (t / TPR) * TPR;
                const int co = nt * 16 + c;
                float s[4][6];
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
                const float b = a.b1[co];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int yr = 4 * tro + i, irow = 4 * r - 1 + yr;        // Y row / image row
                    if (yr < 6) {
                        const float* q = s[i];
                        const float p12 = q[1] + q[2], m12 = q[1] - q[2], p34 = q[3] + q[4], m34 = q[3] - q[4];
                        f32x4 y = f32x4{q[0] + p12 + p34 + b, fmaf(2.f, m34, m12) + b, fmaf(4.f, p34, p12) + b, fmaf(8.f, m34, m12) + q[5] + b};
                        const bool in_img = irow >= 0 && irow < H;
                        y[0] = in_img ? fmaxf(y[0], 0.f) : 0.f; y[1] = in_img ? fmaxf(y[1], 0.f) : 0.f;
                        y[2] = in_img ? fmaxf(y[2], 0.f) : 0.f; y[3] = in_img ? fmaxf(y[3], 0.f) : 0.f;
                        *reinterpret_cast<f32x4*>(Y + co * kY + yr * WD + 4 * tx) = y;
                    }
                }
            }
        }
        __syncthreads();                                 // Y complete, the staging area free again
        GRK_BB_STAMP(tk4);
        GRK_BB_PHASE(3, tk3, tk4);
    }

    // =========================================================================================== phase 2: conv2 on one tile row, from Y
    {
        f32x4 acc[9][NB];
#pragma unroll
        for (int p = 0; p < 9; ++p)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[p][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        GRK_BB_STAMP(tk5);
        if (act2) {
            Tf t;
            tf_read(t, Y + rpos2);
            tf_halo(t, real2);
            tf_rows(t, half2);
            tf_cols(t, V + vpos2, kCK * 16);
        }
        GRK_BB_STAMP(tk6);
        GRK_BB_PHASE(4, tk5, tk6);
        float av[2][6];
        auto load_a = [&](int buf, int c, int set) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int p = wave * 9 + 3 * c + (k >> 1), ks = k & 1;
                av[set][k] = V[buf * kV2 + p * (kCK * 16) + (ks * 4 + lq) * 16 + l15];
            }
        };
        auto chunk = [&](int buf, bool with_transform, int next) {
            Tf t;
            const float* rp = Y + next * (kCK * kY) + rpos2;
            float* vp = V + (next & 1) * kV2 + vpos2;
            const bool tfm = with_transform && act2;
            load_a(buf, 0, 0);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int k = 0; k < 6; ++k) landed(av[c & 1][k]);
                if (tfm) {
                    if (c == 1) { tf_halo(t, real2); tf_rows(t, half2); }
                    if (c == 2) tf_cols(t, vp, kCK * 16);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (c < 2) load_a(buf, c + 1, (c + 1) & 1);
                if (with_transform && c > 0) {
#pragma unroll
                    for (int k = 0; k < NLD / 3; ++k) bq[(NLD / 3) * (c - 1) + k] = load_u(u2_rsrc, next, (NLD / 3) * (c - 1) + k);
                }
                if (tfm && c == 0) tf_read(t, rp);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 6; ++k)
#pragma unroll
                    for (int n = 0; n < NB; ++n) {
                        const int pi = 3 * c + (k >> 1);
                        const float bfr = bq[pi * NBLK + (n >> 1)][(k & 1) * 2 + (n & 1)];
                        acc[pi][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c & 1][k], bfr, acc[pi][n], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (with_transform) {
#pragma unroll
                for (int k = 0; k < NLD / 3; ++k) bq[2 * (NLD / 3) + k] = load_u(u2_rsrc, next, 2 * (NLD / 3) + k);
            }
        };
        for (int ch = 0; ch + 1 < nchunks; ++ch) {
            __syncthreads();
            chunk(ch & 1, true, ch + 1);
        }
        __syncthreads();
        chunk((nchunks - 1) & 1, false, 0);
        GRK_BB_STAMP(tk7);
        GRK_BB_PHASE(5, tk6, tk7);

        // ---- epilogue 2: A^T M A + b2 + x, ReLU -> out
        constexpr int NT2 = TPR;                          // tiles of the output tile row
#pragma unroll
        for (int nt = 0; nt < NB; ++nt) {
            __syncthreads();
#pragma unroll
            for (int pi = 0; pi < 9; ++pi) {
                f32x4 v = acc[pi][0];
#pragma unroll
                for (int n = 1; n < NB; ++n)
                    if (n == nt) v = acc[pi][n];
                *reinterpret_cast<f32x4*>(Mx + ((wave * 9 + pi) * 16 + l15) * kMrow + lq * 4) = v;
            }
            __syncthreads();
            if (tid < NT2 * 16) {
                const int c = tid / NT2, t = tid - c * NT2;
                const int co = nt * 16 + c, orow = 4 * r;
                float s[4][6];
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
                const float b = a.b2[co];
                const float* xres = inb + (size_t)co * HW + orow * WD + 4 * t;
                const size_t obase = ((size_t)img * a.out_ctot + a.out_coff + co) * HW + orow * WD + 4 * t;
                f32x4 xr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) xr[i] = *reinterpret_cast<const f32x4*>(xres + i * WD);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float* q = s[i];
                    const float p12 = q[1] + q[2], m12 = q[1] - q[2], p34 = q[3] + q[4], m34 = q[3] - q[4];
                    f32x4 y = f32x4{q[0] + p12 + p34 + b, fmaf(2.f, m34, m12) + b, fmaf(4.f, p34, p12) + b, fmaf(8.f, m34, m12) + q[5] + b};
                    y += xr[i];
                    y[0] = fmaxf(y[0], 0.f); y[1] = fmaxf(y[1], 0.f); y[2] = fmaxf(y[2], 0.f); y[3] = fmaxf(y[3], 0.f);
                    *reinterpret_cast<f32x4*>(a.out + obase + i * WD) = y;
                }
            }
        }
        GRK_BB_STAMP(tk8);
        GRK_BB_PHASE(6, tk7, tk8);
#ifdef GRNET_ABLATION
        if (threadIdx.x == 0) atomicAdd(&g_phase_bb[7], 1ull);
#endif
    }
}

template <int WD>
__global__ __launch_bounds__(256) void bblock_wino4_f32(const BlockArgs a) { bblock_body<WD>(a); }

}  // namespace

bool bblock_wino4_eligible(int c, int h, int w) { return (c == 32 && h == 56 && w == 56) || (c == 64 && h == 28 && w == 28); }

// a.w1 / a.w2: pack_wino4_weights(.., cin_pad = cout_pad = C, ..) in the 32-channel layout (conv_wino4_blocks(C, W) == 2)
hipError_t launch_bblock_wino4(BlockArgs a, int c, int h, int w, hipStream_t s) {
    static bool attr_done[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(bblock_wino4_f32<56>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Geo<56>::lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(bblock_wino4_f32<28>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Geo<28>::lds_bytes);
        if (e != hipSuccess) return e;
        attr_done[dev] = true;
    }
    if (!bblock_wino4_eligible(c, h, w) || conv_wino4_blocks(c, w) != 2 || a.N < 1) return hipErrorInvalidValue;
    a.gx = a.N * (h / 4);
    a.xcd = a.gx % 8 == 0 && a.gx >= 16 ? 1 : 0;
    const hipError_t e = w == 56 ? launch_k(bblock_wino4_f32<56>, dim3(a.gx), dim3(256), Geo<56>::lds_bytes, s, a)
                                 : launch_k(bblock_wino4_f32<28>, dim3(a.gx), dim3(256), Geo<28>::lds_bytes, s, a);
#ifdef GRNET_ABLATION
    static const bool phases = getenv("GRNET_BB_PHASES") != nullptr;
    if (phases && e == hipSuccess) {
        unsigned long long h[8] = {}, z[8] = {};
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase_bb), sizeof(h));
        hipMemcpyToSymbol(HIP_SYMBOL(g_phase_bb), z, sizeof(z));
        const double n = h[7] ? (double)h[7] : 1.0;
        fprintf(stderr, "[fused block phases] c %d w %d N %d wgs %llu: per WG ticks  start->raw %.0f  transform0 %.0f  loop1 %.0f  epilogue1 %.0f  transform0' %.0f  "
                "loop2 %.0f  epilogue2 %.0f\n", c, w, a.N, h[7], h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n, h[5] / n, h[6] / n);
    }
#endif
    return e;
}

}  // namespace grk
