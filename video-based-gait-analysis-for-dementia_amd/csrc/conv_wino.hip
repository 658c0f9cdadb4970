// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores for the wide 3x3 stride-1 layers on 56x56 maps (upsample heads and
// PARE head: hrnet.py:444-451, pare.py:197-210,388-397 -- Conv2d 3x3 pad 1 + BatchNorm2d(eval) + ReLU, no residual): 42 % of the
// path's multiplies.  Y = A^T [ (G g G^T) . (B^T d B) ] A turns every 2x2 output tile into 16 independent products, so the layer is
// 16 GEMMs  M_p[tile][cout] = sum_cin V_p[tile][cin] U_p[cin][cout]  with 4 multiplies per output instead of 9 (2.25x fewer MFMAs).
// The filter transform U = G g G^T is applied to the BN-folded weights once at load, in fp64.  Same fp32 operands, fp32 accumulation;
// the sums are re-associated (transform adds before the products), so results agree with the direct kernel to ~1e-6 of the output
// scale, not bit for bit -- inside the 1e-3 bar by three orders of magnitude and covered by the same parity tests.
//
// One workgroup (4 waves): one image, 2 tile rows = 56 tiles (output rows 4r .. 4r+3, all 56 columns; padded to 64 = 4 MFMA row
// tiles), 64 (or 32) output channels, ALL 16 transform points -- wave w owns points 4w .. 4w+3, i.e. 4 x (4 x 4) accumulator tiles = 256
// accumulation registers.  Per chunk of 8 input channels:
//   * the 6 input rows of the chunk (contiguous in the NCHW plane, 16-byte aligned: 84 units per channel) and the chunk's transformed
//     weights [16][8][64] arrive by LDS-DMA (both double-buffered, requested a whole iteration ahead);
//   * every thread transforms (tile, channel) patches: 16 LDS reads, 32 adds, 16 LDS writes into V[point][channel][tile];
//   * 128 MFMAs per wave: per point and k-step 4 V fragments + 4 U fragments feed 16 MFMAs.
// Epilogue in four passes of 16 channels: accumulators -> LDS [point][tile][channel], inverse transform A^T M A (24 adds per 2x2
// tile), + folded-BN bias, ReLU, two 8-byte stores per thread with the lanes walking a row of the image.
#include "kernels.h"

namespace grk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define GRNET_GLOBAL_AS __attribute__((address_space(1)))
#define GRNET_LDS_AS __attribute__((address_space(3)))

namespace {

constexpr int kWCK = 8;                 // input channels per chunk
constexpr int kWTiles = 56;             // real tiles per workgroup (2 tile rows of 28)
constexpr int kWVT = 72;                // V row stride (tiles): 72 % 32 = 8 -> the four k-rows of an A fragment fall on disjoint bank pairs
constexpr int kWRaw = 6 * 56;           // raw floats per channel: 6 input rows
template <int NB> constexpr int kWUf = 16 * kWCK * NB * 16;     // floats of one transformed-weight chunk (NB 16-channel blocks per workgroup)
constexpr int kWV = 16 * kWCK * kWVT;
constexpr int kWMrow = 17;              // epilogue: [point][tile][16 channels + 1]
template <int NB> constexpr size_t kWinoLdsB = sizeof(float) * (2 * kWCK * kWRaw + 2 * kWUf<NB> + 2 * kWV);     // NB = 4: 160 768 B of the 160 KB
static_assert(sizeof(float) * 16 * 64 * kWMrow <= kWinoLdsB<2>, "the epilogue tile reuses the staging area");

__device__ __forceinline__ void dma16(const float* src, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const GRNET_GLOBAL_AS void*)src, (GRNET_LDS_AS void*)lds_wave_base, 16, 0, 0);
}

// NB: 16-channel blocks per workgroup -- 4 (64 output channels) or 2 (the 32-channel layers: 56x56 branch of the HR modules, transition1)
template <int NB>
__global__ __launch_bounds__(256) void conv_wino_f32(const ConvArgs a) {
    constexpr int TC = NB * 16, kWU = kWUf<NB>, UPR = TC / 4, NUI = (128 * UPR) / 256;
    extern __shared__ __align__(16) float smem[];
    float* raw = smem;                                  // [2][8][336]
    float* U = raw + 2 * kWCK * kWRaw;                  // [2][16][8][64]   (16-byte units XOR-swizzled by channel parity)
    float* V = U + 2 * kWU;                             // [2][16][8][72]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;

    // block -> (image, tile-row group, channel block); XCD-aware like conv_kernels.hip: the channel blocks of one input tile are
    // consecutive blocks of ONE XCD (ids congruent mod 8) and an XCD owns a contiguous range of tiles (a speed heuristic only)
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
    const int groups = a.H >> 2;                         // tile-row groups per image (14)
    const int img = bx / groups, r = bx - img * groups, co0 = by * TC;
    const int HW = a.H * a.W;
    const float* inb = a.in + ((size_t)img * a.in_ctot + a.in_coff) * HW;
    const int g0 = (4 * r - 1) * 56;                     // plane index of raw[.][0]

    // chunk-invariant DMA source offsets
    int uoff[NUI];                                       // transformed weights: 128 rows of UPR 16-byte units per chunk, NUI per thread
#pragma unroll
    for (int i = 0; i < NUI; ++i) {
        const int u = i * 256 + tid, row = u / UPR, j = u - row * UPR, p = row >> 3, ch = row & 7;
        uoff[i] = (p * a.CinPad + ch) * a.CoutPad + co0 + 4 * (j ^ ((ch & 1) << 2));
    }
    int roff[3];                                         // raw rows: 672 units per chunk; -1 = no unit, -2 = outside the image (zeros)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = i * 256 + tid, ch = u / 84, k = u - ch * 84, gi = g0 + 4 * k;
        roff[i] = u < kWCK * 84 ? ((gi >= 0 && gi < HW) ? ch * HW + gi : -2) : -1;
    }
    auto issue_u = [&](int chunk, int buf) {
        const float* src = a.w + (size_t)chunk * kWCK * a.CoutPad;
        float* dst = U + buf * kWU;
#pragma unroll
        for (int i = 0; i < NUI; ++i) dma16(src + uoff[i], dst + (i * 256 + wave * 64) * 4);
    };
    auto issue_raw = [&](int chunk) {
        const float* src = inb + (size_t)chunk * kWCK * HW;
        float* dst = raw + (chunk & 1) * (kWCK * kWRaw);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (roff[i] != -1) dma16(roff[i] >= 0 ? src + roff[i] : a.zeros, dst + (i * 256 + wave * 64) * 4);
    };
    // input transform of chunk c (raw[c & 1] -> V[c & 1]): V[p][ch][t] = (B^T d B)[p], thread -> (tile t, channel ch) pairs tid and tid + 256
    auto transform = [&](int c) {
        const float* rawc = raw + (c & 1) * (kWCK * kWRaw);
        float* Vc = V + (c & 1) * kWV;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            // branch-free (the second round has 192 real pairs): lanes without a pair redo pair 0 and write into the padding tiles 56..71
            // of V, so the whole transform is straight-line code the scheduler can interleave with the MFMAs of the current chunk
            const int pr0 = it * 256 + tid;
            const bool real = pr0 < kWTiles * kWCK;
            const int pr = real ? pr0 : 0;
            const int ch = pr / kWTiles, t = pr - ch * kWTiles, ty2 = t >= 28 ? 1 : 0, tx = t - 28 * ty2;
            const float* rp = rawc + ch * kWRaw + (2 * ty2) * 56 + 2 * tx - 1;
            float d[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) d[i][j] = rp[i * 56 + j];
                d[i][0] = tx == 0 ? 0.f : d[i][0];       // column -1 / 56: the zero padding (the flat rows have no column halo)
                d[i][3] = tx == 27 ? 0.f : d[i][3];
            }
            float e[4][4];                               // B^T d : rows (d0-d2, d1+d2, d2-d1, d1-d3)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                e[0][j] = d[0][j] - d[2][j]; e[1][j] = d[1][j] + d[2][j];
                e[2][j] = d[2][j] - d[1][j]; e[3][j] = d[1][j] - d[3][j];
            }
            float* vp = Vc + ch * kWVT + (real ? t : kWTiles + (lane & 15));
#pragma unroll
            for (int i = 0; i < 4; ++i) {                // (B^T d) B : columns (e0-e2, e1+e2, e2-e1, e1-e3)
                vp[(i * 4 + 0) * (kWCK * kWVT)] = e[i][0] - e[i][2];
                vp[(i * 4 + 1) * (kWCK * kWVT)] = e[i][1] + e[i][2];
                vp[(i * 4 + 2) * (kWCK * kWVT)] = e[i][2] - e[i][1];
                vp[(i * 4 + 3) * (kWCK * kWVT)] = e[i][1] - e[i][3];
            }
        }
    };

    f32x4 acc[4][4][NB];                                 // [point of this wave][tile block][channel block]
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[p][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.CinPad / kWCK;
    issue_raw(0);
    issue_u(0, 0);
    if (nchunks > 1) issue_raw(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    transform(0);
    const int va = lq * kWVT + l15;                                         // + (p*8 + 4*ks) * kWVT + mt*16
    int ubn[NB];                                                            // the 16-float block of channel block n in this lane's (swizzled) row
#pragma unroll
    for (int n = 0; n < NB; ++n) ubn[n] = lq * TC + ((n ^ (lq & 1)) * 16) + l15;
    // One barrier per chunk.  At the top of iteration ch: V[ch&1] is complete, U[ch&1] and raw[(ch+1)&1] have landed.  The next chunk's
    // weights and the raw rows of the chunk after it are requested first, then the next chunk is transformed (VALU + LDS) and this
    // chunk's 128 MFMAs per wave run -- the DMA has the whole iteration to land.
    // ---- the chunk loop, scheduled by hand.  One wave per SIMD (316 registers) has nobody to hide its LDS latency or its vector ALU
    // work behind, and a wave issues in order: an instruction overlaps the matrix pipe only if it sits BETWEEN two MFMAs in program
    // order.  So the 128 MFMAs of chunk ch carry, one micro-step behind each: the 8 operand reads of the next MFMA group (two register
    // sets alternate) and the input transform of chunk ch+1 cut into 96 micro-steps (16 patch reads, 16 + 16 adds, 16 V writes, for each of
    // the thread's two (tile, channel) pairs).  sched_barrier(0) after every pair pins that order (the compiler's own scheduler clusters
    // the MFMAs; a sched_group_barrier pipeline description moved the operand reads but left the transform behind the MFMA section).
    // Transform work item of a thread: TWO horizontally adjacent tiles of one channel (14 pairs x 2 tile rows x 8 channels = 224 threads;
    // the other 32 redo pair 0 and write into the padding tiles of V).  The pair's six input columns are read as b32 | b64 | b64 | b32 per
    // row (the 64-bit reads sit on even columns: aligned) and its 16 x 2 outputs leave as 64-bit writes -- 32 LDS instructions per chunk
    // instead of 64, so that with the 8 operand reads of an MFMA group at most 12 LDS operations are pending at any wait and the
    // compiler's s_waitcnt can name a count instead of draining the queue (lgkmcnt is a 4-bit, in-order counter).
    struct Patch { const float* rp; float* vp; bool left, right; };
    auto patch_of = [&](int c) {
        const bool real = tid < 28 * kWCK;
        const int pr = real ? tid : 0;
        const int chn = pr / 28, pair = pr - chn * 28, ty2 = pair >= 14 ? 1 : 0, tx = 2 * (pair - 14 * ty2);
        Patch q;
        q.rp = raw + (c & 1) * (kWCK * kWRaw) + chn * kWRaw + (2 * ty2) * 56 + 2 * tx;           // column 2*tx of the pair's first row
        q.vp = V + (c & 1) * kWV + chn * kWVT + (real ? 28 * ty2 + tx : kWTiles + 2 * (lane & 7));
        q.left = tx == 0; q.right = tx == 26;
        return q;
    };
    float av[2][4], bv[2][NB];
    constexpr int NOP = 4 + NB, NMF = 4 * NB;                               // operand reads / MFMAs per group
    auto load_group = [&](int buf, int g, int set, int part) {              // part 0..7: one of the 8 operand reads of MFMA group g = (pi, ks)
        const int p = wave * 4 + (g >> 1), ks = g & 1;
        if (part < 4) av[set][part] = V[buf * kWV + va + (p * kWCK + 4 * ks) * kWVT + part * 16];
        else bv[set][part - 4] = U[buf * kWU + (p * kWCK + 4 * ks) * TC + ubn[part - 4]];
    };
    auto chunk = [&](int buf, bool with_transform, int next) {
        const Patch q = patch_of(next);
        float d[4][6], e[4][6];
#pragma unroll
        for (int part = 0; part < NOP; ++part) load_group(buf, 0, 0, part);
        if (with_transform) {                                                // the DMA requests of the coming chunks go out under the LDS latency
            issue_u(next, buf ^ 1);                                          // of the first operand reads (nothing else can cover it: the barrier
            if (next + 1 < nchunks) issue_raw(next + 1);                     // above is where V of this chunk became complete)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
#pragma unroll
            for (int k = 0; k < NMF; ++k) {
                const int m = k / NB, n = k % NB, pi = g >> 1;
                acc[pi][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][m], bv[g & 1][n], acc[pi][m][n], 0, 0, 0);
                if (g < 7 && k < NOP) load_group(buf, g + 1, (g + 1) & 1, k);
                // hipcc drains the whole LDS queue (lgkmcnt(0)) in front of every MFMA group, whatever is pending: all LDS traffic of a group
                // -- the next group's 8 operand reads and 7 transform micro-steps (7 x 8 = the 56 of a chunk) -- therefore sits in the
                // group's FIRST eight slots, and the eight MFMAs behind them (256 cycles) let it land before the next drain
                if (with_transform && k < 7) {
                    const int st = g * 7 + k;
                    if (st < 16) {                                           // loads: row i = st / 4, piece st % 4 of (c-1 | c0 c1 | c2 c3 | c4)
                        const int i = st >> 2, pc = st & 3;
                        const float* r = q.rp + i * 56;
                        if (pc == 0) d[i][0] = r[-1];
                        else if (pc == 1) { const f32x2 v = *reinterpret_cast<const f32x2*>(r); d[i][1] = v[0]; d[i][2] = v[1]; }
                        else if (pc == 2) { const f32x2 v = *reinterpret_cast<const f32x2*>(r + 2); d[i][3] = v[0]; d[i][4] = v[1]; }
                        else d[i][5] = r[4];
                    } else if (st < 40) {                                    // B^T d per column: rows (d0-d2, d1+d2, d2-d1, d1-d3); the padding
                        const int kk = st - 16, j = kk >> 2, i = kk & 3;       // columns -1 / 56 are zeroed here (the row transform is linear)
                        float ev = i == 0 ? d[0][j] - d[2][j] : i == 1 ? d[1][j] + d[2][j] : i == 2 ? d[2][j] - d[1][j] : d[1][j] - d[3][j];
                        if (j == 0) ev = q.left ? 0.f : ev;
                        if (j == 5) ev = q.right ? 0.f : ev;
                        e[i][j] = ev;
                    } else if (st < 56) {                                    // (B^T d) B for both tiles of the pair, one 64-bit write per point
                        const int kk = st - 40, i = kk >> 2, j = kk & 3;
                        const float vl = j == 0 ? e[i][0] - e[i][2] : j == 1 ? e[i][1] + e[i][2] : j == 2 ? e[i][2] - e[i][1] : e[i][1] - e[i][3];
                        const float vr = j == 0 ? e[i][2] - e[i][4] : j == 1 ? e[i][3] + e[i][4] : j == 2 ? e[i][4] - e[i][3] : e[i][3] - e[i][5];
                        *reinterpret_cast<f32x2*>(q.vp + (i * 4 + j) * (kWCK * kWVT)) = f32x2{vl, vr};
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    for (int ch = 0; ch + 1 < nchunks; ++ch) {
        const int buf = ch & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // this wave's shares of U(ch) and raw(ch+1) have landed
        __syncthreads();                                                     // ... everybody's; everybody is past MFMA(ch-1) and transform(ch)
        chunk(buf, true, ch + 1);                                            // requests U(ch+1) and raw(ch+2) [into raw[ch&1], which transform(ch) has finished reading]
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    chunk((nchunks - 1) & 1, false, 0);

    // ---- epilogue: inverse transform, + bias, ReLU, store; 16 output channels per pass
    float* Mx = smem;                                                        // [16][64][17]
    const bool has_add = a.n_add == 1;                                       // the BasicBlock residual (same shape as the output)
    for (int nt = 0; nt < NB; ++nt) {
        __syncthreads();                                                     // staging area / previous pass no longer read
#pragma unroll
        for (int pi = 0; pi < 4; ++pi)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                f32x4 v;
                // static register index: select the pass's channel block without dynamic indexing of the accumulator array
                if constexpr (NB == 4) v = nt == 0 ? acc[pi][m][0] : nt == 1 ? acc[pi][m][1] : nt == 2 ? acc[pi][m][2] : acc[pi][m][3];
                else v = nt == 0 ? acc[pi][m][0] : acc[pi][m][1];
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) Mx[((wave * 4 + pi) * 64 + m * 16 + lq * 4 + rr) * kWMrow + l15] = v[rr];
            }
        __syncthreads();
        for (int pr = tid; pr < kWTiles * 16; pr += 256) {
            const int c = pr / kWTiles, t = pr - c * kWTiles, ty2 = t >= 28 ? 1 : 0, tx = t - 28 * ty2;
            const int co = co0 + nt * 16 + c;
            if (co >= a.Cout) continue;
            float m[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) m[p] = Mx[(p * 64 + t) * kWMrow + c];
            float s[4], q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] = m[j] + m[4 + j] + m[8 + j]; q[j] = m[4 + j] - m[8 + j] - m[12 + j]; }
            const float b = a.bias[co];
            float y00 = s[0] + s[1] + s[2] + b, y01 = s[1] - s[2] - s[3] + b, y10 = q[0] + q[1] + q[2] + b, y11 = q[1] - q[2] - q[3] + b;
            if (has_add) {
                const float* ap = a.add[0] + ((size_t)img * a.add_ctot[0] + a.add_coff[0] + co) * HW + (4 * r + 2 * ty2) * 56 + 2 * tx;
                const f32x2 r0 = *reinterpret_cast<const f32x2*>(ap), r1 = *reinterpret_cast<const f32x2*>(ap + 56);
                y00 += r0[0]; y01 += r0[1]; y10 += r1[0]; y11 += r1[1];
            }
            if (a.relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
            float* op = a.out + ((size_t)img * a.out_ctot + a.out_coff + co) * HW + (4 * r + 2 * ty2) * 56 + 2 * tx;
            *reinterpret_cast<f32x2*>(op) = f32x2{y00, y01};
            *reinterpret_cast<f32x2*>(op + 56) = f32x2{y10, y11};
        }
    }
}

}  // namespace

bool conv_wino_eligible(int cin, int cout, int ks, int stride, int h, int w, int n_add) {
    return ks == 3 && stride == 1 && h == 56 && w == 56 && n_add <= 1 && cin % kWCK == 0 && cout % 32 == 0 && cin >= 32;
}

// a.w: transformed weights [16][CinPad][CoutPad] (pack_wino_weights), CinPad % 8 == 0, CoutPad % 64 == 0
hipError_t launch_conv_wino(ConvArgs a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_f32<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLdsB<4>);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_f32<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLdsB<2>);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nb = a.Cout % 64 == 0 ? 4 : 2;
    if (!conv_wino_eligible(a.Cin, a.Cout, a.ks, a.stride, a.H, a.W, a.n_add) || a.CinPad % kWCK != 0 || a.CoutPad % (nb * 16) != 0) return hipErrorInvalidValue;
    if (a.n_add == 1 && a.add_shift[0] != 0) return hipErrorInvalidValue;
    a.gx = a.N * (a.H >> 2);
    a.gy = a.CoutPad / (nb * 16);
    a.gx8 = (a.gx + 7) / 8;
    a.xcd = a.gx >= 16 ? 1 : 0;
    const dim3 grid((a.xcd ? a.gx8 * 8 : a.gx) * a.gy);
    if (nb == 4) return launch_k(conv_wino_f32<4>, grid, dim3(256), kWinoLdsB<4>, s, a);
    return launch_k(conv_wino_f32<2>, grid, dim3(256), kWinoLdsB<2>, s, a);
}

// U = G g G^T per (cout, cin) in fp64 -> [16][cin_pad][cout_pad] fp32; w: (cout, cin, 3, 3) folded weights (double)
void pack_wino_weights(const double* w, int cout, int cin, int cin_pad, int cout_pad, float* out) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (size_t i = 0; i < (size_t)16 * cin_pad * cout_pad; ++i) out[i] = 0.f;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            const double* g = w + ((size_t)co * cin + ci) * 9;
            double t[4][3];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[j] + G[i][1] * g[3 + j] + G[i][2] * g[6 + j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                    out[((size_t)(i * 4 + j) * cin_pad + ci) * cout_pad + co] = (float)u;
                }
        }
}

}  // namespace grk
