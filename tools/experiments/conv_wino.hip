// Winograd F(2x2, 3x3) convolution on the fp32 matrix cores for the 3x3 stride-1 layers on 56x56 maps (layer1, HR branch 0,
// transition1, upsample heads, PARE head: hrnet.py:54-57,444-451, pare.py:197-210,388-397 -- Conv2d 3x3 pad 1 + BatchNorm2d(eval)
// [+ residual] + ReLU) and the wide ones on 28x28 maps (upsample heads): 64 % of the path's multiplies.
// Y = A^T [ (G g G^T) . (B^T d B) ] A turns every 2x2 output tile into 16 independent products, so the layer is 16 GEMMs
// M_p[tile][cout] = sum_cin V_p[tile][cin] U_p[cin][cout]  with 4 multiplies per output instead of 9 (2.25x fewer MFMAs).
// The filter transform U = G g G^T is applied to the BN-folded weights once at load, in fp64.  Same fp32 operands, fp32 accumulation;
// the sums are re-associated (transform adds before the products), so results agree with the direct kernel to ~1e-6 of the output
// scale, not bit for bit -- inside the 1e-3 bar by three orders of magnitude and covered by the same parity tests.
//
// One workgroup (4 waves, one per SIMD): one image, 56 tiles (2 tile rows of 28, or 4 of 14 on a 28-wide map; padded to 64 = 4 MFMA
// row tiles), 64 (or 32) output channels, ALL 16 transform points -- wave w owns points 4w .. 4w+3, i.e. 4 x (4 x 4) accumulator
// tiles = 256 accumulation registers.  Per chunk of 8 input channels:
//   * the chunk's input rows (contiguous in the NCHW plane, 16-byte aligned) arrive by LDS-DMA, two buffers, a chunk ahead;
//   * every thread transforms TWO adjacent tiles of one channel: 4 16-byte LDS reads, DPP for the neighbour columns, 56 adds,
//     16 8-byte LDS writes into V[point][channel][16][4];
//   * 128 MFMAs per wave: per (point, k-step) ONE 16-byte LDS read gives the 4 A fragments, ONE 16-byte load straight from L2
//     (requested a chunk ahead) the 4 B fragments, feeding 16 MFMAs.
// The loop is built around one fact of gfx950 (see "the chunk loop" below): the fp32 MFMA shares the SIMD's FP32 lanes with the
// vector ALU, so every other instruction in the loop is matrix-pipe time -- few of them, clustered in front of each MFMA group.
// Epilogue in passes of 16 channels: accumulators -> LDS [point][channel][68], inverse transform A^T M A (24 adds per 2x2 tile),
// + folded-BN bias, + residual, ReLU, two 8-byte stores per thread with the lanes walking a row of the image.
// DESIGN.md 4.1c has the measurements (direct kernel / first version / now per layer shape, ablations of the loop).
#include "kernels.h"

namespace grk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define GRNET_GLOBAL_AS __attribute__((address_space(1)))
#define GRNET_LDS_AS __attribute__((address_space(3)))

namespace {

constexpr int kWCK = 8;                 // input channels per chunk
constexpr int kWTiles = 56;             // real tiles per workgroup (2 tile rows of 28, or 4 of 14)
constexpr int kWRawMax = 6 * 56;        // raw floats per channel: 6 input rows of 56 (10 rows of 28 are fewer)
constexpr int kWV = 16 * kWCK * 64;      // V[point][channel][64 tile slots]
constexpr int kWinoCUs = 256;          // CUs of an MI355X: workgroups of a full round
constexpr int kWMrow = 68;              // epilogue: [point][channel][64 MFMA rows + 4]
constexpr size_t kWinoLdsB = sizeof(float) * (2 * kWCK * kWRawMax + 2 * kWV);     // 87 040 B
static_assert(sizeof(float) * 16 * 16 * kWMrow <= kWinoLdsB, "the epilogue tile reuses the staging area");

// a use of x the compiler cannot move: its wait for the LDS read that produces x lands here (a "v" constraint is a device-side
// thing: inside the __global__ template itself the host pass rejects it and silently drops the kernel's stub)
template <typename T>
__device__ __forceinline__ void landed(T& x) { asm volatile("" : "+v"(x)); }

// NB: 16-channel blocks per workgroup -- 4 (64 output channels) or 2 (the 32-channel layers: 56x56 branch of the HR modules, transition1)
// (the body is a __device__ function: the host pass type-checks the body of a __global__ template, rejects device-only constructs
// in it without a diagnostic and then emits no launch stub -- the library fails to load with an undefined kernel symbol)
// WD: map width, 56 or 28.  A workgroup's 56 tiles are 2 tile rows of 28 (WD = 56: output rows 4r .. 4r+3, 6 input rows) or 4 tile rows
// of 14 (WD = 28: output rows 8r .. 8r+7, 10 input rows; 14 tile rows per image = 3.5 groups, the last group's lower half lies below
// the image: its input rows are zeros, its outputs are not stored).
template <int NB, int WD, int ABL>
__device__ __forceinline__ void conv_wino_body(const ConvArgs& a) {
    constexpr int TC = NB * 16;
    // The 64 convolutions of the 56x56 HR branch are the longest dependency chain of stages 2-4 (eight per module, one after the other,
    // while the other branches' shorter kernels share the SIMDs): their waves ask the issue arbiter for precedence.  +1 % on the forward.
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio >= 2) __builtin_amdgcn_s_setprio(3);
    constexpr int TRW = WD / 2, TRG = 56 / TRW, kWRaw = (2 * TRG + 2) * WD, UPC = kWRaw / 4;   // tiles per tile row, tile rows per workgroup, raw floats / 16-byte units per channel
    static_assert(WD == 56 || WD == 28, "tile geometry");
    extern __shared__ __align__(16) float smem[];
    float* raw = smem;                                  // [2][8][kWRaw]
    float* V = raw + 2 * kWCK * kWRawMax;                  // [2][16][8][16][4]   (a lane's four tile blocks contiguous)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;

    // block -> (image, tile-row group, channel block).  Tile ids are dense.  With gx % 8 == 0 the order is XCD-aware like
    // conv_kernels.hip: the channel blocks of one input tile are consecutive ids of ONE XCD (ids congruent mod 8) and an XCD owns a
    // contiguous range of input tiles (a speed heuristic only).  a.wsplit: two half-workgroups (32 channels each) per tile id.
    const int id = a.blk0 + (a.wsplit ? (int)(blockIdx.x >> 1) : (int)blockIdx.x), half = a.wsplit ? (int)(blockIdx.x & 1) : 0;
    int bx, by;
    if (a.xcd) {
        const int j = id >> 3, x = id & 7, q = j / a.gy;
        by = j - q * a.gy;
        bx = x * (a.gx >> 3) + q;
    } else {
        bx = id / a.gy;
        by = id - bx * a.gy;
    }
    const int groups = ((a.H >> 1) + TRG - 1) / TRG;     // tile-row groups per image (14 or 4)
    const int img = bx / groups, r = bx - img * groups;
    const int co0 = a.wsplit ? by * 64 : by * TC;           // first channel of the weight block; channel n*16 + l sits at l*cstr + n
    const int cstr = a.wsplit ? 4 : NB, nb0 = 2 * half;
    const int HW = a.H * a.W;
    const float* inb = a.in + ((size_t)img * a.in_ctot + a.in_coff) * HW;
    const int g0 = (2 * TRG * r - 1) * WD;               // plane index of raw[.][0]

    // ---- LDS-DMA through buffer descriptors: per-lane byte offsets are chunk-invariant (a VGPR each), the chunk moves the scalar
    // offset -- no vector-ALU address arithmetic in the loop (on gfx950 every vector-ALU instruction is matrix-pipe time, see below)
    const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, (short)0, 16 * a.CinPad * a.CoutPad * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, a.Cin * HW * 4, 0x00020000);
    // transformed weights: a wave's MFMA B fragments are ITS four points' rows -- no other wave reads them, so they go from L2 straight
    // into registers (one 16-byte load per lane and MFMA group, requested a whole chunk ahead), not through the LDS
    typedef float bfrag __attribute__((ext_vector_type(NB)));
    const int ub = ((wave * 4 * a.CinPad + lq) * a.CoutPad + co0 + l15 * cstr + nb0) * 4;
    const int u_point = a.CinPad * a.CoutPad * 4, u_kstep = 4 * a.CoutPad * 4, u_chunk = kWCK * a.CoutPad * 4;
    auto load_u = [&](int chunk, int g) -> bfrag {                           // group g = (point wave*4 + g/2, k-step g%2)
        const int soff = chunk * u_chunk + (g >> 1) * u_point + (g & 1) * u_kstep;
        if constexpr (NB == 4) return __builtin_bit_cast(bfrag, __builtin_amdgcn_raw_buffer_load_b128(u_rsrc, ub, soff, 0));
        else return __builtin_bit_cast(bfrag, __builtin_amdgcn_raw_buffer_load_b64(u_rsrc, ub, soff, 0));
    };
    int roff[3];                                         // raw rows: 672 units per chunk; -1 = no unit or a row outside the image
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int u = i * 256 + tid, ch = u / UPC, k = u - ch * UPC, gi = g0 + 4 * k;
        const bool unit = u < kWCK * UPC, inside = gi >= 0 && gi < HW;
        roff[i] = unit && inside ? (ch * HW + gi) * 4 : -1;
        if (unit && !inside) {                           // rows above / below the image: zero once in both buffers, the DMA never writes there
            *reinterpret_cast<f32x4*>(raw + u * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(raw + kWCK * kWRaw + u * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    auto issue_raw = [&](int chunk) {
        const int soff = chunk * (kWCK * 4) * HW;
        float* dst = raw + (chunk & 1) * (kWCK * kWRaw);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (roff[i] >= 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r_rsrc, (GRNET_LDS_AS void*)(dst + (i * 256 + wave * 64) * 4), 16, roff[i], soff, 0, 0);
        asm volatile("" ::: "memory");                   // later loads stay behind these requests: the vmcnt(8) waits below count on the order
    };

    // ---- input transform.  A thread owns TWO horizontally adjacent tiles of one channel.  The 256 threads are 16 rows of 16 lanes:
    // row = (channel 0..7, tile row 0..1), lane = tile pair 0..13 (lanes 14, 15 idle: they fill the four padding pairs of V).  The
    // pair's four own input columns 4x .. 4x+3 come as ONE aligned 16-byte LDS read per input row; column 4x-1 is the left neighbour's
    // last column and 4x+4 the right neighbour's first: two DPP row shifts, whose out-of-row zero (bound_ctrl) IS the image's zero
    // padding on the left, an idle lane's zero on the right.  V[point][channel][pair & 15][2 * (pair >> 4) + half]: the MFMA row tile m
    // of a tile is 2 * (pair >> 4) + (tile & 1), its row pair & 15 -- so the four A fragments of a lane are 16 contiguous bytes, and a
    // thread's two results per point one 8-byte write.
    // WD = 28: the 16 lanes are two tile rows of 7 pairs + 1 idle lane each; idle lanes supply the zero on BOTH sides there.
    const int row16 = tid >> 4, px = tid & 15, chn = row16 >> 1, rh = row16 & 1;
    const int pc = WD == 56 ? px : (px & 7), trl = WD == 56 ? rh : 2 * rh + (px >> 3);       // pair column, tile row within the workgroup
    const bool real = pc < TRW / 2;
    const int pairt = real ? trl * (TRW / 2) + pc : 28 + 2 * rh + (WD == 56 ? px - 14 : (px >> 3));
    const int rpos = chn * kWRaw + (2 * trl) * WD + 4 * (real ? pc : TRW / 2 - 1);
    const int vpos = chn * 64 + (pairt & 15) * 4 + 2 * (pairt >> 4);
    struct Tf { float d[4][6]; float e[4][6]; };
    auto tf_read = [&](Tf& t, const float* rp) {        // 4 LDS reads
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(rp + i * WD);
            t.d[i][1] = v[0]; t.d[i][2] = v[1]; t.d[i][3] = v[2]; t.d[i][4] = v[3];
        }
    };
    auto tf_halo = [&](Tf& t) {                         // 12 vector-ALU instructions
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            t.d[i][0] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(WD == 56 || real ? t.d[i][4] : 0.f), 0x111, 0xf, 0xf, true));   // row_shr:1
            t.d[i][5] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(real ? t.d[i][1] : 0.f), 0x101, 0xf, 0xf, true));    // row_shl:1
        }
    };
    auto tf_rows = [&](Tf& t) {                         // B^T d per column: rows (d0-d2, d1+d2, d2-d1, d1-d3); 24
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            t.e[0][j] = t.d[0][j] - t.d[2][j]; t.e[1][j] = t.d[1][j] + t.d[2][j];
            t.e[2][j] = t.d[2][j] - t.d[1][j]; t.e[3][j] = t.d[1][j] - t.d[3][j];
        }
    };
    auto tf_cols = [&](Tf& t, int i, float* vp) {       // (B^T d) B, row i, both tiles of the pair: 8 + 4 LDS writes
        const float* e = t.e[i];
        *reinterpret_cast<f32x2*>(vp + (i * 4 + 0) * 512) = f32x2{e[0] - e[2], e[2] - e[4]};
        *reinterpret_cast<f32x2*>(vp + (i * 4 + 1) * 512) = f32x2{e[1] + e[2], e[3] + e[4]};
        *reinterpret_cast<f32x2*>(vp + (i * 4 + 2) * 512) = f32x2{e[2] - e[1], e[4] - e[3]};
        *reinterpret_cast<f32x2*>(vp + (i * 4 + 3) * 512) = f32x2{e[1] - e[3], e[3] - e[5]};
    };

    f32x4 acc[4][4][NB];                                 // [point of this wave][tile block][channel block]
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[p][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = a.CinPad / kWCK;
    bfrag bq[8];                                                             // B fragments of the 8 MFMA groups; each is re-requested for the
                                                                             // next chunk right behind the group that consumed it
    issue_raw(0);
    if (nchunks > 1) issue_raw(1);
#pragma unroll
    for (int g = 0; g < 8; ++g) bq[g] = load_u(0, g);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                        // the raw rows (requested before the 8 weight loads) have landed
    __syncthreads();
    {
        Tf t;
        tf_read(t, raw + rpos);
        tf_halo(t);
        tf_rows(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) tf_cols(t, i, V + vpos);
    }
    // ---- the chunk loop.  One barrier per chunk: at the top of iteration ch, V[ch&1] is complete and raw[(ch+1)&1] has landed.
    // On gfx950 the fp32 MFMA runs on the SIMD's FP32 lanes (its 64 FLOP/clk/SIMD IS the vector rate): nothing a wave issues between
    // two fp32 MFMAs is free -- tools/micro/mfma_interleave.hip: one v_add_f32 behind every MFMA 33 -> 47 cycles per MFMA, one ds_read_b32
    // 33 -> 43.5, each further vector instruction +4..6, and a second wave per SIMD recovers only part of it (39.6 / 39.7).  The first
    // non-MFMA instruction after an MFMA is the expensive one.  So this loop (a) has few instructions besides its 128 MFMAs per chunk --
    // operand fragments as 16-byte loads (2 per 16 MFMAs), addresses in scalar registers, the transform's halo by DPP -- and (b) keeps
    // them in ONE cluster in front of each group of 16 MFMAs, which then issue back to back.  A cluster consumes only LDS data requested
    // one cluster earlier (a whole MFMA group, 512+ cycles, ago): its single lgkmcnt(0) wait finds the queue empty.
    f32x4 av[2];
    auto load_a = [&](int buf, int g, int set) {                             // A fragments of MFMA group g = (point pi, k-step ks): one LDS read
        const int p = wave * 4 + (g >> 1), ks = g & 1;
        av[set] = *reinterpret_cast<const f32x4*>(V + buf * kWV + p * 512 + (ks * 4 + lq) * 64 + l15 * 4);
    };
    auto chunk = [&](int buf, bool with_transform, int next) {
        Tf t;
        const float* rp = raw + (next & 1) * (kWCK * kWRaw) + rpos;
        float* vp = V + (next & 1) * kWV + vpos;
        if (ABL != 3 && ABL != 5) load_a(buf, 0, 0);
        if (with_transform && ABL != 1 && ABL != 5 && next + 1 < nchunks) issue_raw(next + 1);     // under the latency of that first read
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            // cluster g: (1) everything requested one cluster ago has landed
            landed(av[g & 1]);
            if (with_transform && ABL != 2 && ABL != 5) {
                if (g == 1 && ABL != 7) tf_halo(t);
                if (g == 1 && ABL == 7) { for (int i = 0; i < 4; ++i) { t.d[i][0] = t.d[i][2]; t.d[i][5] = t.d[i][3]; } }
                if (g == 2) tf_rows(t);
                if (g >= 3 && g < 7 && ABL != 6) tf_cols(t, g - 3, vp);
                if (g >= 3 && g < 7 && ABL == 6) { float sk = 0.f; for (int j = 0; j < 6; ++j) sk += t.e[g - 3][j]; if (sk == 12345.f) vp[0] = sk; }
            }
            __builtin_amdgcn_sched_barrier(0);
            // (2) requests: the next group's A fragments, the next chunk's B fragments of the previous group, the next chunk's input rows
            if (g < 7 && ABL != 3 && ABL != 5) load_a(buf, g + 1, (g + 1) & 1);
            if (with_transform && ABL != 1 && ABL != 5 && g > 0) bq[g - 1] = load_u(next, g - 1);
            if (with_transform && ABL != 2 && ABL != 5 && g == 0) tf_read(t, rp);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4 * NB; ++k) {
                const int m = k / NB, n = k % NB, pi = g >> 1;
                acc[pi][m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g & 1][m], bq[g][n], acc[pi][m][n], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (with_transform && ABL != 1 && ABL != 5) bq[7] = load_u(next, 7);
    };
    auto meet = [&](bool last) {
        if (ABL == 4 || ABL == 5) return;
        // this wave's share of raw(ch+1) has landed: it was requested before the 8 weight loads of the previous iteration, which may stay
        // in flight (loads return in order)
        if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __syncthreads();                                                     // ... everybody's; everybody is past MFMA(ch-1) and transform(ch)
    };
    for (int ch = 0; ch + 1 < nchunks; ++ch) {
        meet(false);
        chunk(ch & 1, true, ch + 1);                                         // requests raw(ch+2) [into raw[ch&1], which transform(ch) has finished reading]
    }
    meet(nchunks > 1);
    chunk((nchunks - 1) & 1, false, 0);

    // ---- epilogue: inverse transform, + bias, ReLU, store; 16 output channels per pass
    float* Mx = smem;                                                        // [16 points][16 channels][68]
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
                // the 4 MFMA rows of a lane are consecutive: one 16-byte write (a quarter wave = 16 channels x 16 bytes at a stride of
                // 68 floats: all 64 banks once)
                *reinterpret_cast<f32x4*>(Mx + ((wave * 4 + pi) * 16 + l15) * kWMrow + m * 16 + lq * 4) = v;
            }
        __syncthreads();
        for (int pr = tid; pr < kWTiles * 16; pr += 256) {
            const int c = pr / kWTiles, t = pr - c * kWTiles, tro = t / TRW, tx = t - TRW * tro;
            const int orow = 2 * (TRG * r + tro);                            // output row of the tile's upper pixel pair
            if (WD == 28 && orow >= a.H) continue;                           // the last group's lower half (below the image)
            const int co = co0 + (nb0 + nt) * 16 + c;
            if (co >= a.Cout) continue;
            float m[16];
            const int pairo = t >> 1, ri = (2 * (pairo >> 4) + (t & 1)) * 16 + (pairo & 15);     // the tile's MFMA row (see the V layout)
#pragma unroll
            for (int p = 0; p < 16; ++p) m[p] = Mx[(p * 16 + c) * kWMrow + ri];
            float s[4], q[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] = m[j] + m[4 + j] + m[8 + j]; q[j] = m[4 + j] - m[8 + j] - m[12 + j]; }
            const float b = a.bias[co];
            float y00 = s[0] + s[1] + s[2] + b, y01 = s[1] - s[2] - s[3] + b, y10 = q[0] + q[1] + q[2] + b, y11 = q[1] - q[2] - q[3] + b;
            if (has_add) {
                const float* ap = a.add[0] + ((size_t)img * a.add_ctot[0] + a.add_coff[0] + co) * HW + orow * WD + 2 * tx;
                const f32x2 r0 = *reinterpret_cast<const f32x2*>(ap), r1 = *reinterpret_cast<const f32x2*>(ap + WD);
                y00 += r0[0]; y01 += r0[1]; y10 += r1[0]; y11 += r1[1];
            }
            if (a.relu) { y00 = fmaxf(y00, 0.f); y01 = fmaxf(y01, 0.f); y10 = fmaxf(y10, 0.f); y11 = fmaxf(y11, 0.f); }
            float* op = a.out + ((size_t)img * a.out_ctot + a.out_coff + co) * HW + orow * WD + 2 * tx;
            *reinterpret_cast<f32x2*>(op) = f32x2{y00, y01};
            *reinterpret_cast<f32x2*>(op + WD) = f32x2{y10, y11};
        }
    }
}

template <int NB, int WD = 56, int ABL = 0>            // ABL: timing-only ablations (GRNET_ABLATION builds, tools/wino_micro.py); 0 in the product
__global__ __launch_bounds__(256) void conv_wino_f32(const ConvArgs a) { conv_wino_body<NB, WD, ABL>(a); }

}  // namespace

bool conv_wino_eligible(int cin, int cout, int ks, int stride, int h, int w, int n_add) {
    return ks == 3 && stride == 1 && ((h == 56 && w == 56) || (h == 28 && w == 28)) && n_add <= 1 && cin % kWCK == 0 && cout % 32 == 0 && cin >= 32;
}

template <int WD>
static hipError_t launch_wino_w(ConvArgs a, hipStream_t s, int nb, int* n_launches) {
    if (n_launches) *n_launches = 1;
    constexpr int TRG = 56 / (WD / 2);
    a.gx = a.N * (((a.H >> 1) + TRG - 1) / TRG);
    a.gy = a.CoutPad / (nb * 16);
    a.xcd = a.gx % 8 == 0 && a.gx >= 16 ? 1 : 0;
    a.blk0 = 0;
    a.wsplit = 0;
    const int total = a.gx * a.gy;
    static const int split_env = getenv("GRNET_WINO_SPLIT") ? atoi(getenv("GRNET_WINO_SPLIT")) : 1;
#ifdef GRNET_ABLATION
    if constexpr (WD == 56) {
        if (nb == 4 && a.dbg) {
            const dim3 grid(total);
            auto go = [&](auto kern) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLdsB); return launch_k(kern, grid, dim3(256), kWinoLdsB, s, a); };
            if (a.dbg == 1) return go(conv_wino_f32<4, 56, 1>);
            if (a.dbg == 2) return go(conv_wino_f32<4, 56, 2>);
            if (a.dbg == 3) return go(conv_wino_f32<4, 56, 3>);
            if (a.dbg == 4) return go(conv_wino_f32<4, 56, 4>);
            if (a.dbg == 5) return go(conv_wino_f32<4, 56, 5>);
            if (a.dbg == 6) return go(conv_wino_f32<4, 56, 6>);
            if (a.dbg == 7) return go(conv_wino_f32<4, 56, 7>);
        }
    }
#endif
    if (nb == 2) return launch_k(conv_wino_f32<2, WD>, dim3(total), dim3(256), kWinoLdsB, s, a);
    // One workgroup per CU (kWinoCUs of them): a layer whose last round is at most half full (896 workgroups at 16 frames and 256
    // output channels: 3.5 rounds) runs that round as twice as many HALF workgroups -- the 32-channel kernel on the same packed
    // weights -- so the round costs about 0.6 of a full one instead of 1.
    const int full = total / kWinoCUs * kWinoCUs, rest = total - full;
    if (split_env && full > 0 && rest > 0 && 2 * rest <= kWinoCUs && (!a.xcd || full % 8 == 0)) {
        hipError_t e = launch_k(conv_wino_f32<4, WD>, dim3(full), dim3(256), kWinoLdsB, s, a);
        if (e != hipSuccess) return e;
        a.blk0 = full;
        a.wsplit = 1;
        if (n_launches) *n_launches = 2;
        return launch_k(conv_wino_f32<2, WD>, dim3(2 * rest), dim3(256), kWinoLdsB, s, a);
    }
    return launch_k(conv_wino_f32<4, WD>, dim3(total), dim3(256), kWinoLdsB, s, a);
}

// a.w: transformed weights [16][CinPad][CoutPad] (pack_wino_weights), CinPad % 8 == 0, CoutPad % 64 == 0 (or 32: the 32-channel kernel)
hipError_t launch_conv_wino(ConvArgs a, hipStream_t s, int* n_launches) {
    static bool attr_done[64] = {};                       // per device: function attributes belong to the device's copy of the code object
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    bool& attr_set = attr_done[dev];
    if (!attr_set) {
        hipError_t e = hipSuccess;
        auto set = [&](auto kern) { if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLdsB); };
        set(conv_wino_f32<4, 56>); set(conv_wino_f32<2, 56>); set(conv_wino_f32<4, 28>); set(conv_wino_f32<2, 28>);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int nb = a.Cout % 64 == 0 ? 4 : 2;
    if (!conv_wino_eligible(a.Cin, a.Cout, a.ks, a.stride, a.H, a.W, a.n_add) || a.CinPad % kWCK != 0 || a.CoutPad % (nb * 16) != 0) return hipErrorInvalidValue;
    if (a.n_add == 1 && a.add_shift[0] != 0) return hipErrorInvalidValue;
    return a.W == 56 ? launch_wino_w<56>(a, s, nb, n_launches) : launch_wino_w<28>(a, s, nb, n_launches);
}

// U = G g G^T per (cout, cin) in fp64 -> [16][cin_pad][cout_pad] fp32; w: (cout, cin, 3, 3) folded weights (double)
void pack_wino_weights(const double* w, int cout, int cin, int cin_pad, int cout_pad, float* out) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (size_t i = 0; i < (size_t)16 * cin_pad * cout_pad; ++i) out[i] = 0.f;
    const int nb = cout % 64 == 0 ? 4 : 2, tc = nb * 16;                    // the kernel variant launch_conv_wino picks for this layer
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
            // within a workgroup's tc channels, channel n*16 + l sits at l*nb + n: lane l's nb MFMA B fragments are one LDS read
            const int cpos = (co / tc) * tc + (co % 16) * nb + (co % tc) / 16;
            const double* g = w + ((size_t)co * cin + ci) * 9;
            double t[4][3];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[j] + G[i][1] * g[3 + j] + G[i][2] * g[6 + j];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
                    out[((size_t)(i * 4 + j) * cin_pad + ci) * cout_pad + cpos] = (float)u;
                }
        }
}

}  // namespace grk
