// bf16 path, round 6: the stem pair (hrnet.py:470-476: conv1-BN-ReLU 3 -> 64 s2, conv2-BN-ReLU 64 -> 64 s2) and each layer1 Bottleneck
// (hrnet.py:62-100: 1x1 reduce -> 3x3 -> 1x1 expand + residual, ReLU) as ONE launch, a workgroup walking a frame ROW BY ROW.
//
// Why: at 256 frames stem + layer1 were 1.46 ms of the 9.5 ms step and nothing but HBM round trips (profiles/r05_bf16_n256_layer_table.md): the
// 411 MB stem intermediate (64 ch @112x112) crossed HBM twice, and per Bottleneck the 64-channel tensors t and u were written and read back
// (4 x 103 MB) beside the 256-channel tensor's one read and one write (2 x 411 MB).  The MFMA work of these layers is 0.18 ms at peak.
// Here every intermediate lives in LDS: a launch reads its input once and writes its output once (stem: 154 MB of fp32 frames in, 103 MB out;
// Bottleneck: 411 MB in, 411 MB out; the first one 103 MB in).
//
// Shape of both kernels: workgroup = one frame (or one of S row segments of it, for calls with fewer frames than CUs), 8 waves, persistent over the
// rows.  The only spatial operator is the 3x3 convolution on a 64-channel tensor, so three rows of that tensor form a ring in LDS and one row
// step is: make the ring's newest row (Bottleneck: 1x1 reduce of x row y; stem: conv1 rows 2Y, 2Y + 1 gathered from the fp32 frame), run the 3x3
// for the output row whose lower neighbour has just arrived, finish it (Bottleneck: 1x1 expand + residual) and store it as whole NHWC rows.
// Nothing is recomputed inside a segment; a segment boundary costs one extra ring row.  The 3x3's weights (73.7 KB) stay in LDS for the whole
// launch (XOR-swizzled 64-byte rows as in conv_bf16_block_frame), the 1x1 weights in registers; the next input row is requested into registers
// two steps (Bottleneck) / one step (stem) ahead -- ordinary loads, so the counted vmcnt waits of hipcc apply -- and barriers are
// s_waitcnt lgkmcnt(0) + s_barrier (a __syncthreads() would drain that prefetch).
//
// Roles as everywhere on the bf16 path: A[cout l&15][k = 8(l>>4)+j] = weights, B[k][pixel l&15] = activations, D[cout 4(l>>4)+r][pixel l&15];
// k order = input-channel chunk (32) outer, tap inner -- the order of the launch-per-convolution kernels, so a fused launch differs from them by
// fp32 summation order inside a k-step at most (tests: equal to the fp32 oracle on bf16-rounded operands with bf16-rounded intermediates).
#include "kernels.h"

#include <cstdio>
#include <type_traits>

namespace grk {

#define GRK_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_r __attribute__((ext_vector_type(2)));
typedef float f32x2_r __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2_r(float lo, float hi) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_r{lo, hi}, bf16x2_r)); }
__device__ __forceinline__ float relu_r(float x) { const int i = __float_as_int(x); return __int_as_float(i > 0 ? i : 0); }
__device__ __forceinline__ float bflo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bfhi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ u32x2 pack4_relu(const f32x4 v) { return u32x2{pack2_r(relu_r(v[0]), relu_r(v[1])), pack2_r(relu_r(v[2]), relu_r(v[3]))}; }
__device__ __forceinline__ void lds_sync() {                  // every wave's LDS operations so far are done; global loads and stores stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

constexpr int kW2Bytes = 2 * 9 * 64 * 64;                     // the 64 -> 64 3x3 weights: [chunk 2][tap 9][cout 64][32 k] bf16, 64-byte rows
constexpr int kTSB = 160;                                     // slot stride of a 64-channel pixel in LDS: 128 + 32 bytes = 32 x odd (conflict-free b128 reads of 16 consecutive slots)

// the 3x3's weights global -> LDS, part p of row r at p ^ 2 (r >> 3 & 1) (bank groups of the 16 lanes of a ds_read_b128 group all different)
__device__ __forceinline__ void stage_w2(unsigned char* w2l, const void* w2g, int tid) {
    constexpr int NU = kW2Bytes / 16 / 512;
    u32x4 v[NU];
#pragma unroll
    for (int i = 0; i < NU; ++i) v[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(w2g) + (size_t)(i * 512 + tid) * 16);
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = i * 512 + tid, row = u >> 2, part = u & 3;
        *reinterpret_cast<u32x4*>(w2l + row * 64 + ((part ^ (((row >> 3) & 1) << 1)) * 16)) = v[i];
    }
}

// One output row of the 64 -> 64 3x3 (both strides): wave (blk, tp) owns output channels 16 blk .. + 15 of column tiles 2 tp, 2 tp + 1 (X = 16 T + l15).
// rb[dy]: this lane's B operand of tap row dy at X = l15 of tile 0, chunk 0, column tap 0; XS(dx): byte offset of column tap dx; a tile is TS bytes on.
template <typename XS>
__device__ __forceinline__ void conv3x3_row(f32x4 (&acc)[2], const unsigned char* (&rb)[3], const unsigned char* wl, int tp, XS xs, int ts) {
    // six groups (chunk, tap row) of 3 tap columns x 2 tiles: the next group's nine fragments are requested in front of this group's six MFMAs (left to itself hipcc
    // waits for a group's reads right in front of its MFMAs: six exposed LDS latencies per row)
    bf16x8 af[2][3], bfr[2][3][2];
    auto fetch = [&](int g, int slot) {
        const int c = g / 3, dy = g % 3;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            af[slot][dx] = *reinterpret_cast<const bf16x8*>(wl + ((c * 9 + dy * 3 + dx) * 64) * 64);
#pragma unroll
            for (int i = 0; i < 2; ++i) bfr[slot][dx][i] = *reinterpret_cast<const bf16x8*>(rb[dy] + (2 * tp + i) * ts + xs(dx) + c * 64);
        }
    };
    fetch(0, 0);
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        if (g + 1 < 6) fetch(g + 1, (g + 1) & 1);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][dx], bfr[g & 1][dx][i], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace

struct RollArgs {
    const void* in; int in_ctot, in_coff;                      // Bottleneck: x, NHWC bf16 (256 channels; the first one: the stem's 64)
    const float* frames;                                        // stem: the caller's fp32 NCHW frames (N, 3, 224, 224)
    void* out; int out_ctot, out_coff;
    int N, S;                                                   // frames; row segments per frame (1, 2 or 4)
    const void *w1, *w2, *w3;                                   // packed as the layers' own launches take them (grnet.cpp pack_conv); stem: w1 = pack_stem_weights_bf16
    const float *b1, *b2, *b3;
    int dbg;                                                    // diagnostic builds (make ABLATION=1, GRNET_ROLL_DBG): timing-only ablation bits, results garbage; 0 in the product
};

namespace {

// ---- layer1 Bottleneck.  FIRST: layer1.0 -- x has 64 channels, and relu(BN3(conv3(u)) + BNd(downsample(x))) is ONE GEMM over [u ; x] (K = 64 + 64: the plan's
// two-input 1x1, weights [4][1][256][32]) instead of expand + identity.
// Row step y (ring slot of a row = (row + 3) % 3; row -1 and row 56 are the 3x3's zero padding):
//   1. x row y: registers -> LDS (requested two steps ago); request x row y + 2                                                    | barrier
//   2. reduce: t[y] = relu(W1 x[y] + b1) -> ring; the lanes that will hold row y's outputs keep their residual values of x[y]      | barrier
//   3. 3x3: u[y-1] = relu(W2 * t[y-2 .. y] + b2) -> ubuf                                                                            | barrier
//   4. expand: relu((W3 u[y-1] + b3) + x[y-1]) -> staging (the x row's buffer: x[y] is dead by now)                                 | barrier
//   5. every thread moves the SAME 16-byte units it refills in the next step's phase 1 from the staging to HBM (whole rows of 512 bytes): no barrier between the two
// (A form with TWO rows per step -- half the barriers per row -- needs the LDS of the 3x3's weights for its rows; with those weights streamed from L2 every step
// instead it measured 306 us per Bottleneck against this form's 249: profiles/r06_roll_ablation.txt.)
template <bool FIRST>
__global__ __launch_bounds__(512) void conv_bf16_bneck(const RollArgs a) {
    constexpr int W = 56, CI = FIRST ? 64 : 256, KC1 = CI / 32, XSB = CI * 2 + 32, OSB = 544, TSLOTS = 66;
    constexpr int XROW = W * XSB, UPR = W * (CI / 8), NXU = (UPR + 511) / 512;       // bytes of a staged x row; its 16-byte units; units per thread
    constexpr int KC3 = FIRST ? 4 : 2, OUR = W * 32, NOU = (OUR + 511) / 512;        // k chunks of the expansion; 16-byte units of an output row
    extern __shared__ __align__(16) unsigned char lds[];
    unsigned char* w2l = lds;
    unsigned char* tring = w2l + kW2Bytes;                      // 3 rows x 66 slots: pixel x at slot x + 1, slots 0 and 57 .. 65 stay zero
    unsigned char* ubuf = tring + 3 * TSLOTS * kTSB;            // 56 slots
    unsigned char* xrow = ubuf + W * kTSB;                      // FIRST: two rows (by row parity: the expansion of row y - 1 reads x[y-1] while x[y] is staged)
    unsigned char* ostage = FIRST ? xrow + 2 * XROW : xrow;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x / a.S, seg = blockIdx.x - n * a.S;
    if (n >= a.N) return;
    const int rs = W / a.S, s0 = seg * rs, s1 = s0 + rs;
    const u16* inb = reinterpret_cast<const u16*>(a.in) + (size_t)n * W * W * a.in_ctot + a.in_coff;
    u16* outb = reinterpret_cast<u16*>(a.out) + (size_t)n * W * W * a.out_ctot + a.out_coff;

    // unit i of this thread (16 bytes of a row): its place in the staged row and in a row of HBM -- the same for every row, kept as two 32-bit offsets each.
    // A row is not a whole number of units per thread: the threads past its end LOAD a unit of the row's tail a second time (and drop it), so that every thread issues
    // the same number of loads per step.
    constexpr int XDUP = UPR % 512 ? 512 - UPR % 512 : 0, ODUP = OUR % 512 ? 512 - OUR % 512 : 0;
    int xl[NXU], xg[NXU], ol[NOU], og[NOU];
#pragma unroll
    for (int i = 0; i < NXU; ++i) {
        int u = i * 512 + tid;
        if (u >= UPR) u -= XDUP;
        const int px = u / (CI / 8), part = u - px * (CI / 8);
        xl[i] = px * XSB + part * 16;
        xg[i] = px * a.in_ctot + part * 8;
    }
#pragma unroll
    for (int i = 0; i < NOU; ++i) {
        int u = i * 512 + tid;
        if (u >= OUR) u -= ODUP;
        ol[i] = (u >> 5) * OSB + (u & 31) * 16;
        og[i] = (u >> 5) * a.out_ctot + (u & 31) * 8;
    }
    // hipcc counts the vector-memory queue in order, and its wait in front of the x row's first use came out as vmcnt(0) while loads or stores sat under branches of
    // the row loop ("anything may be pending" at the joins): every step then drained the two-step prefetch AND the previous step's stores -- compute, loads and stores
    // added up exactly (tools/roll_micro.py).  So the steps that store (rows s0 + 1 .. s1) are a loop whose body issues its NXU loads and NOU stores unconditionally --
    // rows the segment does not need are requested all the same (clamped, never used) -- and the two steps in front of it (ring rows s0 - 1, s0) are a copy without
    // the output phases.  (Loads hipcc does not count -- inline asm with a hand-counted wait -- are not an option for registers: it may copy them before the wait.)
    u32x4 nxa[NXU], nxb[NXU];
    auto request = [&](int y, u32x4 (&nx)[NXU]) {               // x row y -> registers
        const u16* rowp = inb + (size_t)(y < 0 ? 0 : y < W ? y : W - 1) * W * a.in_ctot;
#ifdef GRNET_ABLATION
        if (a.dbg & 16) return;
#endif
#pragma unroll
        for (int i = 0; i < NXU; ++i) nx[i] = *reinterpret_cast<const u32x4*>(rowp + xg[i]);
    };
    request(s0 - 1, nxa);
    request(s0, nxb);
    stage_w2(w2l, a.w2, tid);
    for (int u = tid; u < (3 * TSLOTS * kTSB) / 16; u += 512) reinterpret_cast<u32x4*>(tring)[u] = u32x4{0u, 0u, 0u, 0u};

    // ---- per-wave constants: reduce / 3x3 = (channel block blk, tile pair tp); expand = (channel blocks 2 wave, 2 wave + 1) x 4 tiles
    const int blk = wave & 3, tp = wave >> 2;
    bf16x8 w1f[KC1];
#pragma unroll
    for (int c = 0; c < KC1; ++c) w1f[c] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const u16*>(a.w1) + ((size_t)c * 64 + blk * 16 + l15) * 32 + lq * 8);
    bf16x8 w3f[KC3][2];
#pragma unroll
    for (int c = 0; c < KC3; ++c)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) w3f[c][bi] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const u16*>(a.w3) + ((size_t)c * 256 + (2 * wave + bi) * 16 + l15) * 32 + lq * 8);
    const f32x4 b1v = *reinterpret_cast<const f32x4*>(a.b1 + blk * 16 + lq * 4), b2v = *reinterpret_cast<const f32x4*>(a.b2 + blk * 16 + lq * 4);
    f32x4 b3v[2];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) b3v[bi] = *reinterpret_cast<const f32x4*>(a.b3 + (2 * wave + bi) * 16 + lq * 4);
    const unsigned char* wl = w2l + (blk * 16 + l15) * 64 + ((lq ^ (((l15 >> 3) & 1) << 1)) * 16);
    int pxc[4];                                                 // this lane's pixel of tile T, clamped into the row (lanes past pixel 55 compute copies nobody stores)
#pragma unroll
    for (int T = 0; T < 4; ++T) pxc[T] = (16 * T + l15) < W ? 16 * T + l15 : W - 1;
    const int pxr[2] = {32 * tp + l15, (32 * tp + 16 + l15) < W ? 32 * tp + 16 + l15 : W - 1};      // ... of this wave's two tiles of the reduce / 3x3 mapping
    u32x2 res_cur[2][4], res_nxt[2][4];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int T = 0; T < 4; ++T) res_cur[bi][T] = res_nxt[bi][T] = u32x2{0u, 0u};
    lds_sync();

    auto step = [&](auto out_tag, int y, u32x4 (&nx)[NXU]) {
        constexpr bool OUT = decltype(out_tag)::value;          // the step finishes output row y - 1
        const bool has_x = y >= 0 && y < W;
        unsigned char* xr = FIRST ? xrow + (y & 1) * XROW : xrow;
        // ---- 1. x row y -> LDS
        if (has_x) {
#pragma unroll
            for (int i = 0; i < NXU; ++i)
                if (i * 512 + tid < UPR) *reinterpret_cast<u32x4*>(xr + xl[i]) = nx[i];      // (a thread past the row's end holds a second copy of a tail unit: loaded, not written)
        }
        request(y + 2, nx);
        lds_sync();
        // ---- 2. reduce -> ring row y (a zero row for y = 56), residual capture
        unsigned char* trow = tring + ((y + 3) % 3) * (TSLOTS * kTSB);
        if (has_x) {
            f32x4 acc[2] = {b1v, b1v};
#ifdef GRNET_ABLATION
            if (!(a.dbg & 1))
#endif
#pragma unroll
            for (int c = 0; c < KC1; ++c)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 bt = *reinterpret_cast<const bf16x8*>(xr + pxr[i] * XSB + c * 64 + lq * 16);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[c], bt, acc[i], 0, 0, 0);
                }
            if (!FIRST) {
#pragma unroll
                for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                    for (int T = 0; T < 4; ++T) res_nxt[bi][T] = *reinterpret_cast<const u32x2*>(xr + pxc[T] * XSB + ((2 * wave + bi) * 16 + lq * 4) * 2);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int px = 16 * (2 * tp + i) + l15;
                if (px < W) *reinterpret_cast<u32x2*>(trow + (px + 1) * kTSB + (blk * 16 + lq * 4) * 2) = pack4_relu(acc[i]);
            }
        } else if (tid < W * 8) {
            *reinterpret_cast<u32x4*>(trow + ((tid >> 3) + 1) * kTSB + (tid & 7) * 16) = u32x4{0u, 0u, 0u, 0u};
        }
        lds_sync();
        if constexpr (OUT) {                                    // output row y - 1 (rows s0 .. s1 - 1)
            const int yo = y - 1;
            // ---- 3. 3x3 -> u
            {
                const unsigned char* rb[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) rb[dy] = tring + ((yo - 1 + dy + 3) % 3) * (TSLOTS * kTSB) + l15 * kTSB + lq * 16;
                f32x4 acc[2] = {b2v, b2v};
#ifdef GRNET_ABLATION
                if (!(a.dbg & 2))
#endif
                conv3x3_row(acc, rb, wl, tp, [](int dx) { return dx * kTSB; }, 16 * kTSB);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int px = 16 * (2 * tp + i) + l15;
                    if (px < W) *reinterpret_cast<u32x2*>(ubuf + px * kTSB + (blk * 16 + lq * 4) * 2) = pack4_relu(acc[i]);
                }
            }
            lds_sync();
            // ---- 4. expand (+ residual) -> staging
            {
                f32x4 acc[2][4];
#pragma unroll
                for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                    for (int T = 0; T < 4; ++T) acc[bi][T] = b3v[bi];
                const unsigned char* xo = xrow + (yo & 1) * XROW;     // FIRST: x row yo
#ifdef GRNET_ABLATION
                if (!(a.dbg & 4))
#endif
#pragma unroll
                for (int c = 0; c < KC3; ++c)
#pragma unroll
                    for (int T = 0; T < 4; ++T) {
                        const bf16x8 bt = c < 2 ? *reinterpret_cast<const bf16x8*>(ubuf + pxc[T] * kTSB + c * 64 + lq * 16)
                                                : *reinterpret_cast<const bf16x8*>(xo + pxc[T] * XSB + (c - 2) * 64 + lq * 16);
#pragma unroll
                        for (int bi = 0; bi < 2; ++bi) acc[bi][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f[c][bi], bt, acc[bi][T], 0, 0, 0);
                    }
#pragma unroll
                for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                    for (int T = 0; T < 4; ++T) {
                        f32x4 v = acc[bi][T];
                        if (!FIRST) {
                            const u32x2 r = res_cur[bi][T];
                            v[0] += bflo(r[0]); v[1] += bfhi(r[0]); v[2] += bflo(r[1]); v[3] += bfhi(r[1]);
                        }
                        if (16 * T + l15 < W) *reinterpret_cast<u32x2*>(ostage + (16 * T + l15) * OSB + ((2 * wave + bi) * 16 + lq * 4) * 2) = pack4_relu(v);
                    }
            }
            lds_sync();
            // ---- 5. staging -> HBM: unit i of the staging is unit i of the x row this thread writes in the next step's phase 1 (FIRST: a buffer of its own, next
            // written three barriers on) -- no barrier in between
#ifdef GRNET_ABLATION
            if (!(a.dbg & 8))
#endif
            {
                u16* rowp = outb + (size_t)yo * W * a.out_ctot;
#pragma unroll
                for (int i = 0; i < NOU; ++i)
                    if (i * 512 + tid < OUR) *reinterpret_cast<u32x4*>(rowp + og[i]) = *reinterpret_cast<const u32x4*>(ostage + ol[i]);      // (only a unit's owner may move it: the owner refills it without a barrier)
            }
        }
        if (!FIRST) {
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int T = 0; T < 4; ++T) res_cur[bi][T] = res_nxt[bi][T];
        }
    };
    step(std::false_type{}, s0 - 1, nxa);                       // ring rows s0 - 1 (row -1: zeros) and s0
    step(std::false_type{}, s0, nxb);
#pragma unroll 1
    for (int y = s0 + 1; y <= s1; y += 2) {                     // an even number of steps: rows per segment are 56, 28 or 14
        step(std::true_type{}, y, nxa);
        step(std::true_type{}, y + 1, nxb);
    }
}

// ---- layer1.1 - layer1.3 with the x rows prefetched by LDS-DMA.  In the kernel above the next rows travel through registers and hipcc's counted waits: its wait
// at the head of the row loop still asks for the one-step-old row, so a step has about one row of loads and one of stores in flight (57 KB per CU, 3.4 TB/s: the
// loads, the stores and the MFMA phases added up instead of overlapping, profiles/r06_roll_ablation.txt).  Here:
//   * x rows arrive by LDS-DMA into TWO row buffers (row y in buffer y & 1), requested right behind the barrier that ends the reduce of the row the buffer held --
//     1.7 steps ahead; no registers, no staging instructions.  The pieces are inline asm hipcc does not count, and the wait is counted by hand: the queue is in
//     order and a wave's step is [DMA pieces of row y + 2][stores of row y - 1], so in front of the reduce of row y a wave has its pieces of row y when at most
//     (stores + pieces + stores) = 3 ND operations are outstanding (ND = 4 for waves 0-3, 3 for waves 4-7, for pieces and stores alike); the first three steps
//     of a segment, whose queue is shorter, wait for everything;
//   * the LDS for the second row buffer and a staging row of its own: half of the 3x3's weights (input channels 32-63, 9 fragments = 36 registers per wave) live
//     in registers -- the registers the register prefetch held -- and the 256-channel rows are unpadded (512 bytes per pixel), the 16-byte part p of pixel x at
//     p ^ (x & 15): the 16 lanes of a fragment read (16 pixels, one part) hit 16 different bank groups, and the DMA (lane-linear in LDS) applies the permutation
//     in its source addresses; the staging row leaves in the same permuted order (each pixel's 512 bytes by 32 lanes: whole lines all the same).
// Steps, barriers and arithmetic are those of the kernel above.
#ifdef GRNET_ABLATION
__device__ unsigned long long g_roll_phase[8];                 // diagnostic builds, GRNET_ROLL_PHASES: clock ticks of wave 0 of every workgroup per phase of conv_bf16_bneck_dma
#define ROLL_TICK(k) do { if (a.dbg & 32) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc_[k] += t_ - tick_; tick_ = t_; } } while (0)
#else
#define ROLL_TICK(k) do { } while (0)
#endif
constexpr int kXRowB = 56 * 512;
constexpr int kBneckDmaLds = 9 * 64 * 64 + 3 * 58 * kTSB + 56 * kTSB + 3 * kXRowB + 384 * 4;      // 161 216
static_assert(kBneckDmaLds <= 160 * 1024, "LDS");
__device__ __forceinline__ void dma16_row(unsigned off, const void* base, unsigned lds) {      // lane l's 16 bytes at base + off land at lds + 16 l; one wait state between the M0 write and the DMA
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory");
}

__global__ __launch_bounds__(512) void conv_bf16_bneck_dma(const RollArgs a) {
    constexpr int W = 56, TSLOTS = 58, TROW = TSLOTS * kTSB;
    extern __shared__ __align__(16) unsigned char lds[];
    unsigned char* w2l = lds;                                   // chunk 0 of the 3x3's weights: [9][64][32] bf16
    unsigned char* tring = w2l + 9 * 64 * 64;                   // 3 rows x 58 slots (a lane past pixel 55 reads up to 8 slots past its row: inside the allocation, dropped)
    unsigned char* ubuf = tring + 3 * TROW;
    unsigned char* xb = ubuf + W * kTSB;                        // two x rows
    unsigned char* ost = xb + 2 * kXRowB;                       // the staging row
    float* bl = reinterpret_cast<float*>(ost + kXRowB);         // the three biases (64 + 64 + 256 floats): 16 registers the reduce's read batches need
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x / a.S, seg = blockIdx.x - n * a.S;
    if (n >= a.N) return;
    const int rs = W / a.S, s0 = seg * rs, s1 = s0 + rs;
    const u16* inb = reinterpret_cast<const u16*>(a.in) + (size_t)n * W * W * a.in_ctot + a.in_coff;
    u16* outb = reinterpret_cast<u16*>(a.out) + (size_t)n * W * W * a.out_ctot + a.out_coff;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;      // LDS byte address of the allocation

    // a row is 28 pieces of 1 KiB (two pixels each): wave w moves pieces w, w + 8, w + 16 (and w + 24 for w < 4); unit (lane) of a piece = pixel 2q + (l >> 5), slot l & 31
    const int ND = wave < 4 ? 4 : 3;
    // (piece w + 8 j starts 16 j pixels on: the permutation (x & 15) is the same for all four, so one lane offset + a uniform stride)
    const unsigned doff0 = (unsigned)((2 * wave + (lane >> 5)) * a.in_ctot + ((lane & 31) ^ ((2 * wave + (lane >> 5)) & 15)) * 8) * 2u, dstep = (unsigned)(16 * a.in_ctot) * 2u;
    auto request = [&](int y) {                                 // x row y -> buffer (y + 2) & 1 (rows outside the frame: clamped, landed and never used -- the counts stay the same)
        const u16* rowp = inb + (size_t)(y < 0 ? 0 : y < W ? y : W - 1) * W * a.in_ctot;
        const unsigned dst = lds0 + (unsigned)(xb - lds) + ((y + 2) & 1) * kXRowB + wave * 1024;
#ifdef GRNET_ABLATION
        if (a.dbg & 16) return;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < ND) dma16_row(doff0 + j * dstep, rowp, dst + j * 8192);
    };
    request(s0 - 1);
    request(s0);
    {   // chunk 0 of the 3x3's weights -> LDS (swizzled 64-byte rows), the rings' zeros
        constexpr int NU = 9 * 64 * 4;
        for (int u = tid; u < NU; u += 512) {
            const int row = u >> 2, part = u & 3;
            *reinterpret_cast<u32x4*>(w2l + row * 64 + ((part ^ (((row >> 3) & 1) << 1)) * 16)) = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned char*>(a.w2) + (size_t)u * 16);
        }
        for (int u = tid; u < (3 * TROW) / 16; u += 512) reinterpret_cast<u32x4*>(tring)[u] = u32x4{0u, 0u, 0u, 0u};
        if (tid < 64) { bl[tid] = a.b1[tid]; bl[64 + tid] = a.b2[tid]; }
        if (tid < 256) bl[128 + tid] = a.b3[tid];
    }
    const int blk = wave & 3, tp = wave >> 2;
    bf16x8 w1f[8], w2r[9], w3f[2][2];
#pragma unroll
    for (int c = 0; c < 8; ++c) w1f[c] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const u16*>(a.w1) + ((size_t)c * 64 + blk * 16 + l15) * 32 + lq * 8);
#pragma unroll
    for (int t = 0; t < 9; ++t) w2r[t] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const u16*>(a.w2) + ((size_t)(9 + t) * 64 + blk * 16 + l15) * 32 + lq * 8);
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) w3f[c][bi] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const u16*>(a.w3) + ((size_t)c * 256 + (2 * wave + bi) * 16 + l15) * 32 + lq * 8);
    const float* b1p = bl + blk * 16 + lq * 4;
    const float* b3p = bl + 128 + 2 * wave * 16 + lq * 4;
    const unsigned char* wl = w2l + (blk * 16 + l15) * 64 + ((lq ^ (((l15 >> 3) & 1) << 1)) * 16);
    int pxc[4];
#pragma unroll
    for (int T = 0; T < 4; ++T) pxc[T] = (16 * T + l15) < W ? 16 * T + l15 : W - 1;
    const int pxr[2] = {32 * tp + l15, (32 * tp + 16 + l15) < W ? 32 * tp + 16 + l15 : W - 1};
    // byte offset of (pixel px, 16-byte part p) in an unpadded row
    auto xat = [](int px, int p) { return px * 512 + ((p ^ (px & 15)) * 16); };
    // this thread's units of the staging row (linear in LDS: unit i * 512 + tid) and where they go in a row of HBM (16 i pixels on: same permutation)
    const int sg0 = (tid >> 5) * a.out_ctot + ((tid & 31) ^ ((tid >> 5) & 15)) * 8, sgstep = 16 * a.out_ctot;
    // the residuals of the current and of the next output row: x[y] as the lanes that will finish row y hold its outputs, taken while the reduce has the row.
    // (ONE set, refilled behind its use in the expansion, frees 32 registers but moves the request of row y + 2 behind that phase, next to the stores: measured
    // 244 us per launch against 227 -- the pieces issue slowly there and arrive late.)
    u32x2 res_cur[2][4], res_nxt[2][4];
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int T = 0; T < 4; ++T) res_cur[bi][T] = res_nxt[bi][T] = u32x2{0u, 0u};

#ifdef GRNET_ABLATION
    unsigned long long tick_ = __builtin_readcyclecounter(), tacc_[8] = {};
#endif
    auto step = [&](auto out_tag, auto steady_tag, int y) {
        constexpr bool OUT = decltype(out_tag)::value, STEADY = decltype(steady_tag)::value;
        const bool has_x = y >= 0 && y < W;
        const unsigned char* xr = xb + ((y + 2) & 1) * kXRowB;
        ROLL_TICK(7);
        // ---- 1. this wave's pieces of row y have landed; everybody's
        if (STEADY) {
            if (wave < 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ROLL_TICK(0);
        lds_sync();
        ROLL_TICK(1);
        // ---- 2. reduce -> ring row y (a zero row outside the frame), residual capture
        unsigned char* trow = tring + ((y + 3) % 3) * TROW;
        if (has_x) {
            const f32x4 b1v = *reinterpret_cast<const f32x4*>(b1p);
            f32x4 acc[2] = {b1v, b1v};
#ifdef GRNET_ABLATION
            if (!(a.dbg & 1))
#endif
            {   // the row's fragments in two batches of eight reads, the second requested in front of the first one's MFMAs (left to itself hipcc keeps one or two reads
                // in flight: 16 LDS latencies in a row were 1 500 of the phase's 2 400 cycles, GRNET_ROLL_PHASES)
                bf16x8 bt[2][4][2];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < 2; ++i) bt[0][c][i] = *reinterpret_cast<const bf16x8*>(xr + xat(pxr[i], 4 * c + lq));
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int i = 0; i < 2; ++i) bt[1][c][i] = *reinterpret_cast<const bf16x8*>(xr + xat(pxr[i], 4 * (c + 4) + lq));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[4 * h + c], bt[h][c][i], acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int T = 0; T < 4; ++T) res_nxt[bi][T] = *reinterpret_cast<const u32x2*>(xr + xat(pxc[T], (2 * wave + bi) * 2 + (lq >> 1)) + (lq & 1) * 8);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int px = 16 * (2 * tp + i) + l15;
                if (px < W) *reinterpret_cast<u32x2*>(trow + (px + 1) * kTSB + (blk * 16 + lq * 4) * 2) = pack4_relu(acc[i]);
            }
        } else if (tid < W * 8) {
            *reinterpret_cast<u32x4*>(trow + ((tid >> 3) + 1) * kTSB + (tid & 7) * 16) = u32x4{0u, 0u, 0u, 0u};
        }
        lds_sync();
        ROLL_TICK(2);
        request(y + 2);                                         // the buffer of row y is free
        ROLL_TICK(3);
        if constexpr (OUT) {
            const int yo = y - 1;
            // ---- 3. 3x3 -> u: chunk 0's weights from LDS, chunk 1's from registers
            {
                const unsigned char* rb[3];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) rb[dy] = tring + ((yo - 1 + dy + 3) % 3) * TROW + l15 * kTSB + lq * 16;
                const f32x4 b2v = *reinterpret_cast<const f32x4*>(b1p + 64);
                f32x4 acc[2] = {b2v, b2v};
#ifdef GRNET_ABLATION
                if (!(a.dbg & 2))
#endif
                {   // six groups (chunk, tap row) of 3 tap columns x 2 tiles; the next group's fragments are requested in front of this group's MFMAs
                    bf16x8 bfr[2][3][2], afl[2][3];
                    auto fetch = [&](int g, int slot) {
                        const int c = g / 3, dy = g % 3;
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            if (c == 0) afl[slot][dx] = *reinterpret_cast<const bf16x8*>(wl + ((dy * 3 + dx) * 64) * 64);
#pragma unroll
                            for (int i = 0; i < 2; ++i) bfr[slot][dx][i] = *reinterpret_cast<const bf16x8*>(rb[dy] + (16 * (2 * tp + i) + dx) * kTSB + c * 64);
                        }
                    };
                    fetch(0, 0);
#pragma unroll
                    for (int g = 0; g < 6; ++g) {
                        if (g + 1 < 6) fetch(g + 1, (g + 1) & 1);
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                            for (int i = 0; i < 2; ++i)
                                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g < 3 ? afl[g & 1][dx] : w2r[(g - 3) * 3 + dx], bfr[g & 1][dx][i], acc[i], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int px = 16 * (2 * tp + i) + l15;
                    if (px < W) *reinterpret_cast<u32x2*>(ubuf + px * kTSB + (blk * 16 + lq * 4) * 2) = pack4_relu(acc[i]);
                }
            }
            lds_sync();
            ROLL_TICK(4);
            // ---- 4. expand + residual -> staging row
            {
                f32x4 acc[2][4];
#pragma unroll
                for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                    for (int T = 0; T < 4; ++T) acc[bi][T] = *reinterpret_cast<const f32x4*>(b3p + bi * 16);
#ifdef GRNET_ABLATION
                if (!(a.dbg & 4))
#endif
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int T = 0; T < 4; ++T) {
                        const bf16x8 bt = *reinterpret_cast<const bf16x8*>(ubuf + pxc[T] * kTSB + c * 64 + lq * 16);
#pragma unroll
                        for (int bi = 0; bi < 2; ++bi) acc[bi][T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f[c][bi], bt, acc[bi][T], 0, 0, 0);
                    }
#pragma unroll
                for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                    for (int T = 0; T < 4; ++T) {
                        f32x4 v = acc[bi][T];
                        const u32x2 r = res_cur[bi][T];
                        v[0] += bflo(r[0]); v[1] += bfhi(r[0]); v[2] += bflo(r[1]); v[3] += bfhi(r[1]);
                        if (16 * T + l15 < W) *reinterpret_cast<u32x2*>(ost + xat(16 * T + l15, (2 * wave + bi) * 2 + (lq >> 1)) + (lq & 1) * 8) = pack4_relu(v);
                    }
            }
            lds_sync();
            ROLL_TICK(5);
            // ---- 5. staging row -> HBM (the staging is next written three barriers on)
#ifdef GRNET_ABLATION
            if (!(a.dbg & 8))
#endif
            {
                u16* rowp = outb + (size_t)yo * W * a.out_ctot;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < ND) *reinterpret_cast<u32x4*>(rowp + sg0 + i * sgstep) = *reinterpret_cast<const u32x4*>(ost + (i * 512 + tid) * 16);
            }
            ROLL_TICK(6);
        }
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int T = 0; T < 4; ++T) res_cur[bi][T] = res_nxt[bi][T];
    };
    step(std::false_type{}, std::false_type{}, s0 - 1);         // ring rows s0 - 1 (row -1: zeros) and s0
    step(std::false_type{}, std::false_type{}, s0);
    step(std::true_type{}, std::false_type{}, s0 + 1);          // the queue in front of these two is shorter than the steady one: they wait for everything
    step(std::true_type{}, std::false_type{}, s0 + 2);
#pragma unroll 1
    for (int y = s0 + 3; y <= s1; ++y) step(std::true_type{}, std::true_type{}, y);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the clamped pieces of the rows past the segment)
#ifdef GRNET_ABLATION
    if ((a.dbg & 32) && tid == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_roll_phase[k], tacc_[k]);
#endif
}

// ---- the stem pair.  conv1 as conv_bf16_stem computes it (K = (channel, tap) flattened to ONE 32-wide k-step, B gathered from the fp32 frame and rounded
// to bf16), a wave owning two column tiles of one of the two new conv1 rows of a step; its rows go to LDS de-interleaved by column parity -- the stride-2
// 3x3 then reads constant offsets: tap column 1 = even plane at X, tap columns 0 / 2 = odd plane at X / X + 1 (the odd plane's slot 0 is column -1: zero).
// Row step Y (output row of conv2): conv1 rows 2Y, 2Y + 1 -> ring (row 2Y - 1 is the previous step's second row; row -1: zeros) | barrier | gathers of the
// next step requested; conv2 row Y -> staging | barrier | staging -> 16-byte stores of whole rows (the next step's barrier orders them before the staging is rewritten).
constexpr int kStemPlane = 65, kStemRowSlots = 2 * kStemPlane;      // even plane: slots 0 .. 64 (columns 0, 2, .. 110 at 0 .. 55), odd plane: 65 .. 129 (column 2X - 1 at 65 + X)
__global__ __launch_bounds__(512) void conv_bf16_stem_pair(const RollArgs a) {
    constexpr int H = 224, W = 224, WO = 56;
    extern __shared__ __align__(16) unsigned char lds[];
    unsigned char* w2l = lds;
    unsigned char* ring = w2l + kW2Bytes;                       // 3 conv1 rows
    unsigned char* ostage = ring + 3 * kStemRowSlots * kTSB;    // 56 x 160
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x / a.S, seg = blockIdx.x - n * a.S;
    if (n >= a.N) return;
    const int rs = WO / a.S, s0 = seg * rs, s1 = s0 + rs;
    const float* fb = a.frames + (size_t)n * 3 * H * W;
    u16* outb = reinterpret_cast<u16*>(a.out) + (size_t)n * WO * WO * a.out_ctot + a.out_coff;

    // conv1: this wave's row of a step (0: 2Y, 1: 2Y + 1) and its column tiles (2 of the row's 7; the last wave of a row has one)
    const int crow = wave >> 2, ct0 = (wave & 3) * 2, nct = ct0 + 1 < 7 ? 2 : 1;
    bf16x8 wq[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) wq[mt] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const u16*>(a.w1) + ((size_t)mt * 64 + lane) * 8);
    f32x4 b1v[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) b1v[mt] = *reinterpret_cast<const f32x4*>(a.b1 + mt * 16 + lq * 4);
    int koff[8];                                                 // this lane's 8 K elements: k = 8 lq + j = channel * 9 + ky * 3 + kx (k >= 27: zero)
    int kky[8];
    bool kreal[8], kleft[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * lq + j, c = k / 9, tap = k - 9 * c, ky = tap / 3, kx = tap - 3 * ky;
        kreal[j] = k < 27;
        kleft[j] = kx == 0;
        kky[j] = ky - 1;
        koff[j] = (c * H + (ky - 1)) * W + (kx - 1);
    }
    float g[2][8];
    // conv1 row y1 (0 .. 111; clamped: rows past the frame are requested and never used), this wave's tiles -> registers (fp32, as they lie in the frame).  Every
    // load is issued by every lane -- padding elements read a clamped address and are replaced by zeros -- and the storing loop below has no branch around its loads
    // and stores: hipcc then counts its vmcnt waits (with the predicated gathers of the first form every use waited for vmcnt(0), i.e. also for the output row stored
    // just before: gathers, stores and MFMA phases added up exactly, tools/roll_micro.py).
    auto gather = [&](int y1r) {
        const int y1 = y1r < 0 ? 0 : y1r > 111 ? 111 : y1r;
#ifdef GRNET_ABLATION
        if (a.dbg & 16) return;
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int x = 16 * (ct0 + (i < nct ? i : 0)) + l15;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = kreal[j] && (2 * y1 + kky[j] >= 0) && !(kleft[j] && x == 0);
                const float v = fb[ok ? koff[j] + 2 * y1 * W + 2 * x : 0];
                g[i][j] = ok ? v : 0.f;
            }
        }
    };
    // a segment's first step also needs conv1 row 2 s0 - 1: a step of its own in front computes rows 2 s0 - 2, 2 s0 - 1 (for s0 = 0: two rows of nothing, kept zero)
    gather(2 * (s0 - 1) + crow);
    stage_w2(w2l, a.w2, tid);
    for (int u = tid; u < (3 * kStemRowSlots * kTSB) / 16; u += 512) reinterpret_cast<u32x4*>(ring)[u] = u32x4{0u, 0u, 0u, 0u};
    const int blk = wave & 3, tp = wave >> 2;
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(a.b2 + blk * 16 + lq * 4);
    const unsigned char* wl = w2l + (blk * 16 + l15) * 64 + ((lq ^ (((l15 >> 3) & 1) << 1)) * 16);
    lds_sync();

    auto conv1_rows = [&](int Y) {                               // conv1 rows 2Y, 2Y + 1 -> ring slots (row + 3) % 3 (rows above the frame: nothing is written, the slot keeps its zeros)
        const int y1 = 2 * Y + crow;
        if (y1 < 0) return;
        unsigned char* row = ring + ((y1 + 3) % 3) * (kStemRowSlots * kTSB);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i >= nct) break;
            const u32x4 bp = {pack2_r(g[i][0], g[i][1]), pack2_r(g[i][2], g[i][3]), pack2_r(g[i][4], g[i][5]), pack2_r(g[i][6], g[i][7])};
            const bf16x8 b = __builtin_bit_cast(bf16x8, bp);
            const int x1 = 16 * (ct0 + i) + l15;                 // conv1 column: even -> even plane at x1 / 2, odd -> odd plane at (x1 + 1) / 2
            unsigned char* dst = row + (((x1 & 1) ? kStemPlane + (x1 + 1) / 2 : x1 / 2)) * kTSB + lq * 8;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[mt], b, b1v[mt], 0, 0, 0);
                *reinterpret_cast<u32x2*>(dst + mt * 32) = pack4_relu(v);
            }
        }
    };
    conv1_rows(s0 - 1);
    lds_sync();
    gather(2 * s0 + crow);
    lds_sync();
    const int sl = tid >> 3, sp = tid & 7;                        // the thread's unit of an output row (56 pixels x 8 units of 16 bytes: threads 0 .. 447)
#pragma unroll 1
    for (int Y = s0; Y < s1; ++Y) {
        conv1_rows(Y);
        lds_sync();
        gather(2 * (Y + 1) + crow);                              // in flight under the 3x3 and the next barrier
        {
            // ---- conv2 row Y: tap row dy = conv1 row 2Y + dy - 1
            const unsigned char* rb[3];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) rb[dy] = ring + ((2 * Y + dy - 1 + 3) % 3) * (kStemRowSlots * kTSB) + l15 * kTSB + lq * 16;
            f32x4 acc[2] = {b2v, b2v};
#ifdef GRNET_ABLATION
            if (!(a.dbg & 2))
#endif
            conv3x3_row(acc, rb, wl, tp, [](int dx) { return dx == 1 ? 0 : (kStemPlane + (dx >> 1)) * kTSB; }, 16 * kTSB);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int px = 16 * (2 * tp + i) + l15;
                if (px < WO) *reinterpret_cast<u32x2*>(ostage + px * kTSB + (blk * 16 + lq * 4) * 2) = pack4_relu(acc[i]);
            }
        }
        lds_sync();
#ifdef GRNET_ABLATION
        if (!(a.dbg & 8))
#endif
        if (wave < 7)                                             // (wave-uniform: 448 threads move the row; the next step's barrier orders these reads before the staging is rewritten)
            *reinterpret_cast<u32x4*>(outb + (size_t)(Y * WO + sl) * a.out_ctot + sp * 8) = *reinterpret_cast<const u32x4*>(ostage + sl * kTSB + sp * 16);
    }
}

template <typename K>
hipError_t set_lds_roll(K kern, int bytes) { return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }

constexpr int kBneckLds = kW2Bytes + 3 * 66 * kTSB + 56 * kTSB + 56 * 544;                       // 144 832
constexpr int kBneckFirstLds = kW2Bytes + 3 * 66 * kTSB + 56 * kTSB + 2 * 56 * 160 + 56 * 544;  // 162 752
constexpr int kStemPairLds = kW2Bytes + 3 * kStemRowSlots * kTSB + 56 * kTSB;                   // 145 088
static_assert(kBneckFirstLds <= 160 * 1024 && kStemPairLds <= 160 * 1024, "LDS");

}  // namespace

hipError_t conv_bf16_roll_init() {
    GRK_TRY(set_lds_roll(conv_bf16_bneck<false>, kBneckLds));
    GRK_TRY(set_lds_roll(conv_bf16_bneck<true>, kBneckFirstLds));
    GRK_TRY(set_lds_roll(conv_bf16_bneck_dma, kBneckDmaLds));
    GRK_TRY(set_lds_roll(conv_bf16_stem_pair, kStemPairLds));
    return hipSuccess;
}

// row segments per frame: one workgroup per CU wants frames x segments >= CUs; a segment boundary costs one ring row (2 / 14 at four segments)
int conv_bf16_roll_segments(int n_frames) {
    int cus = 0;
    if (device_cu_count(&cus) != hipSuccess || cus <= 0) cus = 256;
    return n_frames >= cus ? 1 : 2 * n_frames >= cus ? 2 : 4;
}

// first: layer1.0 (64-channel input, expansion over [u ; x] with the downsample folded in: w3 packed [4][1][256][32])
hipError_t launch_conv_bf16_bneck(const void* in, int in_ctot, int in_coff, void* out, int out_ctot, int out_coff, int N, bool first, const void* w1, const float* b1,
                                  const void* w2, const float* b2, const void* w3, const float* b3, hipStream_t s) {
    const int cin = first ? 64 : 256;
    if (N < 1 || in_ctot % 8 != 0 || in_coff % 8 != 0 || out_ctot % 8 != 0 || out_coff % 8 != 0 || in_ctot - in_coff < cin || out_ctot - out_coff < 256) return hipErrorInvalidValue;
    RollArgs a{};
    a.in = in; a.in_ctot = in_ctot; a.in_coff = in_coff; a.out = out; a.out_ctot = out_ctot; a.out_coff = out_coff;
    a.N = N; a.S = conv_bf16_roll_segments(N);
    a.w1 = w1; a.w2 = w2; a.w3 = w3; a.b1 = b1; a.b2 = b2; a.b3 = b3;
    a.dbg = GRNET_AB(ROLL_DBG, 0);
    if (first) return launch_k(conv_bf16_bneck<true>, dim3(N * a.S), dim3(512), (size_t)kBneckFirstLds, s, a);
    if (GRNET_AB(ROLL_DMA, 1) && in_ctot * 2 * 56 * 56 < (1 << 30)) {     // (the pieces' lane offsets are 32-bit)
#ifdef GRNET_ABLATION
        if (GRNET_AB_SET(ROLL_PHASES)) {
            a.dbg |= 32;
            unsigned long long z[8] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_roll_phase), z, sizeof(z));
            hipError_t e = launch_k(conv_bf16_bneck_dma, dim3(N * a.S), dim3(512), (size_t)kBneckDmaLds, s, a);
            (void)hipStreamSynchronize(s);
            unsigned long long h[8] = {};
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_roll_phase), sizeof(h));
            const double wg = (double)N * a.S * 57.0 / a.S;      // per workgroup and row step (approximately: the warm-up steps have no output phases)
            fprintf(stderr, "[roll phases] ticks per step: wait-dma %.0f  barrier1 %.0f  reduce+barrier %.0f  request %.0f  3x3+barrier %.0f  expand+barrier %.0f  store %.0f  tail %.0f\n",
                    h[0] / wg, h[1] / wg, h[2] / wg, h[3] / wg, h[4] / wg, h[5] / wg, h[6] / wg, h[7] / wg);
            return e;
        }
#endif
        return launch_k(conv_bf16_bneck_dma, dim3(N * a.S), dim3(512), (size_t)kBneckDmaLds, s, a);
    }
    return launch_k(conv_bf16_bneck<false>, dim3(N * a.S), dim3(512), (size_t)kBneckLds, s, a);
}

hipError_t launch_conv_bf16_stem_pair(const float* frames, void* out, int out_ctot, int out_coff, int N, const void* w1pk, const float* b1, const void* w2, const float* b2,
                                      hipStream_t s) {
    if (N < 1 || out_ctot % 8 != 0 || out_coff % 8 != 0 || out_ctot - out_coff < 64) return hipErrorInvalidValue;
    RollArgs a{};
    a.frames = frames; a.out = out; a.out_ctot = out_ctot; a.out_coff = out_coff;
    a.N = N; a.S = conv_bf16_roll_segments(N);
    a.w1 = w1pk; a.w2 = w2; a.b1 = b1; a.b2 = b2;
    a.dbg = GRNET_AB(ROLL_DBG, 0);
    return launch_k(conv_bf16_stem_pair, dim3(N * a.S), dim3(512), (size_t)kStemPairLds, s, a);
}

}  // namespace grk
