// Temporal / spatial attention block of the pose-feature corrector.
// Reference: TSAttnBlock.forward with use_jwff=True, eval -- lib/models/layers/attention_utils.py:261-270, built from
//   MultiAttention.forward :164-217, JointWiseFeedForward.forward :123-130, LayerNormalization.forward :17-27,
//   LocallyConnected2d.forward (locallyconnected2d.py:39-48); configuration of feature_correction.py:92-101 for one layer:
//   in_dim = out_dim = 3072 (128 features x 24 joints, index c*24+j), encode_dim 1000, 4 heads, 24 (+1) tokens.
// With R = b*n rows (frames) the block is
//   1. QT = X . Wqkv_t^T + b (R x 3000)   QS = XS . Wqkv_s^T + b (R x 3000)        fp32 MFMA GEMMs
//   2. temporal attention per (clip, head): softmax over the n frames, dim_head 250, scaled by 1/sqrt(250)
//   3. spatial attention per (frame, head): the 250 dims are (C=10, tokens=25); 25x25 scores, NOT scaled
//   4. gate: clip mean of [x_t | x_s] -> Linear 2000x2000 -> pairs (2e, 2e+1) -> softmax -> scales x_t / x_s
//   5. Y = fc_t(x_t') + fc_s(x_s')        two GEMMs (R x 3072)
//   6. X1 = LN1(X + Y);  out = LN2(JWFF(X1) + X1)   LN = the reference's own: unbiased std, (std + eps)
// Everything is fp32; reductions run in a fixed order (deterministic).
#include "kernels.h"

#include <algorithm>
#include <cstdlib>

namespace grk {

#define GRK_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

namespace {

constexpr int kD = 3072, kE = 1000, kH = 4, kDh = 250, kTok = 24, kTokS = 25, kC = 10, kFF = 64, kF = 128;

__device__ __forceinline__ float block_reduce(float v, float* red, bool is_max) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (tid < s) red[tid] = is_max ? fmaxf(red[tid], red[tid + s]) : red[tid] + red[tid + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// grid (n, H, b): one query frame i of one head of one clip.  attention_utils.py:190-205.
__global__ __launch_bounds__(256) void temporal_attn_kernel(const float* __restrict__ qkv, float* __restrict__ xt, int n) {
    extern __shared__ float sm[];
    float* q = sm;               // [250]
    float* p = sm + 256;         // [n]
    float* red = p + n;          // [256]
    const int i = blockIdx.x, h = blockIdx.y, bi = blockIdx.z, tid = threadIdx.x;
    const float* base = qkv + (size_t)bi * n * 3 * kE;
    const float* qrow = base + (size_t)i * 3 * kE + (0 * kH + h) * kDh;
    if (tid < kDh) q[tid] = qrow[tid];
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)kDh);
    float lmax = -INFINITY;
    for (int j = tid; j < n; j += 256) {
        const float* krow = base + (size_t)j * 3 * kE + (1 * kH + h) * kDh;
        float s = 0.f;
        for (int d = 0; d < kDh; ++d) s = fmaf(q[d], krow[d], s);
        s *= scale;
        p[j] = s;
        lmax = fmaxf(lmax, s);
    }
    const float m = block_reduce(lmax, red, true);
    float lsum = 0.f;
    for (int j = tid; j < n; j += 256) {
        const float e = expf(p[j] - m);
        p[j] = e;
        lsum += e;
    }
    const float inv = 1.0f / block_reduce(lsum, red, false);
    if (tid < kDh) {
        float acc = 0.f;
        for (int j = 0; j < n; ++j) acc = fmaf(p[j], base[(size_t)j * 3 * kE + (2 * kH + h) * kDh + tid], acc);
        xt[((size_t)bi * n + i) * kE + h * kDh + tid] = acc * inv;
    }
}

// The same attention for LONG clips, blocked.  The kernel above reads every key and value of the clip once per QUERY: at 10 000 frames that is
// 10 000 x 4 x 2 x 10 MB = 800 GB through L2 and 107 ms.  Here a workgroup owns 128 queries of one head (8 waves x 16); keys and values stream through LDS
// in blocks of 32 (read once per 128 queries), and both products run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products and sums).
// The softmax is the running-maximum form, per query: m <- max(m, max_j s_j), O <- O 2^(m_old - m) + sum_j 2^(s_j - m) v_j, l likewise, out = O / l, with
// log2(e) / sqrt(d) folded into Q (v_exp_f32) -- the arithmetic of softmax(QK^T / sqrt(d)) V in fp32 up to the order of the sums and the last bit of the
// exponential; deterministic (no atomics, fixed order).
// Round 6: everything TRANSPOSED.  S^T = K Q^T (A operand: a K tile from LDS, B operand: the wave's Q fragment, 63 registers per lane), so a lane holds eight
// scores of ONE query (keys 4 lq + r and 16 + 4 lq + r of the block, query l15): the row maximum is 7 in-lane maxima and two cross-lane steps (round 5: 32
// ds_bpermute per block), the rescale factor is one scalar per lane, the row sum stays lane-partial until the end.  The accumulator layout of S^T IS the B
// operand layout of O^T += V^T P^T (k index = lq, column = query l15), so P never leaves the registers (round 5: written to LDS and read back), and the O^T
// tile (rows = feature 16 dt + 4 lq + r, column = query l15) keeps every lane on its own query: the rescale is in-lane and skipped while no maximum of
// the wave moved.  Without the P tiles the K / V blocks fit twice: the next block is requested into registers before the current one is computed and stored
// into the OTHER buffer behind it -- one barrier per block.  Round 5's form ran 9.54 ms at 10 000 frames (profiles/r06_tail_overlap.txt).
// Row strides: 258 floats for K (2 mod 32: the 16 keys x 2 columns of a 32-lane ds_read_b32 group fall on 32 different banks), 260 for V (4 x 260 = 16 mod
// 32: 2 key groups x 16 columns likewise).
// Keys are split over `kparts` workgroups per (query tile, head) where one round of workgroups would leave CUs idle or a second round nearly empty (10 000
// frames: 79 x 4 = 316 workgroups on 256 CUs); each part leaves (O^T unnormalised, m, l) and temporal_attn_combine_kernel merges them in part order.
constexpr int kFW = 8, kFQ = 16 * kFW, kFK = 32, kFLd = 258, kFLdV = 260;
constexpr int kFBufFloats = kFK * kFLd + kFK * kFLdV;
constexpr int kFlashLdsFloats = 2 * kFBufFloats;
constexpr int kFU = kFK * (kDh / 2), kFLoads = (kFU + 64 * kFW - 1) / (64 * kFW);       // float2 units of a K (or V) block; per thread
constexpr int kFlashMaxParts = 8, kFlashMinPartKeys = 128;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64 * kFW) void temporal_attn_flash_kernel(const float* __restrict__ qkv, float* __restrict__ xt, int n, int kparts,
                                                                         float* __restrict__ part_o, float* __restrict__ part_ml) {
    extern __shared__ float sm[];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4, wave = tid >> 6;
    const int q0 = blockIdx.x * kFQ, h = blockIdx.y, bi = blockIdx.z / kparts, part = blockIdx.z - bi * kparts;
    const float* base = qkv + (size_t)bi * n * 3 * kE;
    const int nblk = (n + kFK - 1) / kFK;
    const int kbeg = (int)((long)nblk * part / kparts) * kFK, kend = min(n, (int)((long)nblk * (part + 1) / kparts) * kFK);
    typedef float f2 __attribute__((ext_vector_type(2)));
    // this lane's part of the wave's Q fragment: query 16 wave + l15, columns 4 j + lq (columns 250, 251: zero; queries past the clip: zero rows, never stored)
    float qf[63];
    {
        const float scale = 1.4426950408889634f / sqrtf((float)kDh);
        const int q = q0 + wave * 16 + l15;
        const float* qrow = base + (size_t)(q < n ? q : 0) * 3 * kE + (0 * kH + h) * kDh + lq;
#pragma unroll
        for (int j = 0; j < 63; ++j) qf[j] = (q < n && 4 * j + lq < kDh) ? qrow[4 * j] * scale : 0.f;
    }
    // K columns 250 .. 257 and V columns 250 .. 259 of both buffers: zero for good (the staging below never writes them)
    for (int u = tid; u < 2 * kFK * (kFLd - kDh); u += 64 * kFW) {
        const int bf = u / (kFK * (kFLd - kDh)), v = u - bf * kFK * (kFLd - kDh);
        sm[bf * kFBufFloats + (v / (kFLd - kDh)) * kFLd + kDh + v % (kFLd - kDh)] = 0.f;
    }
    for (int u = tid; u < 2 * kFK * (kFLdV - kDh); u += 64 * kFW) {
        const int bf = u / (kFK * (kFLdV - kDh)), v = u - bf * kFK * (kFLdV - kDh);
        sm[bf * kFBufFloats + kFK * kFLd + (v / (kFLdV - kDh)) * kFLdV + kDh + v % (kFLdV - kDh)] = 0.f;
    }
    // staging map of a K / V block: unit u = (row, float2 column); rows are 8-byte aligned (a head starts 1000 bytes into its row)
    // a block's K half is requested before the S^T phase and stored behind it, its V half before / behind the O^T phase: eight staging registers, not sixteen
    f2 reg[kFLoads];
    auto request = [&](int k0, int which) {                    // which: 1 keys, 2 values; keys past the clip's end: zero rows (their scores are masked below)
        const float* gb = base + (size_t)k0 * 3 * kE + (which * kH + h) * kDh;
#pragma unroll
        for (int i = 0; i < kFLoads; ++i) {
            const int u = i * 64 * kFW + tid, r = u / (kDh / 2), c2 = u - r * (kDh / 2);      // (recomputed per block: eight registers matter here)
            const bool ok = u < kFU && k0 + r < n;
            reg[i] = ok ? *reinterpret_cast<const f2*>(gb + r * 3 * kE + 2 * c2) : f2{0.f, 0.f};
        }
    };
    auto deposit = [&](float* buf, int ld) {
#pragma unroll
        for (int i = 0; i < kFLoads; ++i)
            if (i * 64 * kFW + tid < kFU) {
                const int u = i * 64 * kFW + tid, r = u / (kDh / 2), c2 = u - r * (kDh / 2);
                *reinterpret_cast<f2*>(buf + r * ld + 2 * c2) = reg[i];
            }
    };
    float m_run = -INFINITY, l_run = 0.f;                      // of query l15, over this lane's keys (4 lq + r, 16 + 4 lq + r of every block) for l
    f32x4_t O[16];                                             // O^T: rows = features 16 dt + 4 lq + r, column = query l15
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) O[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    request(kbeg, 1);
    deposit(sm, kFLd);
    request(kbeg, 2);
    deposit(sm + kFK * kFLd, kFLdV);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += kFK, cur ^= 1) {
        const float* Ks = sm + cur * kFBufFloats;
        const float* Vs = Ks + kFK * kFLd;
        const bool more = k0 + kFK < kend;
        float* nxt = sm + (cur ^ 1) * kFBufFloats;              // free since the barrier that ended the previous block
        if (more) request(k0 + kFK, 1);                        // the next block's keys: in flight under this block's S^T MFMAs
        // S^T (32 keys x 16 queries) = K Q^T: 63 k-steps of 4 (columns 250, 251 are zeros)
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        const float* k0row = Ks + l15 * kFLd + lq;
        const float* k1row = Ks + (16 + l15) * kFLd + lq;
#pragma unroll
        for (int j = 0; j < 63; ++j) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k0row[4 * j], qf[j], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1row[4 * j], qf[j], s1, 0, 0, 0);
            if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);   // operand reads are hoisted at most 4 steps (8 registers) ahead, not 63
        }
        if (more) { deposit(nxt, kFLd); request(k0 + kFK, 2); } // ... and its values under the O^T MFMAs
        if (k0 + kFK > n) {                                    // keys past the clip's end are -inf (only the clip's last block)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (k0 + 4 * lq + r >= n) s0[r] = -INFINITY;
                if (k0 + 16 + 4 * lq + r >= n) s1[r] = -INFINITY;
            }
        }
        float mx = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);                  // finite: key k0 exists
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(s0[r] - m_new);
            s1[r] = __builtin_amdgcn_exp2f(s1[r] - m_new);
            psum += s0[r] + s1[r];
        }
        l_run = l_run * alpha + psum;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f)) {
#pragma unroll
            for (int dt = 0; dt < 16; ++dt) O[dt] *= alpha;
        }
        // O^T (256 features x 16 queries) += V^T P^T: 8 k-steps; k index lq of step (t, r) is key 16 t + 4 lq + r -- the lane's own s_t[r]
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = t ? s1[r] : s0[r];
                const float* vrow = Vs + (16 * t + 4 * lq + r) * kFLdV + l15;
#pragma unroll
                for (int dt = 0; dt < 16; ++dt) O[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[dt * 16], pv, O[dt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);             // (one k-step's 16 V reads in flight, not all eight)
            }
        if (more) deposit(nxt + kFK * kFLd, kFLdV);
        __syncthreads();                                       // every wave has read block k0 and stored its share of block k0 + 32
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const int q = q0 + wave * 16 + l15;
    if (q >= n) return;
    if (kparts == 1) {
        const float inv = 1.0f / l_run;
        float* orow = xt + ((size_t)bi * n + q) * kE + h * kDh + 4 * lq;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) {
            if (dt * 16 + 4 * lq < kDh) *reinterpret_cast<f2*>(orow + dt * 16) = f2{O[dt][0] * inv, O[dt][1] * inv};
            if (dt * 16 + 4 * lq + 2 < kDh) *reinterpret_cast<f2*>(orow + dt * 16 + 2) = f2{O[dt][2] * inv, O[dt][3] * inv};
        }
    } else {                                                   // this part's share: part_o (part, row, 1000) unnormalised, part_ml (part, row, head, {m, l})
        const size_t rows = (size_t)gridDim.z / kparts * n, row = (size_t)bi * n + q;
        float* orow = part_o + ((size_t)part * rows + row) * kE + h * kDh + 4 * lq;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) {
            if (dt * 16 + 4 * lq < kDh) *reinterpret_cast<f2*>(orow + dt * 16) = f2{O[dt][0], O[dt][1]};
            if (dt * 16 + 4 * lq + 2 < kDh) *reinterpret_cast<f2*>(orow + dt * 16 + 2) = f2{O[dt][2], O[dt][3]};
        }
        if (lq == 0) *reinterpret_cast<f2*>(part_ml + (((size_t)part * rows + row) * kH + h) * 2) = f2{m_run, l_run};
    }
}

// out[row, h, :] = sum_p O_p 2^(m_p - M) / sum_p l_p 2^(m_p - M), M = max_p m_p: the parts of one (row, head) merged in part order.  grid (rows), 256 threads.
__global__ __launch_bounds__(256) void temporal_attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, float* __restrict__ xt,
                                                                      size_t rows, int kparts) {
    const size_t row = blockIdx.x;
    for (int e = threadIdx.x; e < kE; e += 256) {
        const int h = e / kDh;
        float M = -INFINITY;
        for (int p = 0; p < kparts; ++p) M = fmaxf(M, part_ml[(((size_t)p * rows + row) * kH + h) * 2]);
        float acc = 0.f, l = 0.f;
        for (int p = 0; p < kparts; ++p) {
            const float* ml = part_ml + (((size_t)p * rows + row) * kH + h) * 2;
            const float w = __builtin_amdgcn_exp2f(ml[0] - M);
            l += ml[1] * w;
            acc += part_o[((size_t)p * rows + row) * kE + e] * w;
        }
        xt[row * kE + e] = acc / l;
    }
}

// Key parts per (query tile, head) of a clip of n frames: the count that minimises rounds of workgroups per part on `cus` CUs (one workgroup per CU: 133 KB of
// LDS), from n alone -- a clip's result does not depend on what else is in the batch.
static int flash_key_parts(int n, int cus) {
    const long wgs = (long)((n + kFQ - 1) / kFQ) * kH;
    int best = 1;
    double best_cost = 1e30;
    for (int p = 1; p <= kFlashMaxParts; ++p) {
        if (p > 1 && n / p < GRNET_AB(TSATTN_PART_KEYS, kFlashMinPartKeys)) break;
        const double cost = (double)((wgs * p + cus - 1) / cus) / p + 0.01 * p;          // rounds x part length (+ a little per part for the merge)
        if (cost < best_cost - 1e-9) { best_cost = cost; best = p; }
    }
    return best;
}

// grid (R, H): the 25 tokens of one frame, one head.  attention_utils.py:207-217.
__global__ __launch_bounds__(256) void spatial_attn_kernel(const float* __restrict__ qkv, float* __restrict__ xs) {
    __shared__ float q[kDh], k[kDh], v[kDh], a[kTokS * kTokS];
    const int r = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const float* row = qkv + (size_t)r * 3 * kE;
    if (tid < kDh) {
        q[tid] = row[(0 * kH + h) * kDh + tid];
        k[tid] = row[(1 * kH + h) * kDh + tid];
        v[tid] = row[(2 * kH + h) * kDh + tid];
    }
    __syncthreads();
    for (int idx = tid; idx < kTokS * kTokS; idx += 256) {
        const int t1 = idx / kTokS, t2 = idx - t1 * kTokS;
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < kC; ++c) s = fmaf(q[c * kTokS + t1], k[c * kTokS + t2], s);
        a[idx] = s;
    }
    __syncthreads();
    if (tid < kTokS) {
        float m = -INFINITY, sum = 0.f;
        for (int t2 = 0; t2 < kTokS; ++t2) m = fmaxf(m, a[tid * kTokS + t2]);
        for (int t2 = 0; t2 < kTokS; ++t2) { const float e = expf(a[tid * kTokS + t2] - m); a[tid * kTokS + t2] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int t2 = 0; t2 < kTokS; ++t2) a[tid * kTokS + t2] *= inv;
    }
    __syncthreads();
    if (tid < kDh) {
        const int c = tid / kTokS, t1 = tid - c * kTokS;
        float acc = 0.f;
        for (int t2 = 0; t2 < kTokS; ++t2) acc = fmaf(a[t1 * kTokS + t2], v[c * kTokS + t2], acc);
        xs[(size_t)r * kE + h * kDh + tid] = acc;                         // index c*25 + t1 (the transpose of :214)
    }
}

// mean over the n frames of a clip of [x_t | x_s]  -> (b, 2000).  attention_utils.py:183-184.
// Two launches: partial sums over blocks of kMeanRows frames (grid z), then the partials in a fixed order -- one workgroup column per clip walked all
// 10 000 frames of a configs[3] job by itself (8 workgroups, 3.7 ms for 80 MB; now ~0.1 ms).  Deterministic: no atomics.
constexpr int kMeanRows = 128;
__global__ __launch_bounds__(256) void gate_mean_partial_kernel(const float* __restrict__ xt, const float* __restrict__ xs, float* __restrict__ part, int n, int nblk) {
    const int e = blockIdx.x * 256 + threadIdx.x, bi = blockIdx.y, blk = blockIdx.z;
    if (e >= 2 * kE) return;
    const float* src = (e < kE ? xt : xs) + (size_t)bi * n * kE + (e < kE ? e : e - kE);
    const int i0 = blk * kMeanRows, i1 = min(n, i0 + kMeanRows);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int i = i0;
    for (; i + 4 <= i1; i += 4) {
        s0 += src[(size_t)i * kE]; s1 += src[(size_t)(i + 1) * kE]; s2 += src[(size_t)(i + 2) * kE]; s3 += src[(size_t)(i + 3) * kE];
    }
    for (; i < i1; ++i) s0 += src[(size_t)i * kE];
    part[((size_t)bi * nblk + blk) * 2 * kE + e] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void gate_mean_final_kernel(const float* __restrict__ part, float* __restrict__ mean, int n, int nblk) {
    const int e = blockIdx.x * 256 + threadIdx.x, bi = blockIdx.y;
    if (e >= 2 * kE) return;
    float s = 0.f;
    for (int k = 0; k < nblk; ++k) s += part[((size_t)bi * nblk + k) * 2 * kE + e];
    mean[(size_t)bi * 2 * kE + e] = s / (float)n;
}

// alpha = softmax over the pairs (2e, 2e+1) of the gate logits; x_t *= alpha0, x_s *= alpha1.  attention_utils.py:185-188.
__global__ __launch_bounds__(256) void gate_apply_kernel(const float* __restrict__ logits, float* __restrict__ xt, float* __restrict__ xs, int n, long total) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / kE;
        const int e = (int)(i - r * kE), bi = (int)(r / n);
        const float a0 = logits[(size_t)bi * 2 * kE + 2 * e], a1 = logits[(size_t)bi * 2 * kE + 2 * e + 1];
        const float m = fmaxf(a0, a1), e0 = expf(a0 - m), e1 = expf(a1 - m), inv = 1.0f / (e0 + e1);
        xt[i] *= e0 * inv;
        xs[i] *= e1 * inv;
    }
}

// the reference's LayerNormalization on a row held in LDS: unbiased std, (std + eps).  attention_utils.py:17-27.
__device__ __forceinline__ void layer_norm_row(const float* z, const float* __restrict__ g, const float* __restrict__ b, float* __restrict__ out,
                                               float* red) {
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < kD; i += 256) s += z[i];
    const float mean = block_reduce(s, red, false) / (float)kD;
    float ss = 0.f;
    for (int i = tid; i < kD; i += 256) { const float d = z[i] - mean; ss = fmaf(d, d, ss); }
    const float sd = sqrtf(block_reduce(ss, red, false) / (float)(kD - 1));
    const float inv = 1.0f / (sd + 1e-6f);
    for (int i = tid; i < kD; i += 256) out[i] = g[i] * ((z[i] - mean) * inv) + b[i];
}

// X1 = LN1(x + (y_t + y_s)).  attention_utils.py:188, :265-266.   grid R.
__global__ __launch_bounds__(256) void residual_ln_kernel(const float* __restrict__ x, const float* __restrict__ yt, const float* __restrict__ ys,
                                                            const float* __restrict__ g, const float* __restrict__ b, float* __restrict__ out) {
    __shared__ float z[kD];
    __shared__ float red[256];
    const size_t r = blockIdx.x;
    for (int i = threadIdx.x; i < kD; i += 256) z[i] = x[r * kD + i] + (yt[r * kD + i] + ys[r * kD + i]);
    __syncthreads();
    layer_norm_row(z, g, b, out + r * kD, red);
}

__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

// out = LN2(JWFF(x1) + x1): two per-joint locally connected layers 128 -> 64 -> 128 with an exact GELU between.
// attention_utils.py:123-130, :267-268; weights w1 [64][128][24], w2 [128][64][24].   grid R.
__global__ __launch_bounds__(256) void jwff_ln_kernel(const float* __restrict__ x1, const float* __restrict__ w1, const float* __restrict__ w2,
                                                        const float* __restrict__ g, const float* __restrict__ b, float* __restrict__ out) {
    __shared__ float xr[kD];
    __shared__ float hid[kFF * kTok];
    __shared__ float red[256];
    const size_t r = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < kD; i += 256) xr[i] = x1[r * kD + i];
    __syncthreads();
    for (int idx = tid; idx < kFF * kTok; idx += 256) {
        const int o = idx / kTok, j = idx - o * kTok;
        const float* w = w1 + (size_t)o * kF * kTok + j;
        float s = 0.f;
        for (int c = 0; c < kF; ++c) s = fmaf(xr[c * kTok + j], w[c * kTok], s);
        hid[idx] = gelu_exact(s);
    }
    __syncthreads();
    float zreg[kD / 256];
#pragma unroll
    for (int u = 0; u < kD / 256; ++u) {
        const int idx = u * 256 + tid, p = idx / kTok, j = idx - p * kTok;
        const float* w = w2 + (size_t)p * kFF * kTok + j;
        float s = 0.f;
        for (int o = 0; o < kFF; ++o) s = fmaf(hid[o * kTok + j], w[o * kTok], s);
        zreg[u] = s + xr[idx];
    }
    __syncthreads();                                       // every thread is done reading xr
#pragma unroll
    for (int u = 0; u < kD / 256; ++u) xr[u * 256 + tid] = zreg[u];
    __syncthreads();
    layer_norm_row(xr, g, b, out + r * kD, red);
}

}  // namespace

size_t tsattn_ws_floats(int b, int n) {
    const size_t R = (size_t)b * n;
    return 2 * R * 3 * kE + 2 * R * kE + 2 * (size_t)b * 2 * kE + 3 * R * kD + 64;
}

// The longest clip the attention kernel takes on the CURRENT device: its query row + softmax row + reduction scratch ((512 + n) floats) must
// fit the LDS one workgroup may have (160 KB on gfx950 -> kTsAttnMaxFrames; 64 KB parts: 15 872 frames).  Both entry points ask this BEFORE
// anything is enqueued, so a clip that cannot run is refused with the remedy instead of failing inside the launch.
static hipError_t device_lds_per_block(int* bytes) {
    static PerDeviceOnce c;
    int dev = 0;
    hipError_t e = current_device(&dev);
    if (e != hipSuccess) return e;
    return once_per_device(c, dev, [&](int* v) { return hipDeviceGetAttribute(v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev); }, bytes);
}
// Smallest clip on the blocked kernel; 0: never.  Whole attention block, ms, per-query against blocked (keys split down to 128 per part): 1 x 256 frames 0.537 / 0.534,
// 1 x 400 0.717 / 0.653, 4 x 400 1.556 / 1.160, 1 x 1000 1.49 / 0.84, 1 x 2000 4.20 / 1.47 (the temporal attention alone at 10 000: 107 / 4.1); below 256 the per-query
// kernel's n workgroups per head fill the device better than one or two query tiles.
static int tsattn_flash_min() { return GRNET_AB(TSATTN_FLASH, 384); }
static int per_query_max_frames() {                                          // what the per-query kernel's softmax row leaves of THIS device's LDS per workgroup
    int lds = 0;
    if (device_lds_per_block(&lds) != hipSuccess || lds <= 0) return 0;
    const long fit = (long)lds / (long)sizeof(float) - 512;
    return (int)std::min<long>(kTsAttnMaxFrames, fit > 0 ? fit : 0);
}
// The per-query kernel holds a softmax row over the clip in LDS; the blocked kernel's LDS use is fixed (133 KB), so with it on, the per-query limit binds
// only the clips that still run the per-query kernel (round-5 advice: the entry check turned away clips the blocked kernel could serve)
int tsattn_max_frames() {
    const int pq = per_query_max_frames(), fm = tsattn_flash_min();
    return (fm > 0 && fm - 1 <= pq) ? kTsAttnMaxFrames : pq;
}

hipError_t launch_tsattn(const float* x, const float* xs, const TsAttnWeights& w, float* ws, float* y, int b, int n, hipStream_t s) {
    if (b < 1 || n < 1 || n > tsattn_max_frames()) return hipErrorInvalidValue;
    const bool flash = tsattn_flash_min() > 0 && n >= tsattn_flash_min();
    const size_t attn_lds = (size_t)(256 + n + 256) * sizeof(float);       // per-query kernel: query row + one softmax row over the n frames of the clip
    if (!flash && attn_lds > 64 * 1024) {                                   // only when THAT kernel is the one launched: raise its dynamic LDS limit once per device
        static PerDeviceOnce attr;
        int dev = 0;
        GRK_TRY(current_device(&dev));
        const int max_n = per_query_max_frames();
        GRK_TRY(once_per_device(attr, dev, [max_n](int*) {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(temporal_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)((256 + max_n + 256) * sizeof(float)));
        }));
    }
    const size_t R = (size_t)b * n;
    float* qkv_t = ws;
    float* qkv_s = qkv_t + R * 3 * kE;
    float* xt = qkv_s + R * 3 * kE;
    float* xsp = xt + R * kE;
    float* mean = xsp + R * kE;
    float* logits = mean + (size_t)b * 2 * kE;
    float* yt = logits + (size_t)b * 2 * kE;
    float* ys = yt + R * kD;
    float* x1 = ys + R * kD;
    GRK_TRY(launch_gemm_nt_bias(x, w.qkv_t_w, w.qkv_t_b, qkv_t, (int)R, 3 * kE, kD, 3 * kE, s));
    GRK_TRY(launch_gemm_nt_bias(xs, w.qkv_s_w, w.qkv_s_b, qkv_s, (int)R, 3 * kE, kD + kF, 3 * kE, s));
    // clips of >= 384 frames: the blocked kernel (keys / values read once per 128 queries, fp32 matrix cores); shorter clips: one workgroup per query
    if (flash) {
        static PerDeviceOnce fattr;
        int dev = 0;
        GRK_TRY(current_device(&dev));
        GRK_TRY(once_per_device(fattr, dev, [](int*) {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(temporal_attn_flash_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kFlashLdsFloats * sizeof(float)));
        }));
        int cus = 0;
        GRK_TRY(device_cu_count(&cus));
        const int kparts = flash_key_parts(n, cus);
        float* part_o = yt;                                     // yt | ys | x1 (3 R x 3072 floats, all written later) hold the parts: kparts <= 8 need 8 064 R
        float* part_ml = part_o + (size_t)kparts * R * kE;
        GRK_TRY(launch_k(temporal_attn_flash_kernel, dim3((n + kFQ - 1) / kFQ, kH, b * kparts), dim3(64 * kFW), kFlashLdsFloats * sizeof(float), s, (const float*)qkv_t, xt, n,
                         kparts, part_o, part_ml));
        if (kparts > 1) GRK_TRY(launch_k(temporal_attn_combine_kernel, dim3((unsigned)R), dim3(256), 0, s, (const float*)part_o, (const float*)part_ml, xt, R, kparts));
    } else {
        GRK_TRY(launch_k(temporal_attn_kernel, dim3(n, kH, b), dim3(256), attn_lds, s, qkv_t, xt, n));
    }
    GRK_TRY(launch_k(spatial_attn_kernel, dim3((unsigned)R, kH), dim3(256), 0, s, qkv_s, xsp));
    {   // partial sums in yt (free until the fc_t GEMM writes it: nblk x 2000 <= n x 3072 floats per clip)
        const int nblk = (n + kMeanRows - 1) / kMeanRows;
        GRK_TRY(launch_k(gate_mean_partial_kernel, dim3((2 * kE + 255) / 256, b, nblk), dim3(256), 0, s, (const float*)xt, (const float*)xsp, yt, n, nblk));
        GRK_TRY(launch_k(gate_mean_final_kernel, dim3((2 * kE + 255) / 256, b), dim3(256), 0, s, (const float*)yt, mean, n, nblk));
    }
    GRK_TRY(launch_gemm_nt_bias(mean, w.ts_w, w.ts_b, logits, b, 2 * kE, 2 * kE, 2 * kE, s));
    const long total = (long)R * kE;
    GRK_TRY(launch_k(gate_apply_kernel, dim3((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), dim3(256), 0, s, logits, xt, xsp, n, total));
    GRK_TRY(launch_gemm_nt_bias(xt, w.fc_t_w, w.fc_t_b, yt, (int)R, kD, kE, kD, s));
    GRK_TRY(launch_gemm_nt_bias(xsp, w.fc_s_w, w.fc_s_b, ys, (int)R, kD, kE, kD, s));
    GRK_TRY(launch_k(residual_ln_kernel, dim3((unsigned)R), dim3(256), 0, s, x, yt, ys, w.n1_g, w.n1_b, x1));
    GRK_TRY(launch_k(jwff_ln_kernel, dim3((unsigned)R), dim3(256), 0, s, x1, w.jw1, w.jw2, w.n2_g, w.n2_b, y));
    return hipGetLastError();
}

}  // namespace grk
