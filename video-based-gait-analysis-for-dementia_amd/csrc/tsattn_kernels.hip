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

// The same attention for LONG clips, blocked (round 5).  The kernel above reads every key and value of the clip once per QUERY: at 10 000 frames that is
// 10 000 x 4 x 2 x 10 MB = 800 GB through L2 and 107 ms -- two thirds of the temporal branch that every rank of a BASELINE configs[3] job repeats after the
// all-gather (profiles/r05_temporal_phases.txt).  Here a workgroup owns 128 queries of one head (8 waves x 16); keys and values stream through LDS in blocks
// of 32 (read once per 128 queries), S = Q K^T and O += P V run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products and sums), the softmax
// is the running-maximum form: per query row m <- max(m, max_j s_j), O <- O exp(m_old - m) + sum_j exp(s_j - m) v_j, l likewise, out = O / l.  Same
// arithmetic as softmax(QK^T / sqrt(d)) V in fp32 up to the order of the sums; deterministic (no atomics, fixed order).
// A first version (4 waves, Q in LDS, K / V block loaded between two barriers: 19.4 ms at 10 000 frames) ran the matrix cores a fifth of the time -- one wave
// per SIMD, and nothing under the block load.  Now: the wave's Q fragment (its A operand of every S tile: 63 values per lane, 1 / sqrt(d) folded in) lives in
// REGISTERS, which frees 66 KB of LDS for eight waves per workgroup (two per SIMD: one wave's softmax and staging under the other's MFMAs); the NEXT block of
// K and V is requested into registers before the current block is computed and stored to LDS behind it.
// Row strides: 258 floats for K (258 = 2 mod 32: the 16 rows x 4 columns of an MFMA operand read fall on 64 different banks), 272 for V (16 mod 32: 4 rows x
// 16 columns likewise), 34 for the wave's own P tile (written in the accumulator layout, read back as the A operand of P V).
constexpr int kFW = 8, kFQ = 16 * kFW, kFK = 32, kFLd = 258, kFLdV = 272, kFLdP = 34;
constexpr int kFlashLdsFloats = kFK * kFLd + kFK * kFLdV + kFW * 16 * kFLdP;
constexpr int kFU = kFK * (kDh / 2), kFLoads = (kFU + 64 * kFW - 1) / (64 * kFW);       // float2 units of a K (or V) block; per thread
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float group16_max(float v) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(64 * kFW) void temporal_attn_flash_kernel(const float* __restrict__ qkv, float* __restrict__ xt, int n) {
    extern __shared__ float sm[];
    float* Ks = sm;
    float* Vs = Ks + kFK * kFLd;
    float* Ps = Vs + kFK * kFLdV + (threadIdx.x >> 6) * 16 * kFLdP;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4, wave = tid >> 6;
    const int q0 = blockIdx.x * kFQ, h = blockIdx.y, bi = blockIdx.z;
    const float* base = qkv + (size_t)bi * n * 3 * kE;
    const float scale = 1.0f / sqrtf((float)kDh);
    typedef float f2 __attribute__((ext_vector_type(2)));
    // this lane's part of the wave's Q fragment: query 16 wave + l15, columns 4 j + lq (columns 250, 251: zero; queries past the clip: zero rows, never stored)
    float qf[63];
    {
        const int q = q0 + wave * 16 + l15;
        const float* qrow = base + (size_t)(q < n ? q : 0) * 3 * kE + (0 * kH + h) * kDh + lq;
#pragma unroll
        for (int j = 0; j < 63; ++j) qf[j] = (q < n && 4 * j + lq < kDh) ? qrow[4 * j] * scale : 0.f;
    }
    // staging map of a K / V block: unit u = (row, float2 column); rows are 8-byte aligned (a head starts 1000 bytes into its row)
    for (int u = tid; u < kFK * (kFLd - kDh); u += 64 * kFW) Ks[(u / (kFLd - kDh)) * kFLd + kDh + u % (kFLd - kDh)] = 0.f;      // K columns 250 .. 257: zero for good
    for (int u = tid; u < kFK * (kFLdV - kDh); u += 64 * kFW) Vs[(u / (kFLdV - kDh)) * kFLdV + kDh + u % (kFLdV - kDh)] = 0.f;
    f2 kreg[kFLoads], vreg[kFLoads];
    auto request = [&](int k0) {                               // keys past the clip's end: zero rows (their scores are masked below)
        const float* kb = base + (size_t)k0 * 3 * kE + (1 * kH + h) * kDh;
        const float* vb = base + (size_t)k0 * 3 * kE + (2 * kH + h) * kDh;
#pragma unroll
        for (int i = 0; i < kFLoads; ++i) {
            const int u = i * 64 * kFW + tid, r = u / (kDh / 2), c2 = u - r * (kDh / 2);      // (recomputed per block: eight registers matter here)
            const bool ok = u < kFU && k0 + r < n;
            kreg[i] = ok ? *reinterpret_cast<const f2*>(kb + r * 3 * kE + 2 * c2) : f2{0.f, 0.f};
            vreg[i] = ok ? *reinterpret_cast<const f2*>(vb + r * 3 * kE + 2 * c2) : f2{0.f, 0.f};
        }
    };
    auto deposit = [&]() {
#pragma unroll
        for (int i = 0; i < kFLoads; ++i)
            if (i * 64 * kFW + tid < kFU) {
                const int u = i * 64 * kFW + tid, r = u / (kDh / 2), c2 = u - r * (kDh / 2);
                *reinterpret_cast<f2*>(Ks + r * kFLd + 2 * c2) = kreg[i];
                *reinterpret_cast<f2*>(Vs + r * kFLdV + 2 * c2) = vreg[i];
            }
    };
    float m_run[4], l_run[4];
    f32x4_t O[16];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
#pragma unroll
    for (int dt = 0; dt < 16; ++dt) O[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    request(0);
    deposit();
    for (int k0 = 0; k0 < n; k0 += kFK) {
        __syncthreads();                                       // block k0 is in LDS
        if (k0 + kFK < n) request(k0 + kFK);                   // the next block: in flight under this block's MFMAs
        // S (16 queries x 32 keys) = Q K^T: 63 k-steps of 4 (columns 250, 251 are zeros)
        f32x4_t s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        const float* k0row = Ks + l15 * kFLd + lq;
        const float* k1row = Ks + (16 + l15) * kFLd + lq;
#pragma unroll
        for (int j = 0; j < 63; ++j) {
            s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[j], k0row[4 * j], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[j], k1row[4 * j], s1, 0, 0, 0);
            if (j % 4 == 3) __builtin_amdgcn_sched_barrier(0);   // operand reads are hoisted at most 4 steps (8 registers) ahead, not 63
        }
        // lane: rows (queries) 4 lq + r, column (key) l15 of each tile; keys past the clip's end are -inf
        const bool in0 = k0 + l15 < n, in1 = k0 + 16 + l15 < n;
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a0 = in0 ? s0[r] : -INFINITY, a1 = in1 ? s1[r] : -INFINITY;
            const float mx = group16_max(fmaxf(a0, a1));
            const float m_new = fmaxf(m_run[r], mx);          // finite: key k0 exists
            alpha[r] = expf(m_run[r] - m_new);
            const float p0 = expf(a0 - m_new), p1 = expf(a1 - m_new);
            l_run[r] = l_run[r] * alpha[r] + group16_sum(p0 + p1);
            m_run[r] = m_new;
            Ps[(4 * lq + r) * kFLdP + l15] = p0;
            Ps[(4 * lq + r) * kFLdP + 16 + l15] = p1;
        }
#pragma unroll
        for (int dt = 0; dt < 16; ++dt) { O[dt][0] *= alpha[0]; O[dt][1] *= alpha[1]; O[dt][2] *= alpha[2]; O[dt][3] *= alpha[3]; }
        // O (16 queries x 256 columns) += P V: 8 k-steps of 4 keys; the P tile is this wave's own (LDS operations of a wave complete in order)
        const float* prow = Ps + l15 * kFLdP + lq;
#pragma unroll
        for (int kk = 0; kk < kFK / 4; ++kk) {
            const float a = prow[kk * 4];
            const float* vrow = Vs + (kk * 4 + lq) * kFLdV + l15;
#pragma unroll
            for (int dt = 0; dt < 16; ++dt) O[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, vrow[dt * 16], O[dt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);                 // (one k-step's 16 V reads in flight, not all eight)
        }
        __syncthreads();                                       // every wave has read block k0
        if (k0 + kFK < n) deposit();
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int q = q0 + wave * 16 + 4 * lq + r;
        if (q >= n) continue;
        const float inv = 1.0f / l_run[r];
        float* orow = xt + ((size_t)bi * n + q) * kE + h * kDh + l15;
#pragma unroll
        for (int dt = 0; dt < 16; ++dt)
            if (dt * 16 + l15 < kDh) orow[dt * 16] = O[dt][r] * inv;
    }
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
// Smallest clip on the blocked kernel (450 frames: 0.98 ms against 0.87 for the per-query kernel -- 32 workgroups; 10 000: 6 against 107); 0: never.
static int tsattn_flash_min() { return GRNET_AB(TSATTN_FLASH, 1024); }
static int per_query_max_frames() {                                          // what the per-query kernel's softmax row leaves of THIS device's LDS per workgroup
    int lds = 0;
    if (device_lds_per_block(&lds) != hipSuccess || lds <= 0) return 0;
    const long fit = (long)lds / (long)sizeof(float) - 512;
    return (int)std::min<long>(kTsAttnMaxFrames, fit > 0 ? fit : 0);
}
// The per-query kernel holds a softmax row over the clip in LDS; the blocked kernel's LDS use is fixed (85 KB), so with it on, the per-query limit binds
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
    // clips of >= 1024 frames: the blocked kernel (keys / values read once per 64 queries, fp32 matrix cores); shorter clips: one workgroup per query
    if (flash) {
        static PerDeviceOnce fattr;
        int dev = 0;
        GRK_TRY(current_device(&dev));
        GRK_TRY(once_per_device(fattr, dev, [](int*) {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(temporal_attn_flash_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(kFlashLdsFloats * sizeof(float)));
        }));
        GRK_TRY(launch_k(temporal_attn_flash_kernel, dim3((n + kFQ - 1) / kFQ, kH, b), dim3(64 * kFW), kFlashLdsFloats * sizeof(float), s, (const float*)qkv_t, xt, n));
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
