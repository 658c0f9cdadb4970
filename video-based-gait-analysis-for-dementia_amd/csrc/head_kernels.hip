// Everything on the per-frame path that is not a convolution: HR-module fuse sums, bilinear
// upsampling, PARE part-attention pooling, the regressor tail, SMPL linear blend skinning and
// the camera projection.  All of it is HBM/L2-bound byte work (< 0.1 % of the FLOPs):
// wavefront reductions, coalesced 16-byte accesses, one pass over each tensor.
#include "kernels.h"

namespace grk {

#define GRK_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return _e; } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------
// HR-module fuse output 0: y = relu(x_0 + up2(t_1) + up4(t_2) + up8(t_3)), nearest upsampling
// (reference: HighResolutionModule.forward, hrnet.py:258-265; nn.Upsample(nearest), hrnet.py:208).
// One workgroup per (frame, channel) plane, a thread per 4 consecutive pixels; 32-bit index arithmetic with one reciprocal multiply
// for the row (the first version walked a flat 64-bit index: two 64-bit divisions per thread, 20 us for a 6 MB output on the chain
// between two HR modules).  The sum order (x_0, then the addends as listed) and every rounding are unchanged.
__global__ __launch_bounds__(256) void fuse_sum_kernel(const SumArgs a) {
    const int HW = a.H * a.W, n = blockIdx.x / a.C, c = blockIdx.x - n * a.C;
    const float inv_w = 1.0f / (float)a.W;
    for (int pix = threadIdx.x * 4; pix < HW; pix += 1024) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < a.n_add; ++k) {
            const int sh = a.add_shift[k];
            if (sh == 0) {
                v += *reinterpret_cast<const f32x4*>(a.add[k] + ((size_t)n * a.add_ctot[k] + a.add_coff[k] + c) * HW + pix);
            } else {
                const int hs = a.H >> sh, ws = a.W >> sh;
                const float* ap = a.add[k] + ((size_t)n * a.add_ctot[k] + a.add_coff[k] + c) * (hs * ws);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int y = (int)(((float)(pix + r) + 0.5f) * inv_w), x = (pix + r) - y * a.W;      // exact for pix < 2^20
                    v[r] += ap[(y >> sh) * ws + (x >> sh)];
                }
            }
        }
        if (a.relu) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
        }
        *reinterpret_cast<f32x4*>(a.out + ((size_t)n * a.out_ctot + a.out_coff + c) * HW + pix) = v;
    }
}

hipError_t launch_fuse_sum(const SumArgs& a, hipStream_t s) {
    if ((a.H * a.W) % 4 != 0 || a.n_add < 1 || a.n_add > 4) return hipErrorInvalidValue;
    GRK_TRY(launch_k(fuse_sum_kernel, dim3(a.N * a.C), dim3(256), 0, s, a));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (hrnet.py:443):
// src = dst * (in-1)/(out-1); taps floor(src), min(floor+1, in-1).
__global__ __launch_bounds__(256) void bilinear2x_kernel(const float* __restrict__ in, float* __restrict__ out, int NC,
                                                           int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    const float sy = (float)(H - 1) / (float)(Ho - 1), sx = (float)(W - 1) / (float)(Wo - 1);
    const long total = (long)NC * Ho * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo);
        const long t = i / Wo;
        const int y = (int)(t % Ho);
        const long nc = t / Ho;
        const float fy = sy * y, fx = sx * x;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = fy - y0, lx = fx - x0;
        const float* p = in + nc * H * W;
        // explicit roundings: the grid-stride loop is unrolled differently for different sizes, and
        // compiler-chosen fma contraction would make a frame's result depend on the batch it is in
        const float wx0 = 1.f - lx, wy0 = 1.f - ly;
        const float top = __fmaf_rn(p[y0 * W + x1], lx, __fmul_rn(p[y0 * W + x0], wx0));
        const float bot = __fmaf_rn(p[y1 * W + x1], lx, __fmul_rn(p[y1 * W + x0], wx0));
        out[i] = __fmaf_rn(bot, ly, __fmul_rn(top, wy0));
    }
}

// Same arithmetic per element, 4 adjacent output pixels of a row per thread: one 16-byte store instead of four 4-byte ones
// (the 256-channel 28 -> 56 launch writes 51 MB and sits on the critical path in front of the last upsample convolution).
__global__ __launch_bounds__(256) void bilinear2x_x4_kernel(const float* __restrict__ in, float* __restrict__ out, int NC, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W, Wq = Wo >> 2;
    const float sy = (float)(H - 1) / (float)(Ho - 1), sx = (float)(W - 1) / (float)(Wo - 1);
    const long total = (long)NC * Ho * Wq;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xq = (int)(i % Wq);
        const long t = i / Wq;
        const int y = (int)(t % Ho);
        const long nc = t / Ho;
        const float fy = sy * y;
        const int y0 = (int)fy, y1 = min(y0 + 1, H - 1);
        const float ly = fy - y0, wy0 = 1.f - ly;
        const float* p0 = in + nc * H * W + y0 * W;
        const float* p1 = in + nc * H * W + y1 * W;
        float r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int x = xq * 4 + k;
            const float fx = sx * x;
            const int x0 = (int)fx, x1 = min(x0 + 1, W - 1);
            const float lx = fx - x0, wx0 = 1.f - lx;
            const float top = __fmaf_rn(p0[x1], lx, __fmul_rn(p0[x0], wx0));
            const float bot = __fmaf_rn(p1[x1], lx, __fmul_rn(p1[x0], wx0));
            r[k] = __fmaf_rn(bot, ly, __fmul_rn(top, wy0));
        }
        *reinterpret_cast<float4*>(out + (nc * Ho + y) * Wo + xq * 4) = make_float4(r[0], r[1], r[2], r[3]);
    }
}

// One workgroup per (frame, channel) plane of at most 32 x 32 inputs: the plane goes to LDS with coalesced 16-byte loads, every tap is
// an LDS read, the output leaves as 16-byte stores with the lanes walking a row.  Same arithmetic per element as the two kernels above
// (bit-identical results).  The upsampling launches sit on the critical path of the upsample heads, in front of their convolutions:
// 28 -> 56 with 256 channels writes 51 MB and took 50 us with 16 gathered global loads and two 64-bit divisions per thread (1.3 TB/s).
template <bool X4>
__global__ __launch_bounds__(256) void bilinear2x_plane_kernel(const float* __restrict__ in, float* __restrict__ out, int H, int W) {
    __shared__ __align__(16) float pl[32 * 32];
    const int Ho = 2 * H, Wo = 2 * W, HW = H * W;
    const float sy = (float)(H - 1) / (float)(Ho - 1), sx = (float)(W - 1) / (float)(Wo - 1);
    const float* p = in + (size_t)blockIdx.x * HW;
    float* o = out + (size_t)blockIdx.x * Ho * Wo;
    if (X4) for (int u = threadIdx.x; u < HW / 4; u += 256) *reinterpret_cast<float4*>(pl + 4 * u) = *reinterpret_cast<const float4*>(p + 4 * u);
    else for (int u = threadIdx.x; u < HW; u += 256) pl[u] = p[u];
    __syncthreads();
    const int Wq = X4 ? Wo >> 2 : Wo;                                           // units per output row: 4 pixels or 1
    const float inv_wq = 1.0f / (float)Wq;
    for (int i = threadIdx.x; i < Ho * Wq; i += 256) {
        const int y = (int)(((float)i + 0.5f) * inv_wq), xq = i - y * Wq;      // exact for i < 2^20
        const float fy = sy * y;
        const int y0 = (int)fy, y1 = min(y0 + 1, H - 1);
        const float ly = fy - y0, wy0 = 1.f - ly;
        const float* p0 = pl + y0 * W;
        const float* p1 = pl + y1 * W;
        float r[4];
#pragma unroll
        for (int k = 0; k < (X4 ? 4 : 1); ++k) {
            const int x = X4 ? xq * 4 + k : xq;
            const float fx = sx * x;
            const int x0 = (int)fx, x1 = min(x0 + 1, W - 1);
            const float lx = fx - x0, wx0 = 1.f - lx;
            const float top = __fmaf_rn(p0[x1], lx, __fmul_rn(p0[x0], wx0));
            const float bot = __fmaf_rn(p1[x1], lx, __fmul_rn(p1[x0], wx0));
            r[k] = __fmaf_rn(bot, ly, __fmul_rn(top, wy0));
        }
        if (X4) *reinterpret_cast<float4*>(o + y * Wo + xq * 4) = make_float4(r[0], r[1], r[2], r[3]);
        else o[i] = r[0];
    }
}

hipError_t launch_bilinear2x(const float* in, float* out, int N, int C, int H, int W, hipStream_t s) {
    if (H <= 32 && W <= 32) {
        if ((2 * W) % 4 == 0 && (H * W) % 4 == 0) GRK_TRY(launch_k(bilinear2x_plane_kernel<true>, dim3(N * C), dim3(256), 0, s, in, out, H, W));
        else GRK_TRY(launch_k(bilinear2x_plane_kernel<false>, dim3(N * C), dim3(256), 0, s, in, out, H, W));
        return hipGetLastError();
    }
    const bool x4 = (2 * W) % 4 == 0;
    const long total = (long)N * C * 4 * H * W / (x4 ? 4 : 1);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (x4) GRK_TRY(launch_k(bilinear2x_x4_kernel, dim3(blocks), dim3(256), 0, s, in, out, N * C, H, W));
    else GRK_TRY(launch_k(bilinear2x_kernel, dim3(blocks), dim3(256), 0, s, in, out, N * C, H, W));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Crop + normalise (the step right before the path; reference: Inference.__getitem__ lib/dataset/inference.py:71-87
// -> get_single_image_crop_demo / generate_patch_image_cv / gen_trans_from_patch_cv lib/data_utils/img_utils.py:252-285,
// 90-113,54-88 with rot = 0, then ToTensor + Normalize :355-363).  For a box [cx,cy,w,h] and scale s the affine map
// sends the box of size (w*s, h*s) centred at (cx,cy) to the 224x224 patch: destination pixel (u,v) samples the source
// at x = (u-112)*w*s/224 + cx, y = (v-112)*h*s/224 + cy (cv2.warpAffine: integer coordinates are pixel centres),
// bilinear, constant border 0, result rounded to uint8 as warpAffine returns it; then /255, (x-mean)/std, HWC -> CHW.
// OpenCV's 1/32-pixel fixed-point interpolation is third-party arithmetic absent offline: parity with it is UNPINNED.
__global__ __launch_bounds__(256) void crop_normalise_kernel(const unsigned char* __restrict__ img, int H, int W, int per_image,
                                                             const float* __restrict__ bbox, float scale, int bgr,
                                                             float* __restrict__ out) {
    const int n = blockIdx.y;
    const unsigned char* src = img + (size_t)(per_image ? n : 0) * H * W * 3;
    const float cx = bbox[n * 4 + 0], cy = bbox[n * 4 + 1], bw = bbox[n * 4 + 2] * scale, bh = bbox[n * 4 + 3] * scale;
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 224 * 224; i += gridDim.x * 256) {
        const int v = i / 224, u = i - v * 224;
        const float x = (u - 112.f) * (bw / 224.f) + cx, y = (v - 112.f) * (bh / 224.f) + cy;
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = (int)xf, y0 = (int)yf;
        const float ax = x - xf, ay = y - yf;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int cs = bgr ? 2 - c : c;
            auto px = [&](int yy, int xx) -> float {
                return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (float)src[((size_t)yy * W + xx) * 3 + cs] : 0.f;
            };
            const float top = px(y0, x0) * (1.f - ax) + px(y0, x0 + 1) * ax;
            const float bot = px(y0 + 1, x0) * (1.f - ax) + px(y0 + 1, x0 + 1) * ax;
            const float val = floorf(top * (1.f - ay) + bot * ay + 0.5f);          // warpAffine returns uint8
            out[((size_t)n * 3 + c) * (224 * 224) + i] = (fminf(fmaxf(val, 0.f), 255.f) / 255.f - mean[c]) / stdv[c];
        }
    }
}

// The same crop with OpenCV's own arithmetic (cv2.warpAffine, INTER_LINEAR, BORDER_CONSTANT 0, 8-bit source; OpenCV 4.1.2,
// requirements.txt:8): the INVERSE affine map M (6 doubles per frame, computed on the host exactly as generate_patch_image_cv /
// getAffineTransform / warpAffine's own inversion do) is evaluated in FIXED POINT --
//   adelta[x] = cvRound(M0*x*1024), bdelta[x] = cvRound(M3*x*1024)                 (AB_BITS = 10, cvRound = round half to even)
//   X0 = cvRound((M1*y + M2)*1024) + 16,  Y0 = cvRound((M4*y + M5)*1024) + 16       (round_delta = 1024/32/2)
//   X = (X0 + adelta[x]) >> 5, Y = (Y0 + bdelta[x]) >> 5                            (INTER_BITS = 5: 1/32-pixel positions)
//   sx = X >> 5, sy = Y >> 5 (saturated to int16), ax = X & 31, ay = Y & 31
// -- and the four taps are blended with the 15-bit table weights 32*(32-ax)(32-ay) ... (they add up to 32768 exactly; the one
// entry OpenCV patches, ax = ay = 0, yields the pixel itself either way), result = (sum + 16384) >> 15; taps outside the image are 0.
// Integer arithmetic end to end: the uint8 patch is bit-identical to the oracle's.  cv2 itself is absent offline, so agreement
// with a real OpenCV build is argued from its source, not measured.
__device__ __forceinline__ int cv_round(double v) { return (int)__builtin_rint(v); }
// one destination pixel (x, y) of warpAffine from an interleaved 8-bit image of H x W x 3: the three channel values
__device__ __forceinline__ void cv_warp_px(const unsigned char* __restrict__ src, int H, int W, const double* m, int x, int y, int v[3]) {
    const int X0 = cv_round((m[1] * y + m[2]) * 1024.0) + 16, Y0 = cv_round((m[4] * y + m[5]) * 1024.0) + 16;
    const int X = (X0 + cv_round(m[0] * x * 1024.0)) >> 5, Y = (Y0 + cv_round(m[3] * x * 1024.0)) >> 5;
    const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);
    const int ax = X & 31, ay = Y & 31;
    const int w00 = (32 - ax) * (32 - ay), w01 = ax * (32 - ay), w10 = (32 - ax) * ay, w11 = ax * ay;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        auto px = [&](int yy, int xx) -> int { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? (int)src[((size_t)yy * W + xx) * 3 + c] : 0; };
        const int sum = px(sy, sx) * w00 + px(sy, sx + 1) * w01 + px(sy + 1, sx) * w10 + px(sy + 1, sx + 1) * w11;
        v[c] = (sum * 32 + 16384) >> 15;
    }
}
// maps: per frame `mstride` doubles -- [0..5] the inverse affine map of the (first) warp; with mstride = 10 also [6] iw, [7] ih, [8] tx,
// [9] ty: iw > 0 is the reference's TWO-warp crop of a non-square box (img_utils.py:97-106): the first warp makes an iw x ih 8-bit
// image, the second moves it by (-tx, -ty) into the patch.  The intermediate image is never stored: a patch pixel blends the (up to
// four) intermediate pixels its second warp samples, each computed from the source on the fly -- same integers as two passes.
__global__ __launch_bounds__(256) void crop_normalise_cv_kernel(const unsigned char* __restrict__ img, int H, int W, int per_image,
                                                                const double* __restrict__ maps, int mstride, int bgr, float* __restrict__ out) {
    const int n = blockIdx.y;
    const unsigned char* src = img + (size_t)(per_image ? n : 0) * H * W * 3;
    double m[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) m[k] = maps[(size_t)n * mstride + k];
    const int iw = mstride >= 10 ? (int)maps[(size_t)n * mstride + 6] : 0, ih = mstride >= 10 ? (int)maps[(size_t)n * mstride + 7] : 0;
    const double t2[6] = {1.0, 0.0, iw > 0 ? maps[(size_t)n * mstride + 8] : 0.0, 0.0, 1.0, iw > 0 ? maps[(size_t)n * mstride + 9] : 0.0};
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 224 * 224; i += gridDim.x * 256) {
        const int y = i / 224, x = i - y * 224;
        int val[3];
        if (iw <= 0) {
            cv_warp_px(src, H, W, m, x, y, val);
        } else {
            const int X0 = cv_round((t2[1] * y + t2[2]) * 1024.0) + 16, Y0 = cv_round((t2[4] * y + t2[5]) * 1024.0) + 16;
            const int X = (X0 + cv_round(t2[0] * x * 1024.0)) >> 5, Y = (Y0 + cv_round(t2[3] * x * 1024.0)) >> 5;
            const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);
            const int ax = X & 31, ay = Y & 31;
            const int wt[4] = {(32 - ax) * (32 - ay), ax * (32 - ay), (32 - ax) * ay, ax * ay};
            int sum[3] = {0, 0, 0};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ix = sx + (t & 1), iy = sy + (t >> 1);
                if (wt[t] != 0 && ix >= 0 && ix < iw && iy >= 0 && iy < ih) {
                    int v[3];
                    cv_warp_px(src, H, W, m, ix, iy, v);
                    sum[0] += v[0] * wt[t]; sum[1] += v[1] * wt[t]; sum[2] += v[2] * wt[t];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) val[c] = (sum[c] * 32 + 16384) >> 15;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) out[((size_t)n * 3 + c) * (224 * 224) + i] = ((float)val[bgr ? 2 - c : c] / 255.f - mean[c]) / stdv[c];
    }
}

hipError_t launch_crop_normalise_cv(const unsigned char* img, int H, int W, int per_image, const double* maps, int mstride, int bgr, float* out, int N,
                                    hipStream_t s) {
    if (mstride != 6 && mstride != 10) return hipErrorInvalidValue;
    GRK_TRY(launch_k(crop_normalise_cv_kernel, dim3(49, N), dim3(256), 0, s, img, H, W, per_image, maps, mstride, bgr, out));
    return hipSuccess;
}

hipError_t launch_crop_normalise(const unsigned char* img, int H, int W, int per_image, const float* bbox, float scale, int bgr,
                                 float* out, int N, hipStream_t s) {
    GRK_TRY(launch_k(crop_normalise_kernel, dim3(49, N), dim3(256), 0, s, img, H, W, per_image, bbox, scale, bgr, out));
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------
// Part attention (KeypointAttention.forward, keypoint_attention.py:42-48; called twice with the
// same heat-maps, pare.py:331-332): softmax over the 3136 positions of each (frame, joint), then
// out[n,c,j] = sum_p prob[n,j,p] * feat[n,c,p], both feature maps (128 + 64 channels) in one launch.
// The pooling is a GEMM per frame -- out[c][j] = sum_p feat[c][p] * prob[j][p], M = 192 channels, N = 24 joints, K = 3136 positions --
// and runs on the fp32 matrix cores: workgroup = (frame n, 96 channels, one of kPoolSplit position ranges) = 6 waves, wave = one
// 16-channel row tile x both 16-joint column tiles (joints 24..31 are zero columns).  The workgroup first builds its range's
// range's exp(h - range max) ONCE, in LDS (32 rows of 448 positions, rows 24..31 zero); then, with NCHW keeping a channel's
// positions contiguous, a lane's 16-byte load IS its A operand of four consecutive k-steps (row = channel l15, k-step s <-> position
// p0 + 4 lq + s) and the same reading of the LDS rows is the B operand.  Every wave writes its partial sums; head_tail_kernel adds the
// kPoolSplit partials in a fixed order.  (Round 2's vector-ALU version staged 64-position tiles in LDS and took 44 us at 16 frames:
// 8 LDS reads per 12 FMAs; a first matrix-core version with every wave rebuilding the probabilities from global memory took 27.)
constexpr int kPoolChunk = 448, kPoolStride = kPoolChunk + 4;   // positions per range (3136 / kPoolSplit) and the LDS row stride
__global__ __launch_bounds__(384) void attn_pool_kernel(const float* __restrict__ heat, int heat_ctot, float* __restrict__ stats,
                                                          const float* __restrict__ featA, int CA, const float* __restrict__ featB,
                                                          int CB, float* __restrict__ part, int P) {
    __shared__ __align__(16) float prob[32 * kPoolStride];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, lq = lane >> 4;
    const int pbeg = blockIdx.z * kPoolChunk;
    // the range's heat-map rows -> LDS, then per row (four rows per wave): max over the range, exp(h - max) in place, sum.  The softmax
    // over all 3136 positions is finished by head_tail_kernel from the kPoolSplit (max, sum) pairs -- the usual online-softmax merge --
    // so no pass over the heat maps runs in front of this kernel (softmax_stats_kernel was 8-10 us on the step's serial tail).
    for (int u = tid; u < 32 * (kPoolChunk / 4); u += 384) {
        const int j = u / (kPoolChunk / 4), q = u - j * (kPoolChunk / 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j < 24) v = *reinterpret_cast<const f32x4*>(heat + ((size_t)n * heat_ctot + 1 + j) * P + pbeg + 4 * q);      // channel 0 = background
        *reinterpret_cast<f32x4*>(prob + j * kPoolStride + 4 * q) = v;
    }
    __syncthreads();
    for (int j = (tid >> 6) * 4; j < (tid >> 6) * 4 + 4; ++j) {
        float* row = prob + j * kPoolStride;
        float hv[kPoolChunk / 64], m = -INFINITY;
#pragma unroll
        for (int i = 0; i < kPoolChunk / 64; ++i) { hv[i] = row[lane + 64 * i]; m = fmaxf(m, hv[i]); }
        m = wave_max(m);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < kPoolChunk / 64; ++i) { const float e = expf(hv[i] - m); row[lane + 64 * i] = e; sum += e; }
        sum = wave_sum(sum);
        if (lane == 0 && blockIdx.y == 0) {
            float* st = stats + (((size_t)n * kPoolSplit + blockIdx.z) * 24 + j) * 2;      // [n][range][joint][max, sum]
            st[0] = m;
            st[1] = sum;
        }
    }
    const int ct = blockIdx.y * 6 + (tid >> 6);                         // row tile: channels 16 ct .. 16 ct + 15 of [featA | featB]
    const int c = ct * 16 + l15;
    const float* frow = (c < CA ? featA + ((size_t)n * CA + c) * P : featB + ((size_t)n * CB + (c - CA)) * P) + pbeg + 4 * lq;
    // four groups of 16 positions in flight per wave (a group is ~0.15 us of work, an L2 / HBM round trip 1-2 us)
    f32x4 fa[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) fa[g] = *reinterpret_cast<const f32x4*>(frow + 16 * g);
    __syncthreads();
    const float* b0 = prob + l15 * kPoolStride + 4 * lq;
    const float* b1 = prob + (16 + l15) * kPoolStride + 4 * lq;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < kPoolChunk; p0 += 64) {
        const int pn = p0 + 64 < kPoolChunk ? p0 + 64 : p0;             // the last round re-requests itself
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 f = fa[g];
            fa[g] = *reinterpret_cast<const f32x4*>(frow + pn + 16 * g);
            const f32x4 u = *reinterpret_cast<const f32x4*>(b0 + p0 + 16 * g), v = *reinterpret_cast<const f32x4*>(b1 + p0 + 16 * g);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[k], u[k], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(f[k], v[k], acc1, 0, 0, 0);
            }
        }
    }
    // D[row = 4 lq + r][col = l15]: channel 16 ct + 4 lq + r, joints l15 and 16 + l15
    float* o = part + (((size_t)n * kPoolSplit + blockIdx.z) * (CA + CB) + ct * 16 + 4 * lq) * 24;      // [n][split][192][24]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        o[r * 24 + l15] = acc0[r];
        if (l15 < 8) o[r * 24 + 16 + l15] = acc1[r];
    }
}

size_t softmax_pool_ws_floats(int N) { return (size_t)N * kPoolStatsFloats + (size_t)N * kPoolSplit * 192 * 24; }

hipError_t launch_softmax_pool(const float* heat, int heat_ctot, const float* featA, int CA, const float* featB, int CB,
                               float* outA, float* outB, float* stats_ws, int N, int P, hipStream_t s) {
    (void)outA; (void)outB;                                  // written by head_tail_kernel from the partial sums
    if (CA != 128 || CB != 64 || P != kPoolChunk * kPoolSplit) return hipErrorInvalidValue;      // 56 x 56 = 7 ranges x 7 rounds of 4 groups of 16 positions
    float* part = stats_ws + (size_t)N * kPoolStatsFloats;
    GRK_TRY(launch_k(attn_pool_kernel, dim3(N, (CA + CB) / 96, kPoolSplit), dim3(384), 0, s, heat, heat_ctot, stats_ws, featA, CA, featB,
                       CB, part, P));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Regressor tail, one workgroup per frame:
//   pose[j][o] = sum_c plf[c][j] * Wp[o][c][j]                     (LocallyConnected2d, locallyconnected2d.py:43-46)
//   shape / cam = Linear(flatten(csf)), flat index c*24+j          (pare.py:342,365-366)
//   rot6d -> rotmat (Gram-Schmidt, geometry.py:395-410), rotmat -> quaternion -> axis-angle with
//   NaN -> 0 (geometry.py:68-97,159-293), theta = [cam, aa, shape] (pare.py:79).
__device__ __forceinline__ void rot6d_to_rotmat_dev(const float* x, float* R) {
    // x viewed as (3,2): a1 = x[0],x[2],x[4]; a2 = x[1],x[3],x[5]; R columns = b1 b2 b3
    const float a1x = x[0], a1y = x[2], a1z = x[4], a2x = x[1], a2y = x[3], a2z = x[5];
    const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-6f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float d = b1x * a2x + b1y * a2y + b1z * a2z;
    const float ux = a2x - d * b1x, uy = a2y - d * b1y, uz = a2z - d * b1z;
    const float n2 = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-6f);
    const float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
    const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
    R[0] = b1x; R[1] = b2x; R[2] = b3x;
    R[3] = b1y; R[4] = b2y; R[5] = b3y;
    R[6] = b1z; R[7] = b2z; R[8] = b3z;
}

__device__ __forceinline__ void rotmat_to_aa_dev(const float* R, float* aa) {
    // m = R^T (geometry.py:243): m[i][j] = R[j][i]
    const float m00 = R[0], m01 = R[3], m02 = R[6];
    const float m10 = R[1], m11 = R[4], m12 = R[7];
    const float m20 = R[2], m21 = R[5], m22 = R[8];
    float q0, q1, q2, q3, t;
    if (m22 < 1e-6f) {
        if (m00 > m11) { t = 1.f + m00 - m11 - m22; q0 = m12 - m21; q1 = t; q2 = m01 + m10; q3 = m20 + m02; }
        else           { t = 1.f - m00 + m11 - m22; q0 = m20 - m02; q1 = m01 + m10; q2 = t; q3 = m12 + m21; }
    } else {
        if (m00 < -m11) { t = 1.f - m00 - m11 + m22; q0 = m01 - m10; q1 = m20 + m02; q2 = m12 + m21; q3 = t; }
        else            { t = 1.f + m00 + m11 + m22; q0 = t; q1 = m12 - m21; q2 = m20 - m02; q3 = m01 - m10; }
    }
    const float sc = 0.5f / sqrtf(t);
    q0 *= sc; q1 *= sc; q2 *= sc; q3 *= sc;
    const float s2 = q1 * q1 + q2 * q2 + q3 * q3;
    const float sn = sqrtf(s2);
    const float two_theta = 2.f * (q0 < 0.f ? atan2f(-sn, -q0) : atan2f(sn, q0));
    const float k = s2 > 0.f ? two_theta / sn : 2.f;
    float a0 = q1 * k, a1 = q2 * k, a2 = q3 * k;
    aa[0] = isnan(a0) ? 0.f : a0;
    aa[1] = isnan(a1) ? 0.f : a1;
    aa[2] = isnan(a2) ? 0.f : a2;
}

// FROM_PARTS: plf / csf are produced here from the pooling partials (first head pass); otherwise they are INPUTS (the second
// head pass of the use_gait_feat branch, grnet.py:165, and the single-op parity hook).
// zstats != nullptr: the partials are sums of feat * exp(h - range max) and zstats holds each range's (max, sum of exp) per joint; the
// softmax over all positions is finished here: weight of range z = exp(max_z - max over ranges), divided by the weighted sum of sums.
template <bool FROM_PARTS>
__global__ __launch_bounds__(256) void head_tail_kernel(const float* __restrict__ part, const float* __restrict__ zstats, float* __restrict__ plf, float* __restrict__ csf, TailWeights w,
                                                          float* __restrict__ rot6d, float* __restrict__ shape,
                                                          float* __restrict__ cam, float* __restrict__ rotmat,
                                                          float* __restrict__ theta) {
    __shared__ float s_plf[128 * 24];
    __shared__ __align__(16) float s_csf[64 * 24];
    __shared__ float s_pose[24 * 6];
    __shared__ float s_sc[13];
    __shared__ float s_pp[4][144];
    __shared__ float s_sp[13][8];
    __shared__ float s_zw[kPoolSplit + 1][24];                // range weights; row kPoolSplit: 1 / (weighted sum of the ranges' sums)
    const int n = blockIdx.x, tid = threadIdx.x;
    if constexpr (FROM_PARTS) {
        if (tid < 24) {
            float m[kPoolSplit], sm[kPoolSplit], M = -INFINITY, S = 0.f;
#pragma unroll
            for (int z = 0; z < kPoolSplit; ++z) {
                m[z] = zstats ? zstats[(((size_t)n * kPoolSplit + z) * 24 + tid) * 2] : 0.f;
                sm[z] = zstats ? zstats[(((size_t)n * kPoolSplit + z) * 24 + tid) * 2 + 1] : 0.f;
                M = fmaxf(M, m[z]);
            }
#pragma unroll
            for (int z = 0; z < kPoolSplit; ++z) {
                const float wz = zstats ? expf(m[z] - M) : 1.f;
                s_zw[z][tid] = wz;
                S += sm[z] * wz;
            }
            s_zw[kPoolSplit][tid] = zstats ? 1.f / S : 1.f;
        }
        __syncthreads();
    }
    // add the pixel-range partials of the pooling in a fixed order; 6 elements x 7 partials of loads in flight per thread
    // (one block per frame: nothing else hides the L2 latency of this kernel, which sits on the critical path)
    static_assert((192 * 24) % (256 * 6) == 0, "partials loop");
    if constexpr (!FROM_PARTS) {
        for (int e = tid; e < 128 * 24; e += 256) s_plf[e] = plf[(size_t)n * 128 * 24 + e];
        for (int e = tid; e < 64 * 24; e += 256) s_csf[e] = csf[(size_t)n * 64 * 24 + e];
    } else
    for (int e0 = tid; e0 < 192 * 24; e0 += 256 * 6) {
        float v[6][kPoolSplit];
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
            for (int sp = 0; sp < kPoolSplit; ++sp) v[u][sp] = part[((size_t)n * kPoolSplit + sp) * (192 * 24) + e0 + u * 256];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int e = e0 + u * 256;
            const int j = e % 24;
            float acc = 0.f;
#pragma unroll
            for (int sp = 0; sp < kPoolSplit; ++sp) acc += v[u][sp] * s_zw[sp][j];
            acc *= s_zw[kPoolSplit][j];
            if (e < 128 * 24) { s_plf[e] = acc; plf[(size_t)n * 128 * 24 + e] = acc; }
            else { s_csf[e - 128 * 24] = acc; csf[(size_t)n * 64 * 24 + e - 128 * 24] = acc; }
        }
    }
    __syncthreads();
    // 154 KB of weights per frame (the same for every frame: L2 hits) and 38 k multiply-adds: this kernel is one workgroup per frame on
    // the critical path of the step, so what counts is how many of its loads are in flight at once.  Threads 0..143: one 16-byte column
    // group of pose_w's [c][j*6+o] rows x one quarter of the 128 channels (32 independent 16-byte loads each); threads 144..247: one
    // eighth of one of the 13 shape / cam rows (48 independent 16-byte loads each); partial sums meet in LDS in a fixed order.
    // (Round 2: 144 threads x 128 dependent-issue scalar loads, then 13 wave-wide dot products: 44 us at 16 frames.)
    if (tid < 144) {
        const int g = tid % 36, cq = tid / 36;
        const f32x4* wp = reinterpret_cast<const f32x4*>(w.pose_w) + (size_t)(cq * 32) * 36 + g;
        const float* pl = s_plf + cq * 32 * 24;
        const int ja = (4 * g) / 6, jb = (4 * g + 1) / 6, jc = (4 * g + 2) / 6, jd = (4 * g + 3) / 6;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
        for (int c = 0; c < 32; ++c) {
            const f32x4 wv = wp[c * 36];
            acc[0] += pl[c * 24 + ja] * wv[0];
            acc[1] += pl[c * 24 + jb] * wv[1];
            acc[2] += pl[c * 24 + jc] * wv[2];
            acc[3] += pl[c * 24 + jd] * wv[3];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) s_pp[cq][4 * g + k] = acc[k];
    } else if (tid < 144 + 13 * 8) {
        const int o = (tid - 144) >> 3, seg = (tid - 144) & 7;
        const float* wr = (o < 10 ? w.shape_w + (size_t)o * 1536 : w.cam_w + (size_t)(o - 10) * 1536) + seg * 192;
        const float* cs = s_csf + seg * 192;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
        for (int k = 0; k < 48; ++k) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + 4 * k), x = *reinterpret_cast<const f32x4*>(cs + 4 * k);
            acc += x * wv;
        }
        s_sp[o][seg] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    __syncthreads();
    if (tid < 144) {
        const float acc = (s_pp[0][tid] + s_pp[1][tid]) + (s_pp[2][tid] + s_pp[3][tid]);
        s_pose[tid] = acc;                                   // tid = j * 6 + o
        rot6d[(size_t)n * 144 + tid] = acc;
    } else if (tid < 157) {
        const int o = tid - 144;
        float acc = 0.f;
#pragma unroll
        for (int seg = 0; seg < 8; ++seg) acc += s_sp[o][seg];
        s_sc[o] = acc + (o < 10 ? w.shape_b[o] : w.cam_b[o - 10]);
    }
    __syncthreads();
    if (tid < 24) {
        float R[9], aa[3];
        rot6d_to_rotmat_dev(&s_pose[tid * 6], R);
#pragma unroll
        for (int k = 0; k < 9; ++k) rotmat[(size_t)n * 216 + tid * 9 + k] = R[k];
        rotmat_to_aa_dev(R, aa);
        theta[(size_t)n * 85 + 3 + tid * 3 + 0] = aa[0];
        theta[(size_t)n * 85 + 3 + tid * 3 + 1] = aa[1];
        theta[(size_t)n * 85 + 3 + tid * 3 + 2] = aa[2];
    }
    if (tid >= 64 && tid < 74) {
        const int o = tid - 64;
        shape[(size_t)n * 10 + o] = s_sc[o];
        theta[(size_t)n * 85 + 75 + o] = s_sc[o];
    }
    if (tid >= 128 && tid < 131) {
        const int o = tid - 128;
        cam[(size_t)n * 3 + o] = s_sc[10 + o];
        theta[(size_t)n * 85 + o] = s_sc[10 + o];
    }
}

hipError_t launch_head_tail(const float* pool_ws, bool range_stats, float* plf, float* csf, TailWeights w, float* rot6d, float* shape, float* cam,
                            float* rotmat, float* theta, int N, hipStream_t s) {
    GRK_TRY(launch_k(head_tail_kernel<true>, dim3(N), dim3(256), 0, s, pool_ws + (size_t)N * kPoolStatsFloats, range_stats ? pool_ws : (const float*)nullptr, plf, csf,
                     w, rot6d, shape, cam, rotmat, theta));
    return hipGetLastError();
}

hipError_t launch_head_tail_from_feats(const float* plf, const float* csf, TailWeights w, float* rot6d, float* shape, float* cam, float* rotmat,
                                       float* theta, int N, hipStream_t s) {
    GRK_TRY(launch_k(head_tail_kernel<false>, dim3(N), dim3(256), 0, s, (const float*)nullptr, (const float*)nullptr, const_cast<float*>(plf), const_cast<float*>(csf), w,
                     rot6d, shape, cam, rotmat, theta));
    return hipGetLastError();
}

// Single-op hooks of the geometry tail (parity tests feed the reference's edge-case goldens straight to the device functions).
__global__ __launch_bounds__(256) void rot6d_to_rotmat_kernel(const float* __restrict__ x, float* __restrict__ R, int m) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    float xi[6], r[9];
#pragma unroll
    for (int k = 0; k < 6; ++k) xi[k] = x[(size_t)i * 6 + k];
    rot6d_to_rotmat_dev(xi, r);
#pragma unroll
    for (int k = 0; k < 9; ++k) R[(size_t)i * 9 + k] = r[k];
}
__global__ __launch_bounds__(256) void rotmat_to_aa_kernel(const float* __restrict__ R, float* __restrict__ aa, int m) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    float r[9], a[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = R[(size_t)i * 9 + k];
    rotmat_to_aa_dev(r, a);
#pragma unroll
    for (int k = 0; k < 3; ++k) aa[(size_t)i * 3 + k] = a[k];
}
hipError_t launch_rot6d_to_rotmat(const float* x, float* R, int m, hipStream_t s) {
    GRK_TRY(launch_k(rot6d_to_rotmat_kernel, dim3((m + 255) / 256), dim3(256), 0, s, x, R, m));
    return hipGetLastError();
}
hipError_t launch_rotmat_to_aa(const float* R, float* aa, int m, hipStream_t s) {
    GRK_TRY(launch_k(rotmat_to_aa_kernel, dim3((m + 255) / 256), dim3(256), 0, s, R, aa, m));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// SMPL linear blend skinning (published algorithm as implemented by smplx 0.1.26, SURVEY A.7;
// reference call sites lib/models/smpl.py:108-130,157-162), batched over the frames of a call.
//   kernel 1 (one wave per frame): rest joints J = J_template + J_shapedirs.betas (the joint
//     regressor is linear, so it is applied to the tables once at load), the 24-joint kinematic
//     chain, skinning matrices A_j = G_j - [0 | G_j.J_j] as 3x4 rows, posed joints; and the frame's
//     row of the blend-shape GEMM: [ (R_1..23 - I) (207) | betas (10) | 1 | 0 0 ].
//   kernel 2 (fp32 matrix cores): v_posed (N x 20670) = rows (N x 220) . blend (220 x 20670), where
//     blend = [posedirs ; shapedirs^T ; v_template ; 0] is assembled once at load -- pose blend shapes,
//     shape blend shapes and the template in ONE GEMM that reads the 18 MB table once per call
//     (the per-vertex loop it replaces re-streamed posedirs for every frame).
//   kernel 3 (vertices): T_v = sum_j w_vj A_j over the vertex's non-zero skinning weights (a padded
//     (joint, weight) list built at load: 4 entries for a real SMPL model), verts = T_v . [v_posed; 1], in place.
//   kernel 4 (one block per frame): the 29 "spin2" joints (smpl.py:113-118) incl. the thorax row
//     of J_regressor_extra, weak-perspective -> perspective camera, projection / 112
//     (geometry.py:427-479, smpl.py:172-186).
__global__ __launch_bounds__(320) void smpl_chain_kernel(const float* __restrict__ betas, const float* __restrict__ rotmat, SmplTables t,
                                                           float* __restrict__ A_ws, float* __restrict__ kp3d, float* __restrict__ feat) {
    __shared__ float J[24][3];
    __shared__ float G[24][12];
    __shared__ float Rs[216];                                    // the frame's rotations, the parent table and the joints' depths in LDS
    __shared__ int par[24], depth[24];
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < 216; e += 320) Rs[e] = rotmat[(size_t)n * 216 + e];
    if (tid < 24) par[tid] = t.parents[tid];
    for (int e = tid; e < 72; e += 320) {
        float v = t.J_template[e];
#pragma unroll
        for (int l = 0; l < 10; ++l) v += t.J_shapedirs[e * 10 + l] * betas[(size_t)n * 10 + l];
        J[e / 3][e % 3] = v;
    }
    for (int e = tid; e < kBlendK; e += 320) {                   // this frame's row of the blend-shape GEMM
        float v = 0.f;
        if (e < 207) v = rotmat[(size_t)n * 216 + 9 + e] - ((e % 9 == 0 || e % 9 == 4 || e % 9 == 8) ? 1.f : 0.f);   // (R[1:] - I).flatten
        else if (e < 217) v = betas[(size_t)n * 10 + e - 207];
        else if (e == 217) v = 1.f;
        feat[(size_t)n * kBlendK + e] = v;
    }
    __syncthreads();
    if (tid < 24) {
        int d = 0;
        for (int p = par[tid]; p >= 0 && d < 24; p = par[p]) ++d;                 // parents[i] < i, parents[0] < 0
        depth[tid] = d;
    }
    __syncthreads();
    // The kinematic chain level by level: thread (joint i, entry r, c) of G_i = G_parent . [R_i | J_i - J_parent] once its parent's level
    // is done -- 8 levels for the SMPL tree instead of 24 joints one after another on one thread (13.6 us per call; same arithmetic per
    // entry, so the same results).
    const int ji = tid / 12, rc = tid - ji * 12, r = rc >> 2, c = rc & 3;
    const bool mine = tid < 288;
    const int myd = mine ? depth[ji] : -1, p = mine ? par[ji] : 0;
    for (int d = 0; d < 24; ++d) {
        if (mine && myd == d) {
            if (d == 0) {
                G[ji][rc] = c < 3 ? Rs[ji * 9 + r * 3 + c] : J[ji][r];
            } else {
                const float g0 = G[p][r * 4 + 0], g1 = G[p][r * 4 + 1], g2 = G[p][r * 4 + 2];
                if (c < 3) {
                    const float* Ri = Rs + ji * 9;
                    G[ji][rc] = g0 * Ri[c] + g1 * Ri[3 + c] + g2 * Ri[6 + c];
                } else {
                    const float t0 = J[ji][0] - J[p][0], t1 = J[ji][1] - J[p][1], t2 = J[ji][2] - J[p][2];
                    G[ji][rc] = g0 * t0 + g1 * t1 + g2 * t2 + G[p][r * 4 + 3];
                }
            }
        }
        __syncthreads();
        if (d >= 1 && __syncthreads_count(mine && myd > d) == 0) break;     // nobody is deeper
    }
    for (int e = tid; e < 24 * 12; e += 320) {
        const int i = e / 12, rc2 = e % 12, r2 = rc2 / 4, c2 = rc2 % 4;
        float v = G[i][rc2];
        if (c2 == 3) {
            kp3d[((size_t)n * 29 + i) * 3 + r2] = v;                 // posed joint = translation of G_i
            v -= G[i][r2 * 4 + 0] * J[i][0] + G[i][r2 * 4 + 1] * J[i][1] + G[i][r2 * 4 + 2] * J[i][2];
        }
        A_ws[(size_t)n * 288 + e] = v;
    }
}

constexpr int kNumVerts = 6890;
constexpr int kBlendCols = kNumVerts * 3;

// v_posed[f][c] = sum_k feat[f][k] * blend[k][c].  A workgroup owns 64 columns (one 16-column MFMA tile per wave) for ALL frames:
// each lane pulls its 55 table values (its column, the k rows of its quarter) into registers with 55 independent loads in flight
// -- the 18 MB table is read from memory exactly once per call, whatever the number of frames -- then walks the frames in passes of
// 64 (4 MFMA row tiles), the pass's rows (<= 64 x 220 floats) staged in LDS.  fp32 MFMA 16x16x4: an exact fp32 fma chain per element.
template <int MT>
__device__ __forceinline__ void blend_pass(const float* __restrict__ As, const float (&bv)[kBlendK / 4], float* __restrict__ Cs, int l15, int lq, int wave) {
    constexpr int LD = kBlendK + 1, KS = kBlendK / 4;
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* a0 = As + l15 * LD + lq;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m * 16 * LD + 4 * ks], bv[ks], acc[m], 0, 0, 0);
    // the tile leaves through LDS as whole 256-byte rows (64 columns of one frame are contiguous in the output)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) Cs[(m * 16 + lq * 4 + r) * 64 + wave * 16 + l15] = acc[m][r];
}

__global__ __launch_bounds__(256, 2) void smpl_blend_mfma_kernel(const float* __restrict__ feat, const float* __restrict__ blend, float* __restrict__ vposed,
                                                               int N) {
    constexpr int LD = kBlendK + 1, KS = kBlendK / 4;          // odd row stride: the 16 frames of an A fragment fall on distinct banks
    __shared__ float As[64 * LD];
    __shared__ __align__(16) float Cs[64 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int col = blockIdx.x * 64 + wave * 16 + l15;
    const bool cok = col < kBlendCols;
    const float* bp = blend + (cok ? col : 0) + (size_t)lq * kBlendCols;
    float bv[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bv[ks] = bp[(size_t)(4 * ks) * kBlendCols];       // 16 lanes read 64 contiguous bytes of a table row
    for (int f0 = 0; f0 < N; f0 += 64) {
        const int nf = min(64, N - f0), mtiles = (nf + 15) >> 4;
        __syncthreads();                                       // the previous pass is done with As
        {   // 64 rows x 55 16-byte units, 14 independent loads per thread in flight (a row is 880 bytes: units stay 16-byte aligned)
            constexpr int UPR = kBlendK / 4, NU = (64 * UPR + 255) / 256;
            f32x4 v[NU];
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int u = tid + 256 * i, f = u / UPR;
                v[i] = (u < mtiles * 16 * UPR && f < nf) ? *reinterpret_cast<const f32x4*>(feat + (size_t)(f0 + f) * kBlendK + 4 * (u - f * UPR))
                                                           : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const int u = tid + 256 * i, f = u / UPR, k = 4 * (u - f * UPR);
                if (u < mtiles * 16 * UPR) {
                    As[f * LD + k] = v[i][0]; As[f * LD + k + 1] = v[i][1]; As[f * LD + k + 2] = v[i][2]; As[f * LD + k + 3] = v[i][3];
                }
            }
        }
        __syncthreads();
        // straight-line MFMA chains (the row-tile count is a template argument: no branch between matrix instructions, so the
        // LDS reads of later k-steps are issued ahead of the chain instead of one exposed LDS latency per MFMA)
        switch (mtiles) {
            case 1: blend_pass<1>(As, bv, Cs, l15, lq, wave); break;
            case 2: blend_pass<2>(As, bv, Cs, l15, lq, wave); break;
            case 3: blend_pass<3>(As, bv, Cs, l15, lq, wave); break;
            default: blend_pass<4>(As, bv, Cs, l15, lq, wave); break;
        }
        __syncthreads();
        const int cbase = blockIdx.x * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) {                          // 8 bytes per lane (a frame's row of 20670 floats is 8-byte aligned, not 16)
            const int row = (tid >> 5) + 8 * i, c2 = (tid & 31) * 2;
            if (row < nf && cbase + c2 < kBlendCols)
                *reinterpret_cast<f32x2*>(vposed + (size_t)(f0 + row) * kBlendCols + cbase + c2) = *reinterpret_cast<const f32x2*>(Cs + row * 64 + c2);
        }
    }
}

// in place: verts[n][v] = T_v . [v_posed; 1] with T_v = sum over the vertex's non-zero weights of w * A_joint
__global__ __launch_bounds__(256) void smpl_skin_kernel(SmplTables t, const float* __restrict__ A_ws, float* __restrict__ verts) {
    __shared__ float sA[288];
    const int n = blockIdx.y, tid = threadIdx.x;
    for (int e = tid; e < 288; e += 256) sA[e] = A_ws[(size_t)n * 288 + e];
    __syncthreads();
    const int v = blockIdx.x * 256 + tid;
    if (v >= kNumVerts) return;
    float* pv = verts + ((size_t)n * kNumVerts + v) * 3;
    const float p0 = pv[0], p1 = pv[1], p2 = pv[2];
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    for (int k = 0; k < t.skin_k; ++k) {                       // ascending joint index, zero weights skipped: the dense sum's order
        const int j = t.skin_idx[(size_t)v * t.skin_k + k];
        const float wj = t.skin_w[(size_t)v * t.skin_k + k];
        if (j < 0) break;
#pragma unroll
        for (int e = 0; e < 12; ++e) T[e] += wj * sA[j * 12 + e];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) pv[r] = T[r * 4 + 0] * p0 + T[r * 4 + 1] * p1 + T[r * 4 + 2] * p2 + T[r * 4 + 3];
}

__global__ __launch_bounds__(256) void smpl_joints_kernel(const float* __restrict__ verts, const float* __restrict__ cam, SmplTables t,
                                                            float* __restrict__ kp3d, float* __restrict__ kp2d) {
    __shared__ float red[4][3];
    __shared__ float sj[29][3];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* vn = verts + (size_t)n * kNumVerts * 3;
    // 'Thorax (MPII)' = row 50-45 of J_regressor_extra (smpl.py:117): its non-zero entries as a (vertex, weight) list built at load
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int k = tid; k < t.thorax_n; k += 256) {
        const int v = t.thorax_idx[k];
        const float w = t.thorax_w[k];
        a0 += w * vn[v * 3]; a1 += w * vn[v * 3 + 1]; a2 += w * vn[v * 3 + 2];
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; red[wave][2] = a2; }
    if (tid < 72) sj[tid / 3][tid % 3] = kp3d[(size_t)n * 87 + tid];          // 24 posed joints (kernel 1)
    // joints45[35,37,40,42] = vertices lthumb 2746, lmiddle 2445, rthumb 6191, rmiddle 5905
    if (tid >= 96 && tid < 108) {
        const int e = tid - 96, jj = e / 3, d = e % 3;
        const int vid = jj == 0 ? 2746 : jj == 1 ? 2445 : jj == 2 ? 6191 : 5905;
        sj[24 + jj][d] = vn[vid * 3 + d];
    }
    __syncthreads();
    if (tid < 3) sj[28][tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    __syncthreads();
    if (tid < 87) kp3d[(size_t)n * 87 + tid] = sj[tid / 3][tid % 3];
    if (tid < 29 && cam != nullptr && kp2d != nullptr) {
        const float s = cam[(size_t)n * 3], tx = cam[(size_t)n * 3 + 1], ty = cam[(size_t)n * 3 + 2];
        const float tz = 2.f * 5000.f / (224.f * s + 1e-9f);
        const float X = sj[tid][0] + tx, Y = sj[tid][1] + ty, Z = sj[tid][2] + tz;
        kp2d[((size_t)n * 29 + tid) * 2 + 0] = 5000.f * (X / Z) / 112.f;
        kp2d[((size_t)n * 29 + tid) * 2 + 1] = 5000.f * (Y / Z) / 112.f;
    }
}

hipError_t launch_smpl(const float* betas, const float* rotmat, const float* cam, SmplTables t, float* A_ws, float* verts,
                       float* kp3d, float* kp2d, int N, hipStream_t s) {
    float* feat = A_ws + (size_t)N * 288;                    // the workspace holds (N,288) skinning matrices + (N,220) GEMM rows
    GRK_TRY(launch_k(smpl_chain_kernel, dim3(N), dim3(320), 0, s, betas, rotmat, t, A_ws, kp3d, feat));
    GRK_TRY(launch_k(smpl_blend_mfma_kernel, dim3((kBlendCols + 63) / 64), dim3(256), 0, s, (const float*)feat, t.blend, verts, N));
    GRK_TRY(launch_k(smpl_skin_kernel, dim3((kNumVerts + 255) / 256, N), dim3(256), 0, s, t, (const float*)A_ws, verts));
    GRK_TRY(launch_k(smpl_joints_kernel, dim3(N), dim3(256), 0, s, verts, cam, t, kp3d, kp2d));
    return hipGetLastError();
}

}  // namespace grk
