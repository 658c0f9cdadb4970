// Internal launch interface between the network plan (grnet.cpp) and the HIP kernels.
// All tensors are fp32 NCHW in HBM; a View addresses a channel slice [coff, coff+c) of a
// buffer that holds ctot channels per image, so producers write straight into their slice of
// the 480-channel concat buffer (reference: torch.cat, hrnet.py:524) with no copy kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <tuple>
#include <utility>
#include <vector>

// ---- environment switches ------------------------------------------------------------------------------------------------------------
// The product library reads FIVE environment variables, all documented in include/grnet_hip.h ("Environment"): GRNET_TRACE, GRNET_RCCL_LIB,
// GRNET_MULTI_LANE, GRNET_WINO, GRNET_BF16_CHAIN (the last three are the process-wide defaults of grnet_set_option values).  Every other
// A/B switch of the kernels and of the plan is a GRNET_AB(NAME, default): the default as a compile-time constant in the product build -- a
// stray GRNET_* variable cannot change kernels, numerics or the schedule there (tests/test_host_cpu.py greps for getenv, tests/
// test_gpu_options.py sets retired variables and compares bits) -- and read from the environment variable GRNET_<NAME> only in diagnostic
// builds (`make ABLATION=1 BUILD=build_abl LIB=../libgrnet_hip_abl.so`, loaded by tools/ through GRNET_LIB_PATH).
#ifdef GRNET_ABLATION
#include <cstdlib>
#define GRNET_AB(name, dflt) (getenv("GRNET_" #name) ? atoi(getenv("GRNET_" #name)) : (dflt))
#define GRNET_AB_F(name, dflt) (getenv("GRNET_" #name) ? atof(getenv("GRNET_" #name)) : (dflt))
#define GRNET_AB_SET(name) (getenv("GRNET_" #name) != nullptr)
#define GRNET_AB_STR(name) getenv("GRNET_" #name)
#else
#define GRNET_AB(name, dflt) (dflt)
#define GRNET_AB_F(name, dflt) (dflt)
#define GRNET_AB_SET(name) false
#define GRNET_AB_STR(name) (static_cast<const char*>(nullptr))
#endif

namespace grk {

// One-time work per device (kernel attributes, device queries) that launchers used to cache in plain function-local statics: two host
// threads driving two handles raced on those (a thread could see the flag set before the other's hipFuncSetAttribute had finished).
// Only SUCCESS is cached (round-4 advice: one transient failure used to disable a kernel family for the life of the process): a failed
// attempt is retried by the next launch.  Device ids index a small vector grown under the lock.
struct PerDeviceOnce {
    std::mutex m;
    std::vector<char> done;
    std::vector<int> val;
};
inline hipError_t current_device(int* dev) {
    return (hipGetDevice(dev) != hipSuccess || *dev < 0) ? hipErrorInvalidDevice : hipSuccess;
}
template <typename F>
hipError_t once_per_device(PerDeviceOnce& c, int dev, F&& f, int* value = nullptr) {      // f(int* value) -> hipError_t
    std::lock_guard<std::mutex> lock(c.m);
    if ((size_t)dev >= c.done.size()) { c.done.resize(dev + 1, 0); c.val.resize(dev + 1, 0); }
    if (!c.done[dev]) {
        int v = 0;
        const hipError_t e = f(&v);
        if (e != hipSuccess) return e;
        c.val[dev] = v;
        c.done[dev] = 1;
    }
    if (value) *value = c.val[dev];
    return hipSuccess;
}
inline hipError_t device_cu_count(int* cus) {                        // CUs of the current device
    static PerDeviceOnce c;
    int dev = 0;
    hipError_t e = current_device(&dev);
    if (e != hipSuccess) return e;
    return once_per_device(c, dev, [&](int* v) { return hipDeviceGetAttribute(v, hipDeviceAttributeMultiprocessorCount, dev); }, cus);
}

// Every kernel launch of the library goes through launch_k().  Normally it is a plain launch on the
// given stream; while a GraphRecorder is installed (grnet.cpp builds the per-forward hipGraph with the
// explicit node API, one parallel branch per lane) it appends a kernel node that depends on
// `rec->deps` instead, and leaves {that node} as the dependency of the next launch of the same op.
struct GraphRecorder {
    hipGraph_t graph = nullptr;
    std::vector<hipGraphNode_t> deps;
    int nodes = 0;
    // edge_order != 0: nodes are created WITHOUT dependencies and the edges added afterwards, all lane-chain edges (a launch -> the next launch of its lane)
    // before the cross-lane ones (1) or after them (2): ROCm's executor walks a node's children in insertion order and keeps the first one on the
    // parent's stream, so the order decides whether a lane of the plan stays one stream of the replay
    int edge_order = 0;
    int n_chain = 0;                                   // how many of the leading entries of `deps` are lane-chain predecessors (0 or 1)
    std::vector<hipGraphNode_t> chain_from, chain_to, cross_from, cross_to;
};
extern thread_local GraphRecorder* g_recorder;

template <typename... KArgs, typename... Args, size_t... I>
hipError_t launch_k_impl(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t s,
                         std::index_sequence<I...>, Args&&... args) {
    if (!g_recorder) {
        hipLaunchKernelGGL(kern, grid, block, shmem, s, std::forward<Args>(args)...);
        return hipGetLastError();
    }
    std::tuple<KArgs...> vals{static_cast<KArgs>(args)...};     // exact parameter types, copied by AddKernelNode
    void* ptrs[] = {static_cast<void*>(&std::get<I>(vals))...};
    hipKernelNodeParams p{};
    p.func = reinterpret_cast<void*>(kern);
    p.gridDim = grid;
    p.blockDim = block;
    p.sharedMemBytes = (unsigned)shmem;
    p.kernelParams = ptrs;
    p.extra = nullptr;
    hipGraphNode_t node = nullptr;
    GraphRecorder& r = *g_recorder;
    hipError_t e = r.edge_order ? hipGraphAddKernelNode(&node, r.graph, nullptr, 0, &p) : hipGraphAddKernelNode(&node, r.graph, r.deps.data(), r.deps.size(), &p);
    if (e != hipSuccess) return e;
    if (r.edge_order)
        for (size_t i = 0; i < r.deps.size(); ++i) {
            const bool chain = (int)i < r.n_chain;
            (chain ? r.chain_from : r.cross_from).push_back(r.deps[i]);
            (chain ? r.chain_to : r.cross_to).push_back(node);
        }
    r.deps.assign(1, node);                            // a second launch of the same op follows the first on its lane
    r.n_chain = 1;
    ++r.nodes;
    return hipSuccess;
}
template <typename... KArgs, typename... Args>
hipError_t launch_k(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t s, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count mismatch");
    return launch_k_impl(kern, grid, block, shmem, s, std::index_sequence_for<KArgs...>{}, std::forward<Args>(args)...);
}

struct View {
    float* p = nullptr;   // base of the whole buffer (image 0, channel 0)
    int ctot = 0;         // channels per image in the underlying buffer
    int coff = 0;         // first channel of this view
    int c = 0, h = 0, w = 0;
};

constexpr int kMaxAdd = 3;
constexpr int kConvCK = 8;      // input channels staged per K-chunk

// One fused convolution: out = act( conv(in, w) + bias + sum_k nearest_up(add_k, 2^shift_k) ).
// BatchNorm (eval) is folded into w / bias on the host in fp64 (reference: conv -> BN,
// hrnet.py:46-52; folding is exact up to fp32 rounding of the folded weights).
struct ConvArgs {
    const float* in; int in_ctot, in_coff;
    int N, Cin, H, W;
    float* out; int out_ctot, out_coff;
    int Cout, Ho, Wo;
    const float* w;            // packed [KS*KS][CinPad][CoutPad], BN scale folded in
    const float* bias;         // [CoutPad], BN shift (+ conv bias) folded in
    int CinPad, CoutPad;
    int ks, stride, relu;
    int relu_from;             // with relu: output channels >= relu_from get the ReLU (0: all) -- merged launches whose first segment is linear (direct kernels only)
    int n_add;
    const float* add[kMaxAdd]; int add_ctot[kMaxAdd], add_coff[kMaxAdd], add_shift[kMaxAdd];
    const float* zeros;        // >= 64 B of zeros in HBM: source for halo / padded-channel loads
    const float* w2; const float* bias2; float* out2; int out2_ctot, out2_coff, relu2;      // bf16 1x1 pair (layer1: Bottleneck k's 64 -> 256 expansion, then Bottleneck k+1's
                                                             // 256 -> 64 reduction from the tile the workgroup still holds in LDS): w2 packed [8][1][64][32], bias2 [64];
                                                             // out2: the reduction's NHWC bf16 view.  w2 == nullptr: none
    const float* in2; int in2_ctot, in2_coff, cin_split;     // bf16 1x1 only: input channels >= cin_split (a multiple of 32) come from a SECOND tensor of the same
                                                             // spatial size (a Bottleneck's last 1x1 and its 1x1 downsample as ONE GEMM over [t ; x]); in2 == nullptr: one input
    // filled by the launcher
    int R, G, Rin, Wp, PSTR, tiles_y, groups, TC, rows;
    float inv_RW, inv_Wo, inv_upc;   // 1 / (R*Wo), 1 / Wo, 1 / (PSTR/4) for the kernels' reciprocal-multiply divisions
    int nbuf;                  // bf16 kernel: patch buffers in LDS (2 = chunks double-buffered)
    int gx, gy, gx8, xcd;           // pixel tiles x output-channel blocks of the 1-D grid; xcd: XCD-aware block order (see xcd_block)
    int pw_stream;             // bf16 1x1 layers: 0 = never the stream kernel (conv_bf16_pw_stream), 1 = from the call size it pays at, 2 = whatever the call size
    int dbg;                   // timing-only ablation bits (tools/conv_micro.py); 0 in the product path
    int prio;                  // Winograd kernel: 0 = default wave priority, 1 / 2 = s_setprio 1 / 3 (critical-chain layers)
    int blk0, wsplit;          // Winograd kernel: first tile id of this launch; 1 = the 32-channel kernel on HALF a 64-channel block
                               // of weights packed for the 64-channel kernel (the half-size workgroups of a layer's last round)
};

// Returns hipSuccess or the launch error.  `tile_hint`: 0 = auto, 7 / 14 = force pixel sub-tiles.
hipError_t launch_conv(ConvArgs a, hipStream_t s, int tile_hint = 0);
int conv_pick_tc(int Cout);                 // cout tile (32 or 64) -> defines CoutPad at pack time
hipError_t conv_init();                     // sets max dynamic LDS on every instantiation
const char* conv_dominant_kernel_name();

// ---- Winograd F(4x4,3x3) on the fp32 matrix cores for the 3x3 stride-1 layers on 56x56 / 28x28 maps (conv_wino4.hip): 2.25 multiplies per output
bool conv_wino4_eligible(int cin, int cout, int ks, int stride, int h, int w, int n_add);
hipError_t launch_conv_wino4(ConvArgs a, hipStream_t s, int* n_launches = nullptr);     // a.w = transformed weights [36][CinPad][CoutPad]
void pack_wino4_weights(const double* w_folded /* (cout,cin,3,3) */, int cout, int cin, int cin_pad, int cout_pad, float* out, int map_width = 56);
int conv_wino4_blocks(int cout, int map_width);   // 16-channel blocks per workgroup of that layer (4 or 2)
int conv_wino4_wide(int cout, int map_width);     // > 0: the layer runs conv_wino4w_f32 (eight waves) with that many 16-channel blocks per wave (4: 128 output channels per
                                                  // workgroup, 2: 64); weights packed as for 4 blocks

void wino4_transform_filter(const double* g33, double* u36);   // U = G g G^T of F(4x4,3x3) in fp64
// ---- register-resident F(4x4,3x3) for the small maps (conv_wino4s.hip): 128 -> 128 @14x14, 256 -> 256 @7x7 (HR branches 2, 3), 256 -> 256 @14x14
// ---- layer1's 1x1 convolutions (64 -> 64 / 256, 256 -> 64 on 56x56 maps) with both operands straight from global memory (conv_pw.hip)
bool conv_pw_eligible(int cin, int cout, int ks, int stride, int h, int w, int n_add);
hipError_t launch_conv_pw(ConvArgs a, hipStream_t s);                     // a.w = the direct kernels' packing
// ---- the stem's first convolution (3 -> 64, 3x3, stride 2) with K = (channel, tap) flattened to 7 k-steps (conv_stem.hip)
bool conv_stem_eligible(int cin, int cout, int ks, int stride, int h, int w, int n_add);
hipError_t launch_conv_stem(ConvArgs a, hipStream_t s);                   // a.w = pack_stem_weights
void pack_stem_weights(const double* w_folded /* (64,3,3,3) */, float* out /* 7*4*64 */);
bool conv_wino4s_eligible(int cin, int cout, int ks, int stride, int h, int w, int n_add);
hipError_t launch_conv_wino4s(ConvArgs a, hipStream_t s, int ksplit);      // a.w = pack_wino4r_weights; ksplit 0: the shape's default
void pack_wino4r_weights(const double* w_folded /* (cout,cin,3,3) */, int cout, int cin, float* out /* 36*cin*cout */);
// out = relu?( sum_k nearest_up(add_k) ), 1..4 addends, out and every addend are Views.
struct SumArgs {
    float* out; int out_ctot, out_coff;
    int N, C, H, W, relu, n_add;
    const float* add[4]; int add_ctot[4], add_coff[4], add_shift[4];
};
hipError_t launch_fuse_sum(const SumArgs& a, hipStream_t s);

// The up half of an HR module's fuse layer in one launch (hr_fuse.hip; hrnet.py:249-267, fuse layers hrnet.py:189-244):
// out_i = act(x_i + bias_i + sum_{j > i} nearest_up(W_ij . x_j) + sum_k extra_k) for outputs i = 0 .. nb-2 of a module with nb
// branches (branch b: 32 << b channels on a (56 >> b)^2 map); extra_k: the finished stride-2 chains D_ij, j < i (same shape as out_i).
struct FuseUpSrc { const float* x; int ctot, coff; const float* w; };      // x_j as a View, W_ij packed by pack_fuse_up_weights
struct FuseUpOut {
    float* out; int out_ctot, out_coff;
    const float* base; int base_ctot, base_coff;     // x_i
    const float* bias;                               // sum over j of the folded BatchNorm shifts, [32 << i]
    int relu, n_extra;
    const float* extra[2]; int extra_ctot[2], extra_coff[2];
    FuseUpSrc src[3];                                // j = i+1 .. nb-1
};
struct FuseUpArgs { int N, nb; FuseUpOut o[3]; int only = -1; };      // only >= 0 (bf16 launch): this output alone, grid N * 7
hipError_t launch_hr_fuse_up(const FuseUpArgs& a, hipStream_t s);
void pack_fuse_up_weights(const double* w_folded /* (cout, cin) */, int cout, int cin, float* out /* cout * cin */);
// the same launch on NHWC bf16 tensors (pointers address bf16 data, bias fp32); weights [cin / 32][cout][32] bf16
hipError_t launch_hr_fuse_up_bf16(const FuseUpArgs& a, hipStream_t s);
void pack_fuse_up_weights_bf16(const double* w_folded /* (cout, cin) */, int cout, int cin, unsigned short* out /* cout * cin */);

// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (hrnet.py:443)
hipError_t launch_bilinear2x(const float* in, float* out, int N, int C, int H, int W, hipStream_t s);

// crop + normalise: uint8 HWC frames (n,H,W,3) [or one shared frame] + boxes (n,4) -> (n,3,224,224) f32 (img_utils.py:252-285,355-363)
hipError_t launch_crop_normalise(const unsigned char* img, int H, int W, int per_image, const float* bbox, float scale, int bgr,
                                 float* out, int N, hipStream_t s);

// the same with OpenCV's fixed-point warpAffine arithmetic; maps: (N,mstride) doubles -- mstride 6: the inverse affine map of each frame;
// mstride 10: + iw, ih, tx, ty of the two-warp crop of a non-square box (iw = 0: one warp)
hipError_t launch_crop_normalise_cv(const unsigned char* img, int H, int W, int per_image, const double* maps, int mstride, int bgr, float* out, int N,
                                    hipStream_t s);

// PARE head tail ---------------------------------------------------------------------------
// softmax over H*W of heat[:,1+j] and attention pooling of feat channels (keypoint_attention.py:42-48)
// heat: (N,25,P) with channel 0 = background; featA (N,CA,P) -> outA (N,CA,24); featB (N,CB,P) -> outB (N,CB,24)
hipError_t launch_softmax_pool(const float* heat, int heat_ctot, const float* featA, int CA, const float* featB, int CB,
                               float* outA, float* outB, float* prob_ws, int N, int P, hipStream_t s);

struct TailWeights {
    const float* pose_w;    // (128,24,6): head.pose_mlp.weight (1,6,128,24,1,1) re-ordered at load so a channel's 144 weights are contiguous
    const float* shape_w;   // (10,1536)
    const float* shape_b;
    const float* cam_w;     // (3,1536)
    const float* cam_b;
};
// plf (N,128,24), csf (N,64,24) -> rot6d (N,24,6), shape (N,10), cam (N,3), rotmat (N,24,9), theta (N,85)
// pool_ws: the workspace launch_softmax_pool filled (per-range partial sums); plf / csf are WRITTEN here
hipError_t launch_head_tail(const float* pool_ws, bool range_stats /* per-range (max, sum) pairs in front of the partials: the softmax is finished here (both paths) */, float* plf, float* csf, TailWeights w, float* rot6d, float* shape, float* cam,
                            float* rotmat, float* theta, int N, hipStream_t s);
// the same tail from given features (second head pass of the use_gait_feat branch, grnet.py:165; plf / csf are inputs)
hipError_t launch_head_tail_from_feats(const float* plf, const float* csf, TailWeights w, float* rot6d, float* shape, float* cam, float* rotmat,
                                       float* theta, int N, hipStream_t s);
hipError_t launch_rot6d_to_rotmat(const float* x, float* R, int m, hipStream_t s);      // geometry.py:395-410
hipError_t launch_rotmat_to_aa(const float* R, float* aa, int m, hipStream_t s);        // geometry.py:68-97,159-293
size_t softmax_pool_ws_floats(int N);
constexpr int kPoolStatsFloats = 7 * 24 * 2;    // per frame, in front of the partial sums: [range][joint][max, sum of exp]
constexpr int kPoolSplit = 7;                 // pixel ranges of the attention pooling (3136 = 7 x 448): ONE constant for the fp32 and bf16 paths

constexpr int kBlendK = 220;       // rows of the blend-shape table: 207 pose + 10 shape + 1 template + 2 of padding (k-steps of 4)
constexpr int kSmplWsFloatsPerFrame = 288 + kBlendK;   // skinning matrices + the frame's row of the blend-shape GEMM
struct SmplTables {
    const float* blend;        // (220,20670) = [posedirs ; shapedirs^T ; v_template ; 0 0], assembled at load
    const int* skin_idx;       // (6890,skin_k) joints with a non-zero skinning weight, ascending, padded with -1
    const float* skin_w;       // (6890,skin_k)
    int skin_k;                // max non-zero weights per vertex (4 for a real SMPL model)
    const float* J_template;   // (24,3)   = J_regressor . v_template           (precomputed at load)
    const float* J_shapedirs;  // (24*3,10) = J_regressor . shapedirs             (precomputed at load)
    const float* lbs_weights;  // (6890,24)
    const int* thorax_idx;     // non-zero entries of row 5 ('Thorax (MPII)') of J_regressor_extra (9,6890): vertex ids ...
    const float* thorax_w;     // ... and weights
    int thorax_n;
    const int* parents;        // (24)
};
// betas (N,10), rotmat (N,24,9), cam (N,3) -> verts (N,6890,3), kp3d (N,29,3), kp2d (N,29,2)
hipError_t launch_smpl(const float* betas, const float* rotmat, const float* cam, SmplTables t, float* A_ws,
                       float* verts, float* kp3d, float* kp2d, int N, hipStream_t s);

// GRU gait encoder (gait_feat_encoder.py:79-104) ------------------------------------------------
struct GruWeights {
    const float* cparam_w;                 // (128,3,24)
    const float* w_ih[2][2]; const float* w_hh[2][2]; const float* b_ih[2][2]; const float* b_hh[2][2];  // [layer][dir]
    const float* speed_w0; const float* speed_b0; const float* speed_w2; const float* speed_b2;
    const float* step_w0;  const float* step_b0;  const float* step_w2;  const float* step_b2;
    const float* phase_w0; const float* phase_b0; const float* phase_w2; const float* phase_b2;
};
struct GruWorkspace {
    float* xin;      // (b*T, 3072)   x + xc
    float* gi;       // (2, b*T, 900) input projections per direction
    float* l0;       // (b*T, 600)
    float* l1;       // (b*T, 600)
    float* hfin;     // (b, 1200)
    int mode = 3;                         // GRNET_OPT_GRU_MODE: recurrence form 0 .. 3 (+ 16: agent-scope granule stores whatever the placement)
    unsigned* fault = nullptr;            // host-visible word the split kernels set when a hand-off poll ran into its bound (the results of that call are NaN-poisoned)
    unsigned long long* xbuf = nullptr;   // b * kGruXbufU64PerSeq 8-byte words: (b, 2 dirs, 2 parities, 300) {h value, step tag} granules of the split recurrence, then
                                          // (b, 2 dirs, 8 slices) {XCC id, 1} placement words (may be null: the unsplit kernel runs)
};
constexpr int kGruXbufU64PerSeq = 2 * 2 * 300 + 2 * 8;
// bf16 path, round 6 (conv_bf16_roll.hip): the stem pair and a whole layer1 Bottleneck as ONE launch, a workgroup walking a frame row by row
hipError_t conv_bf16_roll_init();
int conv_bf16_roll_segments(int n_frames);          // row segments per frame of those launches (1, 2 or 4: frames x segments >= CUs)
hipError_t launch_conv_bf16_bneck(const void* in, int in_ctot, int in_coff, void* out, int out_ctot, int out_coff, int N, bool first, const void* w1, const float* b1,
                                  const void* w2, const float* b2, const void* w3, const float* b3, hipStream_t s);
hipError_t launch_conv_bf16_stem_pair(const float* frames, void* out, int out_ctot, int out_coff, int N, const void* w1pk, const float* b1, const void* w2, const float* b2,
                                      hipStream_t s);
// bf16 path (conv_bf16.hip): NHWC bf16 activations, fp32 accumulation on the bf16 matrix cores ----------------------
hipError_t conv_bf16_init();
hipError_t launch_conv_bf16(ConvArgs a, hipStream_t s, int tile_hint = 0);     // pointers in `a` address bf16 data (bias fp32)
// A chain of BasicBlocks of one HR branch in ONE launch, the frame resident in LDS (conv_bf16_chain.hip): convolutions 2k, 2k+1 are conv1 / conv2 of
// block k (hrnet.py:43-59), every one C -> C, 3x3, stride 1, ReLU; conv2 adds the block's input.  (C, W) in {(64,28), (128,14), (256,7)}: ONE launch;
// (32,56): one launch per block, 19-row bands of a frame resident (conv_bf16_block_band).
constexpr int kMaxChain = 8;
struct ChainArgs {
    const void* in; int in_ctot, in_coff;       // NHWC bf16 view (N, W, W, C)
    void* out; int out_ctot, out_coff;
    int N, nconv;
    const void* w[kMaxChain];                   // [C/32][9][C][32] bf16, BatchNorm folded
    const float* bias[kMaxChain];               // fp32 [C]
    void* mid[kMaxChain / 2 - 1];               // (32, 56) only -- one band-resident launch per BasicBlock: the output of block k < nconv/2 - 1
    int mid_ctot[kMaxChain / 2 - 1], mid_coff[kMaxChain / 2 - 1];
    int flags = 0;                                   // bit 0 (conv_bf16_block_frame): the block's output through the plane + 16-byte stores instead of straight from the accumulators (A/B)
};
int conv_bf16_chain_launches(int c, int w, int nconv);
hipError_t conv_bf16_chain_init();
bool conv_bf16_chain_eligible(int c, int w);
hipError_t launch_conv_bf16_chain(const ChainArgs& a, int c, int w, hipStream_t s);
// one wide 3x3 stride-1 convolution with a band of the input resident in LDS (conv_bf16_chain.hip: conv_bf16_wide_band); pointers as launch_conv_bf16
bool conv_bf16_wide_eligible(const ConvArgs& a);
hipError_t launch_conv_bf16_wide(const ConvArgs& a, hipStream_t s);
// one 3x3 stride-2 convolution with a band of the input resident in LDS, de-interleaved by row / column parity (conv_bf16_chain.hip: conv_bf16_s2_band)
bool conv_bf16_s2_eligible(const ConvArgs& a);
hipError_t launch_conv_bf16_s2(const ConvArgs& a, hipStream_t s);
hipError_t launch_nchw_f32_to_nhwc_bf16(const float* in, void* out, int N, int C, int H, int W, int Cp, hipStream_t s);
// the stem's 3 -> 64 stride-2 convolution straight from the caller's fp32 NCHW frames (N,3,224,224) to NHWC bf16 (N,112,112,out_ctot)
hipError_t launch_conv_bf16_stem(const float* frames, const void* wpk, const float* bias, void* out, int out_ctot, int out_coff, int N, int relu, hipStream_t s);
void pack_stem_weights_bf16(const double* w_folded /* (64,3,3,3) */, unsigned short* out /* 4*64*8 */);
hipError_t launch_nhwc_bf16_to_nchw_f32(const void* in, float* out, int N, int C, int H, int W, int ctot, int coff, hipStream_t s);
hipError_t launch_bilinear2x_bf16(const void* in, void* out, int N, int C, int H, int W, hipStream_t s);
hipError_t launch_fuse_sum_bf16(const SumArgs& a, hipStream_t s);
hipError_t launch_softmax_pool_bf16(const void* heat, int hc, const void* featA, int CA, int ctA, const void* featB, int CB, int ctB, float* pool_ws, int N,
                                    int P, hipStream_t s);

// C[M][N] = A[M][K] . B[N][K]^T + bias[N] on the fp32 matrix cores (row-major, K % 4 == 0, 16-byte aligned rows).
// split-K partial sums of few-row GEMMs live in scratch lent by the caller (kGemmWsFloats floats are always enough: a GEMM only splits
// while M*N < 128 tiles of 64x64, and into at most 16 slices); without it the launcher falls back to hipMallocAsync.
constexpr size_t kGemmWsFloats = (size_t)16 * 128 * 4096;
void set_gemm_workspace(float* ws, size_t floats);
hipError_t launch_gemm_nt_bias(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int ldc, hipStream_t s);

// TSAttnBlock (attention_utils.py:219-270) weights, reference layouts: Linear weights (out, in); jw1 (64,128,24), jw2 (128,64,24).
struct TsAttnWeights {
    const float *n1_g, *n1_b, *n2_g, *n2_b;
    const float *qkv_t_w, *qkv_t_b, *ts_w, *ts_b, *qkv_s_w, *qkv_s_b, *fc_s_w, *fc_s_b, *fc_t_w, *fc_t_b;
    const float *jw1, *jw2;
};
size_t tsattn_ws_floats(int b, int n);
// longest clip the temporal attention takes: its softmax row over the clip's frames lives in LDS ((512 + n) floats <= 160 KB)
constexpr int kTsAttnMaxFrames = 32768;
int tsattn_max_frames();        // min(kTsAttnMaxFrames, what the CURRENT device's LDS per workgroup holds): the up-front refusal matches the device
// x (b,n,3072) index c*24+j, xs (b,n,3200) index c*25+t -> y (b,n,3072); ws: tsattn_ws_floats(b, n) floats of scratch.
hipError_t launch_tsattn(const float* x, const float* xs, const TsAttnWeights& w, float* ws, float* y, int b, int n, hipStream_t s);

// FeatCorrector (feature_correction.py:104-157) pieces around the GRU and the attention block; BatchNorm1d folded to scale / shift at load.
struct FeatCorrWeights {
    const float *t0_w, *t0_b, *t3_w, *t3_b;      // gfeat_mpl_t: (1536,7), (1536), (3072,1536), (3072)
    const float *s0_w, *s0_b, *s3_w, *s3_b;      // gfeat_mpl_s: (64,7), (64), (128,64), (128)
    const float *bn_scale, *bn_shift;            // bn_in   (3072)
    const float *bns_scale, *bns_shift;          // bn_in_s (3200)
};
size_t featcorr_ws_floats(int b, int n);
hipError_t launch_gait_cparams(const float* cam, int cam_ld, const float* bbox, const float* cimg, float* cparams, int M, hipStream_t s);
hipError_t launch_featcorr(const float* x, const float* avg, const float* phase, const FeatCorrWeights& w, const TsAttnWeights& tw, float* ws, float* out,
                           int b, int n, hipStream_t s);

hipError_t launch_gru(const float* x, const float* cparams, GruWeights w, GruWorkspace ws, float* y, float* phase,
                      float* xc, int b, int T, hipStream_t s);

}  // namespace grk
