/*
 * libgrnet_hip.so -- C ABI of the MI355X-native MAX-GRNet per-frame inference path.
 *
 * The reference has no FFI for this path: it sits behind a plain torch.nn.Module.  Each entry
 * point below names the reference interface it replaces (file:line under the reference repo);
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions: every function returns 0 on success or a negative GRNET_E* code and never throws
 * across the ABI; grnet_last_error() returns a static-lifetime (per handle) message.  One handle
 * per GPU per process; a handle is not thread-safe (one caller thread + its stream); different
 * handles are independent.  All device work is enqueued on the stream passed in (a hipStream_t,
 * passed as void*); grnet_forward performs no allocation and no host synchronisation, so it can
 * be captured into a hipGraph by the caller (or by the library: GRNET_OPT_USE_GRAPH).
 * All tensors are fp32, dense, row-major ("C order"); device pointers are plain HIP device
 * pointers (e.g. torch.Tensor.data_ptr()), owned by the caller and never freed by the library.
 *
 * Environment.  The library reads exactly these variables (tests/test_host_cpu.py greps csrc/ for any other getenv):
 *   GRNET_TRACE=1        one line per plan / schedule / tuning decision on stderr; no effect on results
 *   GRNET_MULTI_LANE=0   process-wide default of GRNET_OPT_MULTI_LANE (profiling: the launches one after another on one stream)
 *   GRNET_WINO=0         process-wide default of GRNET_OPT_WINOGRAD
 *   GRNET_BF16_CHAIN=<mask>  process-wide default of GRNET_OPT_BF16_CHAIN
 *   GRNET_RCCL_LIB=<name>    the ONE library the exchange binds instead of librccl.so.1 (csrc/exchange.cpp)
 * The Python host adds GRNET_LIB_PATH (load another build of this library, tools/ only) and bench.py GRNET_BENCH_BACKEND (gloo rehearsals).
 * Every other GRNET_* name that earlier rounds' notes mention is an A/B switch of DIAGNOSTIC builds (make ABLATION=1; csrc/kernels.h
 * GRNET_AB): in this library it is a compile-time constant and setting the variable does nothing.
 */
#ifndef GRNET_HIP_H
#define GRNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct grnet grnet_t;

#define GRNET_OK 0
#define GRNET_EINVAL (-22)   /* bad argument / shape */
#define GRNET_ENOENT (-2)    /* missing weight tensor at finalize */
#define GRNET_ENOMEM (-12)
#define GRNET_EHIP (-5)      /* a HIP runtime call failed; see grnet_last_error */
#define GRNET_ESTATE (-1)    /* call order violated (e.g. forward before finalize) */

#define GRNET_DTYPE_F32 0
#define GRNET_DTYPE_I64 1    /* accepted for num_batches_tracked, ignored */

/* GRNet(...).to(device) -- lib/models/grnet.py:27-91, demo.py:106-111.  Allocates every
 * activation buffer for up to max_frames frames per call (the reference's batch axis N = B*T,
 * grnet.py:136-138).  dtype: 0 = fp32 (the reference's precision; the 1e-3 parity bar is stated for it);
 * 1 = bf16 storage (NHWC activations, folded weights) with fp32 accumulation on the bf16 matrix cores and an fp32 tail
 * (pooling sums, MLPs, SMPL) -- BASELINE configs 3 and 5; inputs and outputs of every entry point stay fp32. */
int grnet_create(grnet_t** out_handle, int device_id, int dtype, int max_frames);

/* model.load_state_dict(...) -- demo.py:116-122, batch_generation.py:214-218, and
 * GRNet.load_pare_dict / load_ckpt_w_prefix -- grnet.py:93-109, lib/utils/utils.py:185-196.
 * Called once per checkpoint tensor under its REFERENCE key name ("backbone.conv1.weight",
 * "head.pose_mlp.weight", ...; GRU keys "pfeat_corrector.featnet.*" or "gru.*").  host_ptr is
 * read during the call only. */
int grnet_load_tensor(grnet_t* h, const char* state_dict_key, const void* host_ptr, const int64_t* shape, int ndim,
                      int dtype);

/* SMPL(...) buffers -- lib/models/smpl.py:97-106,144 (smplx.SMPL tables + J_regressor_extra).
 * v_template (6890,3), shapedirs (6890,3,10), posedirs (207,20670), J_regressor (24,6890),
 * lbs_weights (6890,24), parents (24), J_regressor_extra (9,6890); host pointers. */
int grnet_load_smpl(grnet_t* h, const float* v_template, const float* shapedirs, const float* posedirs,
                    const float* J_regressor, const float* lbs_weights, const int32_t* parents,
                    const float* J_regressor_extra);

/* model.eval() + first use: folds every BatchNorm2d (eval, eps 1e-5) into its convolution in
 * fp64, repacks weights to the kernel layout and uploads them.  Fails with GRNET_ENOENT naming
 * the first missing tensor. */
int grnet_finalize_weights(grnet_t* h);

/* Caller-allocated device buffers for one call of n frames; any pointer may be NULL (that output
 * is then computed into an internal buffer and not returned).  Shapes follow VPRegressor.forward,
 * lib/models/pare.py:78-84, flattened over (B,T) -> n. */
typedef struct grnet_outputs {
    float* theta;             /* (n,85)  [cam(3), axis-angle pose(72), betas(10)]   pare.py:79 */
    float* verts;             /* (n,6890,3)                                         pare.py:80 */
    float* kp_2d;             /* (n,29,2)                                           pare.py:81 */
    float* kp_3d;             /* (n,29,3)                                           pare.py:82 */
    float* rotmat;            /* (n,24,3,3)                                         pare.py:83 */
    float* point_local_feat;  /* (n,128,24)  GRU input, grnet.py:148,163             */
    float* cam_shape_feats;   /* (n,64,24)                                          */
    float* pred_rot6d;        /* (n,24,6)   pare.py:299 */
    float* features;          /* (n,480,56,56) backbone output, hrnet.py:524 (debug / parity) */
    float* part_attn;         /* (n,25,56,56) heat-maps incl. background channel 0, pare.py:312 */
    float* smpl_feats;        /* (n,128,56,56) pare.py:323 */
} grnet_outputs_t;

/* model(batch)[-1] -- GRNet.forward, lib/models/grnet.py:129-175 (use_gait_feat=False path),
 * callers demo.py:164 and batch_generation.py:315.  frames_dev: (n,3,224,224) NCHW fp32 already
 * normalised; 1 <= n_frames <= max_frames. */
int grnet_forward(grnet_t* h, const float* frames_dev, int n_frames, const grnet_outputs_t* out, void* stream);

/* BidirectionalModel.forward -- lib/models/layers/gait_feat_encoder.py:79-104 (use_pareFeat=True,
 * eval).  x (b,T,3072) laid out c*24+j, cparams (b,T,3) -> y (b,3), phase (b,T,4), xc (b,T,3072);
 * device pointers; xc may be NULL. */
int grnet_gru_forward(grnet_t* h, const float* x_dev, const float* cparams_dev, int b, int T, float* y_dev,
                      float* phase_dev, float* xc_dev, void* stream);

/* TSAttnBlock.forward (use_jwff=True, eval) -- lib/models/layers/attention_utils.py:261-270: MultiAttention :164-217
 * (temporal attention over the n frames of a clip + spatial attention over the 25 tokens of a frame, softmax-gated),
 * JointWiseFeedForward :123-130 and the reference's own LayerNormalization :17-27, in the one-layer configuration of
 * feature_correction.py:92-101 (3072 -> 1000 -> 3072, 4 heads, 24 joints + 1 gait token).  Weights are loaded under
 * their reference keys with prefix "tsattn." or "pfeat_corrector.featTencoder.0.".  x (b,n,128,24), xs (b,n,128,25)
 * -> y (b,n,3072); device pointers. */
int grnet_tsattn_forward(grnet_t* h, const float* x_dev, const float* xs_dev, int b, int n, float* y_dev, void* stream);

#define GRNET_OPT_USE_GRAPH 1     /* 1: capture each distinct (n, pointers) forward into a hipGraph and replay it */
#define GRNET_OPT_CONV_TILE 2     /* 0 = cost model; 7 / 14 = whole-K tiles; 1071/1072/1041/1042/1171/1141 = split-K (psw,csw[,8 waves]) (tests / tuning); any forced
                                   * tile also switches the Winograd layers back to the direct kernels */
#define GRNET_OPT_MULTI_LANE 3    /* 1 (default): independent HR-module branches run on parallel streams / graph branches */
#define GRNET_OPT_WINOGRAD 7      /* 1 (default): the 3x3 stride-1 layers on 56x56 / 28x28 maps (layer1, HR branches 0 and 1, transition1, upsample heads, PARE
                                   * head; csrc/conv_wino4.hip) and on 14x14 / 7x7 maps (HR branches 2 and 3, the 256 -> 256 upsample-head layer;
                                   * csrc/conv_wino4s.hip) run as Winograd F(4x4,3x3) on the fp32 matrix cores (4x fewer multiplies, fp32 throughout, sums
                                   * re-associated: ~1e-5 of the output scale from the direct kernel per layer, <= 3.5e-5 on the path's outputs); 0: every
                                   * convolution is the direct implicit GEMM.  grnet_op_conv2d tile hints 2001 / 2020 (+ K split) run the two kernels on
                                   * one convolution.  (Options 4, 5, 6 -- grouped launches, the persistent dataflow launch and its fence -- were removed
                                   * in round 3 after losing every measurement; their sources are in the history: commit 8d3a931.) */
#define GRNET_OPT_BF16_CHAIN 8    /* bf16 handles, a mask of the band- / frame-resident kernel groups of csrc/conv_bf16_chain.hip and csrc/conv_bf16.hip (default: all bits = -1;
                                   * process-wide default from the environment variable GRNET_BF16_CHAIN; 0: one launch of the generic kernel per convolution at every
                                   * call size).  Bits 0-3: the four BasicBlocks (8 convolutions, lib/models/hrnet.py:141-187) of an HR branch as ONE launch with the
                                   * frame resident in LDS, in calls of >= 64 frames -- bit 0: 64 ch @28x28, bit 1: 128 ch @14x14, bit 2: 256 ch @7x7, bit 3: 32 ch
                                   * @56x56 (one launch per BasicBlock there, 8-row bands streamed through LDS).  Bit 4: the wide 3x3 stride-1 layers (upsample
                                   * heads, PARE head, layer1's 3x3) with a band of the input resident, >= 32 frames.  Bit 5: the 3x3 stride-2 layers (fuse layers'
                                   * down paths, transitions, the stem's second convolution) with the band de-interleaved by row / column parity, >= 64 frames.
                                   * Bit 6: layer1's 64 -> 256 expansions (hrnet.py:80-100) also run the NEXT Bottleneck's 256 -> 64 reduction from the tile they
                                   * hold in LDS, >= 19 frames.  Bit 7: the 1x1 layers of layer1 and of the PARE head on the persistent stream kernel
                                   * (conv_bf16_pw_stream, bit-identical to the generic kernel), >= 42 frames.  Bit 8: each layer1 Bottleneck (hrnet.py:62-100: 1x1
                                   * reduce, 3x3, 1x1 expand + residual) and bit 9: the stem pair (hrnet.py:470-476) as ONE launch whose workgroups walk a frame row
                                   * by row with every intermediate in LDS (csrc/conv_bf16_roll.hip), >= 64 frames; they take precedence over bits 4-7 on those layers. */
#define GRNET_OPT_GRU_MODE 9      /* form of the bi-GRU recurrence (csrc/gru_kernels.hip; gait_feat_encoder.py:79-104).  3 (default): W_hh resident in registers, split over 8
                                   * workgroups per (sequence, direction), rows per wave, v_exp / v_rcp gate functions, h_t handed over inside the XCD's L2
                                   * (workgroup-scope granule stores + L1-bypassing polls) where the 8 workgroups verifiably share an XCD; 2: the same with
                                   * expf / tanhf; 1: column slices, agent-scope hand-off; 0: one workgroup per (sequence, direction) (also taken for b > 16 or
                                   * T < 8).  + 16: agent-scope granule stores whatever the placement.  The L2 hand-off relies on write-through of workgroup-scope
                                   * stores into the XCD's shared L2: if a poll ever runs into its bound the kernel sets a host-visible word, the
                                   * NEXT grnet_gru_forward / grnet_gait_correct on the handle returns GRNET_ESTATE once (the earlier call's outputs are
                                   * NaN-poisoned) and the handle moves itself to + 16, then to 0. */
#define GRNET_OPT_BF16_MIN_FRAMES 10 /* bf16 handles: smallest call (frames) from which EVERY kernel group of GRNET_OPT_BF16_CHAIN runs; 0 (default): each group's own
                                   * measured threshold (64 / 32 / 64 / 19 / 42 frames).  1 lets tests and small-batch deployments take the LDS-resident kernels
                                   * at any call size. */
int grnet_set_option(grnet_t* h, int option, int value);

/* Optional, once per distinct n_frames after grnet_finalize_weights: times every launch configuration of every
 * distinct convolution shape (and grouped vs parallel-lane scheduling of the HR modules) on this GPU and keeps the
 * fastest.  Synchronises the stream; overwrites the activation buffers.  Without it the cost model decides. */
int grnet_tune(grnet_t* h, int n_frames, void* stream, int level /* 1: per shape in isolation, 2: + greedy in-context refinement (seconds) */);
/* The tuned table of n_frames as text (returns its length) / re-apply a stored table without measuring. */
int grnet_get_tuning(grnet_t* h, int n_frames, char* buf, int buf_size);
int grnet_set_tuning(grnet_t* h, int n_frames, const char* text);

/* Introspection used by bench.py / tests. */
int grnet_num_kernel_launches(grnet_t* h);      /* launches enqueued by one grnet_forward */
int grnet_num_conv_launches(grnet_t* h);        /* convolution launches of one grnet_forward (incl. the grouped fuse-term launch of each HR module) */
double grnet_conv_flops_per_frame(grnet_t* h);  /* 2 * MACs of all convolutions on the path */
/* The same with the layers that run a Winograd F(4x4,3x3) kernel counted at the 1/4 of their multiplies it executes (x 256/196 on 14x14 and
 * x 64/49 on 7x7 maps, whose tiles are padded), under the kernel choice of the handle's latest forward (that of a 16-frame call before the
 * first one: the 7x7 layers take the Winograd kernel from 12 frames per call on).  fp32 handles with GRNET_OPT_WINOGRAD on; equal to
 * grnet_conv_flops_per_frame otherwise.  Reporting only: the roofline figure is quoted on the algorithmic count above, this one says what
 * the matrix cores were actually asked to do. */
double grnet_conv_executed_flops_per_frame(grnet_t* h);
double grnet_conv_executed_flops_per_frame_n(grnet_t* h, int n_frames);   /* the same for a call of n_frames frames, stated explicitly */
/* The pos-th convolution launch of one forward in the un-grouped launch order (the dispatch order of a
 * GRNET_OPT_MULTI_LANE=0, un-tuned run -- what tools/layer_table.py joins per-dispatch profiler rows on):
 * info[0..11] = Cin, Cout, kernel, stride, Hin, Win, Hout, Wout, fused addends, relu, lane, addend elements per
 * frame; name = state_dict key of its weight (hrnet.py / pare.py module path).  Returns 0 or GRNET_EINVAL. */
int grnet_describe_conv(grnet_t* h, int pos, int32_t* info /* 12 */, char* name, int name_size);
/* fp32 handles run the 1x1 "up" terms of an HR module's fuse layer (hrnet.py:199-210 as summed at :258-265) as ONE launch per module
 * (csrc/hr_fuse.hip); grnet_describe_conv lists it in its place with Cin = 0, Cout = the channels it writes (32 [+ 64 [+ 128]]), kernel 1,
 * the 56x56 map of output 0, "fused addends" = the module's branch count, addend elements = the floats it reads per frame, and the name
 * "<module>.fuse_layers(up)".  Multiply-accumulates per frame of the pos-th launch of that list (either kind; < 0: bad position): */
double grnet_describe_conv_macs(grnet_t* h, int pos);
/* Per-launch figures for bench.py's kernel table (positions as in grnet_describe_conv): the kernel that runs the pos-th launch in a call
 * of n_frames frames (family<shape> name) and the multiply-accumulates per frame the matrix cores execute for it; and the average
 * duration, in microseconds, of `reps` back-to-back launches of it ALONE on `stream` (HIP events around the repetitions, two warm launches
 * first; the launch reads and writes its own planned buffers, whose contents are garbage afterwards like after any forward). */
int grnet_conv_kernel_info(grnet_t* h, int pos, int n_frames, char* name, int name_size, double* executed_macs_per_frame);
int grnet_time_conv(grnet_t* h, int pos, int n_frames, int reps, void* stream, float* us_out);
/* Diagnostic: ONE eager forward on the lane streams with a HIP timing event in front of and behind every op (placed after the op's
 * cross-lane waits), after two untimed warm passes; no profiler involved.  Writes one text line per op in enqueue order --
 * "index lane start_us end_us label", times relative to the first op's start -- into buf and returns the text's length (< 0: error;
 * GRNET_EINVAL if buf_size is too small).  Synchronises the stream.  tools/op_timeline.py prints concurrency and per-section sums. */
int grnet_op_timeline(grnet_t* h, const float* frames_dev, int n_frames, void* stream, char* buf, int buf_size);
/* Re-enqueue ONLY the convolution launches of the last forward, bracketed by HIP events on
 * `stream`; returns elapsed ms in *ms_out (synchronises the stream). */
int grnet_time_convs(grnet_t* h, int n_frames, void* stream, float* ms_out);

/* Single-op entry points (parity tests of each kernel against the oracle; not used by the path).
 * w_host: (Cout,Cin,ks,ks) already folded, bias_host (Cout) or NULL, add_dev: same shape as out or NULL. */
int grnet_op_conv2d(grnet_t* h, const float* in_dev, int n, int cin, int hgt, int wid, const float* w_host,
                    const float* bias_host, int cout, int ks, int stride, int relu, const float* add_dev,
                    float* out_dev, int tile_hint, void* stream);
/* (a bf16 handle runs the bf16 path's NHWC kernel between two layout conversions: c a multiple of 8) */
int grnet_op_bilinear2x(grnet_t* h, const float* in_dev, int n, int c, int hgt, int wid, float* out_dev, void* stream);
/* bf16 handles: a chain of nconv (even, <= 8) 3x3 stride-1 convolutions c -> c on (n,c,wid,wid) maps as ONE launch with the frame resident in LDS
 * (csrc/conv_bf16_chain.hip) -- convolutions 2k, 2k+1 are conv1 / conv2 of BasicBlock k (lib/models/hrnet.py:43-59: conv-BN-ReLU, conv-BN, + block
 * input, ReLU; four of them are one branch of a HighResolutionModule, hrnet.py:141-187).  (c, wid) in {(64,28), (128,14), (256,7)}; (32,56) runs
 * one launch per BasicBlock with 19-row bands of the frame resident.
 * in_dev / out_dev: f32 NCHW device buffers (rounded to / from NHWC bf16 around the launch); w_host: nconv x (c,c,3,3) folded weights,
 * bias_host: nconv x (c).  reps > 0 and us_out != NULL: `reps` more launches are timed with HIP events (us per launch).  Synchronises. */
int grnet_op_conv_chain(grnet_t* h, const float* in_dev, int n, int c, int wid, int nconv, const float* w_host, const float* bias_host,
                        float* out_dev, int reps, float* us_out, void* stream);

/* SMPL(...) forward with rotation matrices -- lib/models/smpl.py:108-130 (smplx LBS + the 29 "spin2" joints) and, when
 * cam_dev != NULL, the projection of smpl.py:172-186.  Used by the --smooth step (lib/utils/smooth_pose.py:59-100), which
 * re-evaluates SMPL on the filtered pose.  betas (n,10), rotmat (n,24,3,3), cam (n,3) or NULL -> verts (n,6890,3),
 * kp3d (n,29,3), kp2d (n,29,2) or NULL; device pointers; n <= max_frames. */
int grnet_smpl_forward(grnet_t* h, const float* betas_dev, const float* rotmat_dev, const float* cam_dev, int n, float* verts_dev,
                       float* kp3d_dev, float* kp2d_dev, void* stream);

/* PareHead.forward + VPRegressor.forward from GIVEN pooled features -- lib/models/pare.py:271-303 (_pare_get_final_preds :338-375:
 * per-joint 128->6, Linear 1536->10/3, rot6d_to_rotmat) and :52-91 (SMPL, projection, rotmat -> axis-angle, theta packing).  This is
 * the second head pass of the use_gait_feat branch (grnet.py:165,171: head(new_point_local_feat, cam_shape_feats, ...) then the
 * regressor) and the single-op hook the tail's parity tests use.  point_local_feat (n,128,24), cam_shape_feats (n,64,24) are device
 * INPUTS; `out` as in grnet_forward (theta, verts, kp_2d, kp_3d, rotmat, pred_rot6d; map outputs are ignored). */
int grnet_head_forward(grnet_t* h, const float* point_local_feat_dev, const float* cam_shape_feats_dev, int n, const grnet_outputs_t* out,
                       void* stream);
/* The use_gait_feat branch of GRNet.forward AFTER the first head pass -- lib/models/grnet.py:154-173: camera parameters in the
 * full image from the crop camera and the box (:156-160), FeatCorrector.forward (lib/models/layers/feature_correction.py:104-157:
 * GRU gait encoder, gait-token MLPs, BatchNorm1d, one TSAttnBlock, residual), the SECOND head pass on the corrected pose features
 * (:165) and the regressor (:171).  The reference class reads names that are defined nowhere (it cannot be constructed as shipped);
 * they are bound as DESIGN.md records and tests/golden/featcorr.npz pins the result to the reference's own code run with those
 * bindings.  Inputs are the FIRST pass's results for the whole clip(s) (the temporal modules need every frame, so on several GPUs
 * this runs after the all-gather): point_local_feat (b*T,128,24), cam_shape_feats (b*T,64,24), cam = pred_cam rows [s,tx,ty] with a
 * row stride of cam_ld floats (3, or 85 to pass theta), bbox (b*T,4) [cx,cy,w,h], cimg (b*T,2) = half the image size
 * (lib/dataset/inference.py:84-85).  `out` as in grnet_head_forward for all b*T frames; `gait` may be NULL. */
typedef struct grnet_gait_outputs {
    float* pred_avg;          /* (b,3)      gait parameters, gait_feat_encoder.py:100-101 */
    float* pred_phase;        /* (b,T,4)    gait_feat_encoder.py:102                      */
    float* pred_cparam;       /* (b*T,3)    grnet.py:160,173                              */
    float* point_local_feat;  /* (b*T,128,24) corrected pose features, feature_correction.py:150 */
} grnet_gait_outputs_t;
int grnet_gait_correct(grnet_t* h, const float* point_local_feat_dev, const float* cam_shape_feats_dev, const float* cam_dev, int cam_ld,
                       const float* bbox_dev, const float* cimg_dev, int b, int T, const grnet_outputs_t* out,
                       const grnet_gait_outputs_t* gait, void* stream);

/* rot6d_to_rotmat -- lib/utils/geometry.py:395-410: (m,6) -> (m,3,3); rotation_matrix_to_angle_axis -- geometry.py:68-97 (via
 * quaternion :213-293,:159-210, NaN -> 0): (m,3,3) -> (m,3).  The device functions the tail kernel calls, exposed so the
 * reference's edge-case vectors (degenerate 6-D pairs, the four quaternion branches, near-pi rotations) reach the GPU code. */
int grnet_op_rot6d_to_rotmat(grnet_t* h, const float* rot6d_dev, int m, float* rotmat_dev, void* stream);
int grnet_op_rotmat_to_aa(grnet_t* h, const float* rotmat_dev, int m, float* aa_dev, void* stream);

/* Inference.__getitem__ -- lib/dataset/inference.py:71-87 (get_single_image_crop_demo + ToTensor + Normalize,
 * lib/data_utils/img_utils.py:252-285,355-363; rot = 0): n uint8 HWC frames (n,H,W,3) [one_image_for_all: a single
 * (H,W,3) frame shared by all boxes] and boxes (n,4) [cx,cy,w,h] -> (n,3,224,224) fp32 normalised crops, all device
 * pointers.  `scale` multiplies w,h (the reference applies its bbox scale here a second time). */
int grnet_crop_normalise(grnet_t* h, const unsigned char* images_dev, int n, int height, int width, int one_image_for_all,
                         const float* bboxes_dev, float scale, int bgr, float* out_dev, void* stream);

/* The same step with OpenCV's OWN arithmetic -- cv2.warpAffine(img, trans, (224,224), INTER_LINEAR, BORDER_CONSTANT) as
 * generate_patch_image_cv calls it (lib/data_utils/img_utils.py:90-113): positions in 1/32-pixel fixed point from the inverse
 * affine map (AB_BITS 10, INTER_BITS 5, round half to even), 15-bit blending weights, taps outside the image 0, then ToTensor +
 * Normalize (:355-363).  inv_affine_dev: (n,6) float64, the inverse of `trans` per frame, which the HOST computes the way
 * gen_trans_from_patch_cv (:54-88: float32 triangle points), cv2.getAffineTransform (6x6 solve in double) and warpAffine's own
 * inversion do -- pipeline.cv_inverse_affine.  This is the default crop of demo.py / batch_generation.py; grnet_crop_normalise
 * (exact bilinear in float) stays for A/B.  cv2 is absent offline: agreement with a real OpenCV build is argued from its source
 * (DESIGN.md), the uint8 patch is bit-identical to the oracle's restatement of the same arithmetic. */
int grnet_crop_normalise_cv(grnet_t* h, const unsigned char* images_dev, int n, int height, int width, int one_image_for_all,
                            const double* inv_affine_dev, int bgr, float* out_dev, void* stream);
/* The same for ANY box, as generate_patch_image_cv crops it (lib/data_utils/img_utils.py:90-113): maps_dev is (n,10) float64 --
 * [0..5] the inverse affine map of the (first) warp, [6] iw, [7] ih, [8] tx, [9] ty.  iw = 0: a square box, one warp, as above.
 * iw > 0: bb_width != bb_height (:97-106) -- the first warp resizes the scaled box, aspect kept, to an iw x ih 8-bit image
 * (iw, ih = int(s*w), int(s*h), s = 224 / max(w, h)), a second warpAffine moves it by (224/2 - iw/2, 224/2 - ih/2) into the patch;
 * (tx, ty) is the inverse translation.  Same fixed-point arithmetic for both warps, the intermediate image rounded to 8 bits as
 * OpenCV returns it (it is computed on the fly, never stored).  The host forms the records: pipeline.cv_crop_maps.  This is the crop
 * GRNet.crop_normalise / demo.py / batch_generation.py use. */
int grnet_crop_normalise_cv_maps(grnet_t* h, const unsigned char* images_dev, int n, int height, int width, int one_image_for_all,
                                 const double* maps_dev, int bgr, float* out_dev, void* stream);

/* Copy a named intermediate of the LAST forward (first n_frames images) into out_dev as a dense
 * (n,C,H,W) tensor; shape_out[3] receives C,H,W (out_dev may be NULL to query the shape).  Names:
 * stem_conv1, stem_conv2, layer1, stage{2,3,4}.{branch}, up{2,3,4}.{layer}.{bilinear,conv}, and per HR module
 * stage{2,3,4}.{module}.x{branch} (the branch outputs = the fuse layer's inputs) / .y{branch} (the module's outputs).
 * Parity tests compare these with the oracle's taps of hrnet.py:469-536. */
int grnet_debug_tensor(grnet_t* h, const char* name, int n_frames, float* out_dev, int64_t* shape_out, void* stream);

/* ---- the exchange: all-gather of the per-frame records of a sharded clip (SURVEY 8b `grnet_allgather`, 8e) -------------------------------------
 * The reference has no counterpart (demo.py:126-188 and batch_generation.py:289-329 are one process on one device).  One process per GPU; every
 * rank runs grnet_forward on its contiguous frame range with the output pointers aimed INTO its send block, then ONE grnet_allgather (RCCL
 * ncclAllGather over xGMI, enqueued on `stream` like every other call) reassembles the sequence on every rank before anything temporal runs.
 * RCCL is looked up at run time (librccl.so.1, the copy the process already holds if any): the library loads and runs on one GPU without it, the
 * grnet_comm_* calls then fail with GRNET_ESTATE.  Bootstrap: rank 0 calls grnet_comm_unique_id and hands the 128 bytes to the other ranks over any host
 * channel (the launcher's store, a file, MPI); all ranks then call grnet_comm_create together (it blocks until every rank has arrived).
 * A host that already owns an ncclComm_t passes it to grnet_comm_adopt instead (not destroyed by grnet_comm_destroy).
 * Errors of this group: grnet_comm_last_error() (per thread), since no grnet_t is involved. */
typedef struct grnet_comm grnet_comm_t;
#define GRNET_COMM_ID_BYTES 128
/* Local and non-collective: 0 when RCCL could be bound in this process (librccl.so.1, or the one library named by the environment variable
 * GRNET_RCCL_LIB), GRNET_ESTATE otherwise.  Every rank calls it FIRST and the ranks agree on the minimum over a host channel before any of them
 * enters grnet_comm_unique_id / grnet_comm_create, so that no rank waits inside a collective the others never enter. */
int grnet_comm_probe(void);
int grnet_comm_unique_id(void* id_out, int id_size /* >= GRNET_COMM_ID_BYTES */);
int grnet_comm_create(grnet_comm_t** out_comm, const void* id, int world, int rank, int device_id);
int grnet_comm_adopt(grnet_comm_t** out_comm, void* nccl_comm /* ncclComm_t */, int world, int rank);
/* recv_dev holds world * bytes_per_rank bytes; rank r's block lands at offset r * bytes_per_rank (send_dev may alias its own slot). */
int grnet_allgather(grnet_comm_t* comm, const void* send_dev, void* recv_dev, size_t bytes_per_rank, void* stream);
int grnet_comm_info(grnet_comm_t* comm, int* world, int* rank);
void grnet_comm_destroy(grnet_comm_t* comm);
const char* grnet_comm_last_error(void);

const char* grnet_last_error(grnet_t* h);
const char* grnet_version(void);
void grnet_destroy(grnet_t* h);

#ifdef __cplusplus
}
#endif
#endif /* GRNET_HIP_H */
