// Runtime of the MAX-GRNet per-frame path on one MI355X: the network plan (a flat list of fused
// kernel launches over statically planned HBM buffers), the weight loader (reference state_dict
// keys -> BN-folded, kernel-layout weights) and the C ABI of include/grnet_hip.h.
//
// Topology restated from the reference constructors / forward passes (not translated from them):
//   backbone  lib/models/hrnet.py:469-536 with DOWNSAMPLE=False, USE_CONV=True (grnet.py:52-57)
//   head      lib/models/pare.py:245-303
//   regressor lib/models/pare.py:52-91, lib/models/smpl.py:149-191
// Data layout: fp32 NCHW, one buffer per intermediate tensor sized for max_frames images (the whole
// activation set is ~103 MB / frame, so 1 250 frames still fit the 288 GB of HBM3E); image stride is
// independent of the number of frames in a call, so a plan built once serves any n <= max_frames.
#include "../../include/grnet_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "kernels.h"

using namespace grk;

namespace grk { thread_local GraphRecorder* g_recorder = nullptr; }

namespace {

constexpr double kBnEps = 1e-5;   // nn.BatchNorm2d default eps

inline uint16_t f32_to_bf16(float f) {                     // round to nearest even, as the kernels do
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
const int kBranchCh[4] = {32, 64, 128, 256};

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    size_t numel() const { return data.size(); }
};

struct AddRef { View v; int shift; };

struct ConvSeg { std::string wkey, bnprefix, biaskey; int cout; };

struct ConvLayer {
    View in, out;
    std::vector<ConvSeg> segs;
    int cout = 0, ks = 1, stride = 1, relu = 0;
    int relu_from = 0;          // relu applies to output channels >= relu_from (a merged launch whose first segment is linear)
    bool solo = false;          // runs with no other launch beside it (stem, layer1, PARE head): isolated timings predict it well
    int cin_w = 0;              // input channels of the weight tensor (< in.c only for the bf16 stem: 3 of the 8 stored)
    std::vector<AddRef> adds;
    float* w_dev = nullptr;
    float* b_dev = nullptr;
    float* wino4_dev = nullptr; // transformed weights [36][cin_pad][cout_pad] of the Winograd F(4x4,3x3) kernel (the widest 56x56 layers only)
    float* stem_dev = nullptr;   // flattened-K weights of the stem's first convolution (conv_stem.hip)
    float* wino4s_dev = nullptr; // transformed weights of the register-resident F(4x4,3x3) kernel of the 14x14 / 7x7 maps (conv_wino4s.hip)
    int cin_pad = 0, cout_pad = 0;
    double macs_per_frame = 0;
    int lane_hint = 0;          // lane of this convolution when it is launched on its own (not as a group member)
    View in2;                   // bf16: second input of a merged 1x1 launch (layer1.0: conv3 over t and the downsample over x as ONE GEMM); in2.c == 0: none
    ConvSeg seg2;               // its weights / BatchNorm (same output channels, summed)
    int pair_next = -1, pair_of = -1;   // bf16 layer1: this 64 -> 256 expansion also runs convolution pair_next (the next Bottleneck's 256 -> 64 reduction) from its tile / this
                                        // reduction runs inside the launch of convolution pair_of (large calls: pair_active())
    int chain = -1, chain_pos = 0;   // bf16: member chain_pos of BasicBlock chain `chain` (conv_bf16_chain.hip); position 0 launches the whole chain in large calls
    int roll = -1, roll_pos = 0;     // bf16: member roll_pos of the row-walking launch `roll` (conv_bf16_roll.hip: the stem pair, a layer1 Bottleneck); position 0 launches it in large calls
    std::map<int, int> tuned;   // n_frames -> launch configuration (tile hint) measured fastest by grnet_tune
};

// The up half of one HR module's fuse layer (hr_fuse.hip): outputs 0 .. nb-2 in one launch.
struct FuseUpPlan {
    int nb = 0;
    std::string prefix;                  // "backbone.stage3.1."
    std::vector<View> xs;                // the module's branch outputs (nb)
    std::vector<View> outs;              // outputs 0 .. nb-2 (final)
    std::vector<std::vector<View>> extra; // per output: the finished down chains D_ij, j < i, added by the grouped launch
    float* w_dev[3][3] = {};             // [output i][source j - i - 1], pack_fuse_up_weights
    float* b_dev[3] = {};                // [output i]: sum over j of the folded BatchNorm shifts
    double macs_per_frame = 0;
    int only = -1;                       // >= 0: this plan finishes output `only` alone (bf16: one launch per output, each on its branch's lane)
};

// The 8 convolutions (4 BasicBlocks) of one branch of one HR module, runnable as ONE launch on the bf16 path (conv_bf16_chain.hip).
struct ChainPlan {
    std::vector<int> convs;              // indices into grnet::convs, in execution order
    int c = 0, w = 0;
};

// Layers that run as ONE row-walking launch on the bf16 path in large calls (conv_bf16_roll.hip)
struct RollPlan {
    int kind = 0;                        // 0: stem pair (conv1, conv2); 1: layer1.0 (conv1, conv2, conv3 over [u ; x]); 2: layer1.1-3 (conv1, conv2, conv3 + residual)
    std::vector<int> convs;              // indices into grnet::convs, in execution order
};

struct Op {
    enum Kind { CONV, SUM, BILINEAR, POOL, TAIL, SMPL, CONVERT, FUSEUP } kind;
    int conv_idx = -1;
    SumArgs sum{};
    View bin, bout;   // bilinear
    // multi-lane execution: independent branches of the HR modules run on parallel HIP streams
    // (captured as parallel branches of the hipGraph); cross-lane read-after-write edges are events
    int lane = 0;
    int follow = -1;          // plan index of an op this one depends on and whose stream it must share (the lane scheduler keeps them together)
    std::vector<int> waits;   // ops (on other lanes) whose completion event this op waits for
    bool record = false;      // some op on another lane consumes this op's output
};

constexpr int kLanes = 8;            // streams available to the lane scheduler (the hand-written plan uses 4)

// Every entry point runs on the handle's device whatever the caller's current device is, and leaves the caller's device as it found it.
// The few-row GEMMs borrow split-K scratch through a thread-local pointer (set_gemm_workspace); this lease takes it back on every
// exit path, so a failed call never leaves the pointer aimed at scratch the handle may free later.
struct GemmWorkspaceLease {
    GemmWorkspaceLease(float* ws, size_t floats) { set_gemm_workspace(ws, floats); }
    ~GemmWorkspaceLease() { set_gemm_workspace(nullptr, 0); }
    GemmWorkspaceLease(const GemmWorkspaceLease&) = delete;
    GemmWorkspaceLease& operator=(const GemmWorkspaceLease&) = delete;
};

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) return fail(GRNET_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

}  // namespace

struct grnet {
    int device = 0, max_frames = 0;
    int dtype = 0;               // 0: fp32 NCHW activations, 1: bf16 NHWC activations (conv_bf16.hip), fp32 tail either way
    View v_in8;                  // bf16 without conv_bf16_stem: the caller's frames converted to NHWC bf16 with 8 channels (3 real)
    bool bf16_stem = false;
    bool finalized = false, smpl_loaded = false, gru_ready = false, tsattn_ready = false, featcorr_ready = false;
    TsAttnWeights tsw{};
    FeatCorrWeights fcw{};
    bool use_graph = false;
    int conv_tile_hint = 0;
    std::string err;

    std::unordered_map<std::string, HostTensor> tensors;
    std::vector<ConvLayer> convs;
    std::vector<Op> ops;        // the plan in the order it is written (build_plan)
    std::vector<Op> ops_flat;   // the same ops placed on the lane streams by schedule_lanes(): the enqueue order
    std::vector<hipEvent_t> op_events_flat;
    int wino_mode = 1;                               // GRNET_OPT_WINOGRAD: 1 = the eligible 3x3 layers on 56x56 maps run the Winograd kernel

    // planned buffers: (pointer slot, floats per image)
    std::vector<std::pair<float**, size_t>> pending;   // pointers patched after the arena exists
    std::vector<std::unique_ptr<float*>> slots;
    float* arena = nullptr;
    size_t arena_floats = 0;
    std::vector<void*> dev_allocs;
    float* zeros = nullptr;

    // named views for outputs / debug
    View v_input, v_cat, v_heat, v_smpl_feats, v_csmap;
    float *d_plf = nullptr, *d_csf = nullptr, *d_stats = nullptr, *d_rot6d = nullptr, *d_shape = nullptr, *d_cam = nullptr;
    float *d_rotmat = nullptr, *d_theta = nullptr, *d_A = nullptr, *d_verts = nullptr, *d_kp3d = nullptr, *d_kp2d = nullptr;

    TailWeights tailw{};
    SmplTables smpl{};
    GruWeights gruw{};
    std::vector<float> J_regressor_host;

    struct GraphKey {
        int n; const void* in; grnet_outputs_t o;
        bool operator<(const GraphKey& r) const {
            if (n != r.n) return n < r.n;
            if (in != r.in) return in < r.in;
            return std::memcmp(&o, &r.o, sizeof(o)) < 0;
        }
    };
    struct GraphEntry { hipGraphExec_t exec; unsigned long long last_use; };
    std::map<GraphKey, GraphEntry> graphs;                  // at most kMaxGraphs captured forwards, least recently used evicted
    unsigned long long graph_clock = 0;
    std::vector<GraphKey> seen_once;
    hipStream_t capture_stream = nullptr;   // the caller's stream may be the (uncapturable) null stream
    hipStream_t side[kLanes] = {};      // lanes 1.. (lane 0 = the caller's stream)
    hipEvent_t ev_fork = nullptr, ev_join[kLanes] = {};
    int lanes_used = 1;
    int cur_lane = 0;
    bool multi_lane = true;
    int launches_last = 0;
    // diagnostic (grnet_op_timeline): timing events around every op of one eager forward
    std::vector<hipEvent_t>* tl_start = nullptr;
    std::vector<hipEvent_t>* tl_end = nullptr;

    int fail(int code, const std::string& msg) {
        err = msg;
        return code;
    }

    // GRNET_OPT_GRU_MODE (gru_kernels.hip): 3 = rows-per-wave split recurrence with v_exp / v_rcp gates (default), 2 = the same with expf / tanhf,
    // 1 = round 3's column slices, 0 = one workgroup per (sequence, direction); + 16: agent-scope granule stores whatever the placement.
    // gru_fault: one host-mapped word the split kernels set when a hand-off poll hits its bound (an XCD-placement or memory-scope assumption broke:
    // round-5 advice).  The next temporal call sees it WITHOUT a synchronisation, reports GRNET_ESTATE once and moves the handle to agent-scope stores.
    int gru_mode = 3;
    unsigned* gru_fault = nullptr;        // host pointer (hipHostMalloc, mapped)
    unsigned* gru_fault_dev = nullptr;    // the same word as the device sees it
    int gru_fault_check() {
        if (!gru_fault) {
            void* q = nullptr;
            if (hipHostMalloc(&q, 64, hipHostMallocMapped) != hipSuccess) return fail(GRNET_ENOMEM, "hipHostMalloc of the GRU fault word failed");
            gru_fault = static_cast<unsigned*>(q);
            *gru_fault = 0u;
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, q, 0) != hipSuccess) return fail(GRNET_EHIP, "hipHostGetDevicePointer failed");
            gru_fault_dev = static_cast<unsigned*>(d);
        }
        if (*reinterpret_cast<volatile unsigned*>(gru_fault)) {
            *gru_fault = 0u;
            const bool was_agent = (gru_mode & 16) != 0;
            gru_mode = was_agent ? 0 : (gru_mode | 16);
            return fail(GRNET_ESTATE, std::string("a hand-off poll of the split GRU recurrence timed out in an earlier call on this handle: the outputs of that call are NaN-poisoned. ") +
                        (was_agent ? "The handle now runs the unsplit recurrence (GRNET_OPT_GRU_MODE 0)." : "The handle now publishes with agent-scope stores (GRNET_OPT_GRU_MODE + 16).") + " Repeat the call.");
        }
        return 0;
    }

    // Scratch of the temporal modules (GRU, attention block, feature corrector): owned by the handle and grown on demand, so a call
    // with a size seen before allocates nothing (graph-capturable, no allocator traffic per call).  Growing synchronises the device.
    float* temporal_ws = nullptr;
    size_t temporal_ws_floats = 0;
    int temporal_scratch(size_t floats, float** out) {
        if (floats > temporal_ws_floats) {
            if (temporal_ws) { (void)hipDeviceSynchronize(); (void)hipFree(temporal_ws); temporal_ws = nullptr; temporal_ws_floats = 0; }
            const size_t want = floats + floats / 4;                 // head-room: clips of slightly different length reuse the buffer
            void* q = nullptr;
            if (hipMalloc(&q, want * sizeof(float)) != hipSuccess) return fail(GRNET_ENOMEM, "temporal workspace (" + std::to_string(want * 4 >> 20) + " MiB)");
            temporal_ws = static_cast<float*>(q);
            temporal_ws_floats = want;
        }
        *out = temporal_ws;
        return 0;
    }

    // releases everything the handle owns (also the clean-up of a failed grnet_create)
    ~grnet() {
        for (auto& g : graphs) (void)hipGraphExecDestroy(g.second.exec);
        if (capture_stream) (void)hipStreamDestroy(capture_stream);
        for (int l = 1; l < kLanes; ++l) {
            if (side[l]) (void)hipStreamDestroy(side[l]);
            if (ev_join[l]) (void)hipEventDestroy(ev_join[l]);
        }
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        for (hipEvent_t e : op_events_flat) if (e) (void)hipEventDestroy(e);
        for (void* p : dev_allocs) (void)hipFree(p);
        if (temporal_ws) (void)hipFree(temporal_ws);
        if (gru_fault) (void)hipHostFree(gru_fault);
        if (arena) (void)hipFree(arena);
    }
    void drop_graphs() {
        for (auto& g : graphs) (void)hipGraphExecDestroy(g.second.exec);
        graphs.clear();
    }

    // ------------------------------------------------------------------ plan construction
    View new_buffer(int c, int h, int w) {
        slots.emplace_back(new float*(nullptr));
        const int ct = dtype == 1 ? (c + 7) / 8 * 8 : c;     // NHWC bf16: 16-byte channel groups (the 25 heat channels -> 32)
        pending.emplace_back(slots.back().get(), (size_t)ct * h * w);
        View v;
        v.p = nullptr;
        v.ctot = ct; v.coff = 0; v.c = c; v.h = h; v.w = w;
        // p is resolved through slot index stored in coff-independent table: keep index in a side map
        view_slot[(int)views_created] = slots.size() - 1;
        v.p = reinterpret_cast<float*>(views_created++ + 1);   // temporary tag, replaced in resolve()
        return v;
    }
    std::unordered_map<int, size_t> view_slot;
    size_t views_created = 0;

    static View slice(View v, int coff, int c) {
        v.coff += coff;
        v.c = c;
        return v;
    }

    View add_conv(View in, std::vector<ConvSeg> segs, int ks, int stride, bool relu, std::vector<AddRef> adds = {},
                  const View* out_override = nullptr) {
        ConvLayer L;
        L.in = in;
        int cout = 0;
        for (auto& s : segs) cout += s.cout;
        const int pad = ks / 2;
        const int ho = (in.h + 2 * pad - ks) / stride + 1, wo = (in.w + 2 * pad - ks) / stride + 1;
        L.out = out_override ? *out_override : new_buffer(cout, ho, wo);
        L.segs = std::move(segs);
        L.cout = cout; L.ks = ks; L.stride = stride; L.relu = relu;
        L.adds = std::move(adds);
        L.cin_w = (dtype == 1 && in.c == 8 && in.ctot == 8) ? 3 : in.c;     // bf16 stem: 3 real channels stored as 8
        L.macs_per_frame = (double)ho * wo * cout * L.cin_w * ks * ks;
        L.lane_hint = cur_lane;
        L.solo = solo_region;
        convs.push_back(L);
        Op op;
        op.kind = Op::CONV;
        op.conv_idx = (int)convs.size() - 1;
        op.lane = cur_lane;
        ops.push_back(op);
        return convs.back().out;
    }
    bool solo_region = false;   // build_plan: convolutions added now are part of a chain nothing else overlaps
    View conv_bn(View in, const std::string& wkey, const std::string& bn, int cout, int ks, int stride, bool relu,
                 std::vector<AddRef> adds = {}, const View* out_override = nullptr) {
        return add_conv(in, {ConvSeg{wkey, bn, "", cout}}, ks, stride, relu, std::move(adds), out_override);
    }

    View add_bilinear(View in) {
        View out = new_buffer(in.c, in.h * 2, in.w * 2);
        Op op;
        op.kind = Op::BILINEAR;
        op.bin = in; op.bout = out;
        op.lane = cur_lane;
        ops.push_back(op);
        return out;
    }

    // HighResolutionModule (hrnet.py:249-267).  out0 (optional) receives fused output 0.
    // Every branch convolution is its own launch on the lane of its branch; schedule_lanes() places the fuse layer's launches.
    std::vector<View> hr_module(std::vector<View> xs, const std::string& p, const View* out0) {
        const int nb = (int)xs.size();
        std::vector<int> branch_tail(nb, -1);                               // plan index of the launch that writes x_b
        cur_lane = 0;
        std::vector<std::vector<int>> branch_ops(nb);                       // plan indices of the branch's convolutions, in order
        for (int k = 0; k < 4; ++k) {
            std::vector<View> y(nb);
            for (int b = 0; b < nb; ++b) {
                cur_lane = b;
                const std::string q = p + "branches." + std::to_string(b) + "." + std::to_string(k) + ".";
                y[b] = conv_bn(xs[b], q + "conv1.weight", q + "bn1", kBranchCh[b], 3, 1, true);
                branch_ops[b].push_back((int)ops.size() - 1);
            }
            for (int b = 0; b < nb; ++b) {
                cur_lane = b;
                const std::string q = p + "branches." + std::to_string(b) + "." + std::to_string(k) + ".";
                xs[b] = conv_bn(y[b], q + "conv2.weight", q + "bn2", kBranchCh[b], 3, 1, true, {AddRef{xs[b], 0}});
                branch_tail[b] = (int)ops.size() - 1;
                branch_ops[b].push_back((int)ops.size() - 1);
            }
        }
        // bf16: the branch's four BasicBlocks are also ONE chain launch (conv_bf16_chain.hip; taken in large calls, chain_active()).  The members
        // keep their own ops -- small calls launch them one by one -- and are pinned to one stream in order, so the events recorded behind the
        // (then empty) member ops still order every consumer behind the chain launch, which sits at the first member's place.
        if (dtype == 1)
            for (int b = 0; b < nb; ++b) {
                if (!conv_bf16_chain_eligible(kBranchCh[b], xs[b].w) || (int)branch_ops[b].size() > kMaxChain) continue;
                ChainPlan cp;
                cp.c = kBranchCh[b]; cp.w = xs[b].w;
                for (size_t i = 0; i < branch_ops[b].size(); ++i) {
                    Op& op = ops[branch_ops[b][i]];
                    convs[op.conv_idx].chain = (int)chains.size();
                    convs[op.conv_idx].chain_pos = (int)i;
                    cp.convs.push_back(op.conv_idx);
                    if (i) op.follow = branch_ops[b][i - 1];
                }
                chains.push_back(cp);
            }
        const std::string tag = p.substr(p.find("stage"));                  // "stage3.1."
        for (int b = 0; b < nb; ++b) name_view(tag + "x" + std::to_string(b), xs[b]);
        // GRNET_FUSE_UP=0: the round-3 fuse layer (one 1x1 launch per up term, an elementwise launch for output 0)
        // (GRNET_BF16_FUSE_UP=0 does the same for the bf16 path, which has the grouped launch since round 5)
        static const int fuse_up_env = GRNET_AB(FUSE_UP, 1);
        const int fuse_up_bf_env = GRNET_AB(BF16_FUSE_UP, 0);    // read per handle: the tests build all three.  0 is the default: at 256 frames the lane-overlapped step is 10.68 / 10.91 / 10.67 ms for 0 / 1 / 2 (one lane: 11.69 / 11.39) -- the small launches hide behind the other lanes, the stored D_ij of layout 1 do not
        std::vector<View> outs = (dtype == 0 ? fuse_up_env : fuse_up_bf_env == 1) ? hr_fuse_grouped(xs, p, out0, branch_tail)
                                                                                  : hr_fuse_separate(xs, p, out0, dtype == 1 && fuse_up_bf_env == 2, branch_tail);
        for (int i = 0; i < nb; ++i) name_view(tag + "y" + std::to_string(i), outs[i]);
        cur_lane = 0;
        return outs;
    }

    // Fuse layer, round 4 (hrnet.py:189-244 as used by :258-265).  Output i = relu(sum_j term_ij) with term_ij = x_i (j == i),
    // nearest_up(BN(conv1x1(x_j))) (j > i), a chain of i-j stride-2 3x3 convolutions (j < i).  The branches of a module end at
    // different times -- the 56x56 branch ~50 us before the 7x7 / 14x14 ones, which are the long pole of stages 3 and 4 -- so the
    // layer is split by WHEN its inputs exist:
    //   early: every down chain that starts at a branch b <= nb-3 runs to its end on that branch's own stream, right behind the
    //          branch's last convolution (no cross-stream hop), as plain convolutions D_ij (no addend, no ReLU after the last one);
    //          the first convolutions of the chains (i,0), i >= 2 (32 -> 32, ReLU) share their input and are one launch;
    //   late:  ONE grouped launch (Op::FUSEUP, hr_fuse.hip) finishes outputs 0 .. nb-2 -- all 1x1 up terms, x_i, the D_ij, ReLU --
    //          and ONE stride-2 convolution from branch nb-2 finishes output nb-1 (adds x_{nb-1} and the D_{nb-1,j}, ReLU).
    // After the last branch output exists, every output of the module is ONE launch away (round 3: 1x1 launch -> sum / finishing
    // convolution, up to four dependent launches with a cross-stream event between each pair).
    // Stage 4: 8 launches per fuse layer (round 3: 17), stage 3: 4 (8), stage 2: 2 (3).
    std::vector<View> hr_fuse_grouped(const std::vector<View>& xs, const std::string& p, const View* out0, const std::vector<int>& branch_tail) {
        const int nb = (int)xs.size();
        std::vector<View> outs(nb);
        auto key = [&](int i, int j, int level) { return p + "fuse_layers." + std::to_string(i) + "." + std::to_string(j) + "." + std::to_string(level) + "."; };
        std::vector<std::vector<View>> d(nb, std::vector<View>(nb));       // running tensor of chain (i,j)
        std::vector<std::vector<int>> d_op(nb, std::vector<int>(nb, -1));  // its latest launch
        for (int i = 1; i < nb; ++i)
            for (int j = 0; j < i; ++j) { d[i][j] = xs[j]; d_op[i][j] = branch_tail[j]; }
        auto follow_last = [&](int op_idx) { ops.back().follow = op_idx; };
        // early: chains from branches 0 .. nb-3 (and, for outputs < nb-1, from branch nb-2 too: D_{i,i-1} with i <= nb-2 starts at a branch <= nb-3).
        // The first convolutions of all chains that start at one branch share their input and are ONE launch: the linear one ((j+1, j): the whole chain
        // of output j+1, no ReLU) first, then the ReLU'd first links of the longer chains (ConvLayer::relu_from)
        for (int j = 0; j < nb - 1; ++j)
            for (int level = 0; level < nb - 1 - j; ++level) {
                std::vector<int> members;                                   // outputs i whose chain (i, j) has a convolution at this level
                for (int i = j + 1; i < nb; ++i)
                    if (level < i - j && !(i == nb - 1 && j == nb - 2)) members.push_back(i);
                // GRNET_FUSE_MERGE (A/B switch): 2 = all first convolutions of a branch in one launch, 1 = only the ReLU'd ones, 0 = none
                // (the bf16 kernels have no per-segment ReLU: every first convolution is its own launch there)
                static const int merge_env_f32 = GRNET_AB(FUSE_MERGE, 2);
                const int merge_env = dtype == 1 ? 0 : merge_env_f32;
                std::vector<int> solo;
                if (level == 0 && merge_env < 2) {
                    std::vector<int> keep;
                    for (int i : members) (merge_env == 1 && i - j >= 2 ? keep : solo).push_back(i);
                    members.swap(keep);
                }
                if (level == 0 && members.size() >= 2) {
                    std::vector<ConvSeg> segs;
                    int lin = 0;
                    for (int i : members) {
                        const bool last = i - j == 1;
                        segs.push_back(ConvSeg{key(i, j, 0) + "0.weight", key(i, j, 0) + "1", "", last ? kBranchCh[i] : kBranchCh[j]});
                        if (last) lin += kBranchCh[i];
                    }
                    cur_lane = j;
                    View m = add_conv(xs[j], segs, 3, 2, true);
                    convs.back().relu_from = lin;                            // members are in ascending i: the linear segment (i = j + 1), if any, comes first
                    follow_last(branch_tail[j]);
                    int off = 0;
                    for (int i : members) {
                        const int c = i - j == 1 ? kBranchCh[i] : kBranchCh[j];
                        d[i][j] = slice(m, off, c);
                        d_op[i][j] = (int)ops.size() - 1;
                        off += c;
                    }
                    members.clear();
                }
                members.insert(members.begin(), solo.begin(), solo.end());
                for (int i : members) {
                    const bool last = level == i - j - 1;
                    cur_lane = j;
                    d[i][j] = conv_bn(d[i][j], key(i, j, level) + "0.weight", key(i, j, level) + "1", last ? kBranchCh[i] : kBranchCh[j], 3, 2, !last);
                    follow_last(d_op[i][j]);
                    d_op[i][j] = (int)ops.size() - 1;
                }
            }
        // late: the grouped launch for outputs 0 .. nb-2 ...
        FuseUpPlan fp;
        fp.nb = nb; fp.prefix = p; fp.xs = xs;
        for (int i = 0; i < nb - 1; ++i) {
            outs[i] = (i == 0 && out0) ? *out0 : new_buffer(kBranchCh[i], xs[i].h, xs[i].w);
            fp.outs.push_back(outs[i]);
            fp.extra.push_back({});
            for (int j = 0; j < i; ++j) fp.extra.back().push_back(d[i][j]);
        }
        for (int i = 0; i < nb - 1; ++i)
            for (int j = i + 1; j < nb; ++j) fp.macs_per_frame += (double)xs[j].h * xs[j].w * kBranchCh[j] * kBranchCh[i];
        fuse_ups.push_back(fp);
        Op op;
        op.kind = Op::FUSEUP;
        op.conv_idx = (int)fuse_ups.size() - 1;
        op.lane = cur_lane = nb - 1;
        op.follow = branch_tail[nb - 1];
        ops.push_back(op);
        // ... and the stride-2 convolution from branch nb-2 that finishes output nb-1
        {
            const int i = nb - 1;
            std::vector<AddRef> adds;
            adds.push_back(AddRef{xs[i], 0});
            for (int j = 0; j < i - 1; ++j) adds.push_back(AddRef{d[i][j], 0});
            cur_lane = nb - 2;
            outs[i] = conv_bn(xs[i - 1], key(i, i - 1, 0) + "0.weight", key(i, i - 1, 0) + "1", kBranchCh[i], 3, 2, true, adds);
            follow_last(branch_tail[i - 1]);
        }
        return outs;
    }
    std::vector<FuseUpPlan> fuse_ups;
    std::vector<ChainPlan> chains;
    std::vector<RollPlan> rolls;
    // the convolutions added last (in order) become ONE row-walking launch in large bf16 calls; the members keep their own ops (small calls launch them one by
    // one), pinned to the launcher's stream in order -- the mechanism of the BasicBlock chains
    void add_roll(int kind, int n_convs) {
        RollPlan rp;
        rp.kind = kind;
        const int first = (int)convs.size() - n_convs;
        for (int i = 0; i < n_convs; ++i) {
            convs[first + i].roll = (int)rolls.size();
            convs[first + i].roll_pos = i;
            rp.convs.push_back(first + i);
        }
        int prev_op = -1;
        for (int i = 0; i < (int)ops.size(); ++i)
            if (ops[i].kind == Op::CONV && ops[i].conv_idx >= first) {
                if (prev_op >= 0) ops[i].follow = prev_op;
                prev_op = i;
            }
        rolls.push_back(rp);
    }

    // Fuse layer as launched until round 3 (kept for the bf16 path and for A/B runs)
    // up0 (bf16, round 5): output 0 -- the full-resolution one, 4 of the layer's launches -- is finished by ONE hr_fuse_up_bf16 launch instead (only = 0);
    // the other outputs keep their finishing stride-2 convolution, which adds everything in its epilogue and writes no D_ij to memory
    std::vector<View> hr_fuse_separate(std::vector<View> xs, const std::string& p, const View* out0, bool up0 = false, const std::vector<int>& branch_tail = {}) {
        const int nb = (int)xs.size();
        // up terms t[i][j], j > i: conv1x1 + BN at the resolution of branch j (nearest upsample is
        // applied where the term is consumed: it commutes with the per-pixel conv/BN)
        std::vector<std::vector<View>> t(nb, std::vector<View>(nb));
        // the up terms W_ij x_j of ONE source branch j (linear 1x1 convolutions at the source's resolution) share their input: one launch, output channels side by side
        // (round 5; 31 -> 18 launches of the 1x1 terms per forward, 10.28 against 10.34 ms at 256 frames bf16; GRNET_FUSE_MERGE_UP=0: one launch per term)
        static const int merge_up_env = GRNET_AB(FUSE_MERGE_UP, 1);
        std::vector<std::vector<char>> tdone(nb, std::vector<char>(nb, 0));
        if (merge_up_env)
            for (int j = 1; j < nb; ++j) {
                std::vector<int> members;
                for (int i = (up0 ? 1 : 0); i < j; ++i) members.push_back(i);
                if (members.size() < 2) continue;
                std::vector<ConvSeg> segs;
                for (int i : members) {
                    const std::string q = p + "fuse_layers." + std::to_string(i) + "." + std::to_string(j) + ".";
                    segs.push_back(ConvSeg{q + "0.weight", q + "1", "", kBranchCh[i]});
                }
                cur_lane = j;
                View m = add_conv(xs[j], segs, 1, 1, false);
                int off = 0;
                for (int i : members) { t[i][j] = slice(m, off, kBranchCh[i]); off += kBranchCh[i]; tdone[i][j] = 1; }
            }
        for (int i = 0; i < nb; ++i)
            for (int j = i + 1; j < nb; ++j) {
                if (up0 && i == 0) continue;
                if (tdone[i][j]) continue;
                const std::string q = p + "fuse_layers." + std::to_string(i) + "." + std::to_string(j) + ".";
                cur_lane = j;
                t[i][j] = conv_bn(xs[j], q + "0.weight", q + "1", kBranchCh[i], 1, 1, false);
            }
        cur_lane = 0;
        std::vector<View> outs(nb);
        if (up0) {
            FuseUpPlan fp;
            fp.nb = nb; fp.prefix = p; fp.xs = xs; fp.only = 0;
            outs[0] = out0 ? *out0 : new_buffer(kBranchCh[0], xs[0].h, xs[0].w);
            fp.outs.push_back(outs[0]);
            fp.extra.push_back({});
            for (int j = 1; j < nb; ++j) fp.macs_per_frame += (double)xs[j].h * xs[j].w * kBranchCh[j] * kBranchCh[0];
            fuse_ups.push_back(fp);
            Op op;
            op.kind = Op::FUSEUP;
            op.conv_idx = (int)fuse_ups.size() - 1;
            op.lane = 0;
            op.follow = branch_tail[0];
            ops.push_back(op);
        } else {   // output 0: elementwise sum of the identity and the upsampled terms
            View o = out0 ? *out0 : new_buffer(kBranchCh[0], xs[0].h, xs[0].w);
            Op op;
            op.kind = Op::SUM;
            op.lane = 0;
            SumArgs& sa = op.sum;
            sa.C = kBranchCh[0]; sa.H = xs[0].h; sa.W = xs[0].w; sa.relu = 1;
            sa.n_add = nb;
            sum_views.push_back({o, {}});
            sum_views.back().second.push_back(AddRef{xs[0], 0});
            for (int j = 1; j < nb; ++j) sum_views.back().second.push_back(AddRef{t[0][j], j});
            op.conv_idx = (int)sum_views.size() - 1;
            ops.push_back(op);
            outs[0] = o;
        }
        // down paths (all 3x3 stride 2), by dependency level: chain conv k of (i,j) is level k; the conv that
        // finishes output i (the single stride-2 conv from branch i-1, which also adds the identity, the
        // finished down chains and the upsampled terms, then applies the ReLU) is level 0 for i = 1, else i.
        std::vector<std::vector<View>> d(nb, std::vector<View>(nb));       // running tensor of chain (i,j)
        for (int i = 2; i < nb; ++i)
            for (int j = 0; j < i - 1; ++j) d[i][j] = xs[j];
        for (int level = 0; level < nb; ++level) {
            // the ReLU'd first links of the chains that start at ONE branch share their input: one launch with their output channels side by side (round 5, bf16 as well:
            // stage 4's chains (2,0) and (3,0) read the 56x56 branch once instead of twice).  GRNET_FUSE_MERGE_FIRST=0: one launch per chain
            static const int merge_first_env = GRNET_AB(FUSE_MERGE_FIRST, 1);
            std::vector<std::vector<char>> merged(nb, std::vector<char>(nb, 0));
            if (level == 0 && merge_first_env)
                for (int j = 0; j < nb - 2; ++j) {
                    std::vector<int> members;
                    for (int i = j + 2; i < nb; ++i)
                        if (i - j - 1 > 0) members.push_back(i);                   // chain (i, j) has more than one link: its first link is ReLU'd, kBranchCh[j] channels
                    if (members.size() < 2) continue;
                    std::vector<ConvSeg> segs;
                    for (int i : members) {
                        const std::string q = p + "fuse_layers." + std::to_string(i) + "." + std::to_string(j) + ".0.";
                        segs.push_back(ConvSeg{q + "0.weight", q + "1", "", kBranchCh[j]});
                    }
                    cur_lane = j;
                    View m = add_conv(xs[j], segs, 3, 2, true);
                    int off = 0;
                    for (int i : members) { d[i][j] = slice(m, off, kBranchCh[j]); off += kBranchCh[j]; merged[i][j] = 1; }
                }
            for (int i = 2; i < nb; ++i)
                for (int j = 0; j < i - 1; ++j) {
                    if (level >= i - j) continue;
                    if (merged[i][j]) continue;
                    const bool last = level == i - j - 1;
                    cur_lane = j;
                    const std::string q = p + "fuse_layers." + std::to_string(i) + "." + std::to_string(j) + "." + std::to_string(level) + ".";
                    d[i][j] = conv_bn(d[i][j], q + "0.weight", q + "1", last ? kBranchCh[i] : kBranchCh[j], 3, 2, !last);
                }
            for (int i = 1; i < nb; ++i) {
                if ((i == 1 ? 0 : i) != level) continue;
                std::vector<AddRef> adds;
                adds.push_back(AddRef{xs[i], 0});
                for (int j = 0; j < i - 1; ++j) adds.push_back(AddRef{d[i][j], 0});
                for (int j = i + 1; j < nb; ++j) adds.push_back(AddRef{t[i][j], j - i});
                cur_lane = i;
                const std::string q = p + "fuse_layers." + std::to_string(i) + "." + std::to_string(i - 1) + ".0.";
                outs[i] = conv_bn(xs[i - 1], q + "0.weight", q + "1", kBranchCh[i], 3, 2, true, adds);
            }
        }
        cur_lane = 0;
        return outs;
    }
    std::vector<std::pair<View, std::vector<AddRef>>> sum_views;
    std::vector<std::pair<std::string, View>> named;   // intermediate tensors exposed to grnet_debug_tensor
    void name_view(const std::string& n, const View& v) { named.emplace_back(n, v); }

    void build_plan() {
        const std::string b = "backbone.";
        v_input.p = nullptr; v_input.ctot = 3; v_input.coff = 0; v_input.c = 3; v_input.h = 224; v_input.w = 224;
        View in = v_input;
        in.p = reinterpret_cast<float*>(~(uintptr_t)0);   // tag: caller's frames pointer
        // bf16: the stem's first convolution reads the caller's fp32 frames itself (conv_bf16_stem, round 4); GRNET_BF16_STEM=0 restores the
        // conversion launch -- frames (N,3,224,224) f32 -> NHWC bf16, 8 channels per pixel -- in front of the generic kernel
        static const int bf16_stem_env = GRNET_AB(BF16_STEM, 1);
        bf16_stem = dtype == 1 && bf16_stem_env;
        if (dtype == 1 && !bf16_stem) {
            v_in8 = new_buffer(8, 224, 224);
            Op cv;
            cv.kind = Op::CONVERT;
            ops.push_back(cv);
            in = v_in8;
        }
        solo_region = true;
        View x = conv_bn(in, b + "conv1.weight", b + "bn1", 64, 3, 2, true);
        name_view("stem_conv1", x);
        x = conv_bn(x, b + "conv2.weight", b + "bn2", 64, 3, 2, true);
        name_view("stem_conv2", x);
        if (bf16_stem) add_roll(0, 2);
        int prev_conv3 = -1;
        for (int k = 0; k < 4; ++k) {                       // layer1: 4 Bottlenecks (hrnet.py:80-100)
            const std::string q = b + "layer1." + std::to_string(k) + ".";
            // bf16: Bottleneck k-1's expansion and this one's reduction are a PAIR (one launch in large calls): the reduction is the first convolution added below
            const int first_new = (int)convs.size() + ((k == 0 && !(dtype == 1 && (GRNET_AB(BF16_MERGE_DS, 1)))) ? 1 : 0);
            struct PairAtExit {
                grnet* g; int& prev; int first_new;
                ~PairAtExit() {
                    if (g->dtype == 1 && prev >= 0 && first_new < (int)g->convs.size() && g->convs[first_new].ks == 1 && g->convs[first_new].in.c == 256 && g->convs[first_new].cout == 64) {
                        g->convs[prev].pair_next = first_new;
                        g->convs[first_new].pair_of = prev;
                        int op_prev = -1, op_new = -1;                 // the member launches nothing in large calls: it shares the expansion's stream, so that a graph
                        for (int i = 0; i < (int)g->ops.size(); ++i) {  // recorded from this plan hangs the member's consumers on the expansion's node (round-5 advice)
                            if (g->ops[i].kind == Op::CONV && g->ops[i].conv_idx == prev) op_prev = i;
                            if (g->ops[i].kind == Op::CONV && g->ops[i].conv_idx == first_new) op_new = i;
                        }
                        if (op_prev >= 0 && op_new >= 0) g->ops[op_new].follow = op_prev;
                    }
                    prev = (int)g->convs.size() - 1;         // this Bottleneck's conv3 is the last convolution added
                }
            } pair_at_exit{this, prev_conv3, first_new};
            // bf16, first Bottleneck: relu(BN3(conv3(t)) + BNd(downsample(x))) is ONE 1x1 GEMM over the concatenated inputs [t ; x] (K = 64 + 64, the two
            // BatchNorms folded into their halves of the weights, the shifts summed): the 411 MB downsample tensor (at 256 frames) is neither written nor read
            // back, and a launch goes away.  GRNET_BF16_MERGE_DS=0: the two launches of the reference's graph (hrnet.py:80-100, 389-406).
            static const int merge_ds = GRNET_AB(BF16_MERGE_DS, 1);
            if (k == 0 && dtype == 1 && merge_ds) {
                View y = conv_bn(x, q + "conv1.weight", q + "bn1", 64, 1, 1, true);
                y = conv_bn(y, q + "conv2.weight", q + "bn2", 64, 3, 1, true);
                View xin = x;
                x = conv_bn(y, q + "conv3.weight", q + "bn3", 256, 1, 1, true);
                convs.back().in2 = xin;
                convs.back().seg2 = ConvSeg{q + "downsample.0.weight", q + "downsample.1", "", 256};
                convs.back().macs_per_frame *= 2;                  // K = 64 (t) + 64 (x)
                add_roll(1, 3);
                name_view("layer1.0", x);
                continue;
            }
            View res = k == 0 ? conv_bn(x, q + "downsample.0.weight", q + "downsample.1", 256, 1, 1, false) : x;
            View y = conv_bn(x, q + "conv1.weight", q + "bn1", 64, 1, 1, true);
            y = conv_bn(y, q + "conv2.weight", q + "bn2", 64, 3, 1, true);
            x = conv_bn(y, q + "conv3.weight", q + "bn3", 256, 1, 1, true, {AddRef{res, 0}});
            if (dtype == 1 && k > 0) add_roll(2, 3);
            name_view("layer1." + std::to_string(k), x);
        }
        name_view("layer1", x);
        solo_region = false;
        std::vector<View> xs;
        xs.push_back(conv_bn(x, b + "transition1.0.0.weight", b + "transition1.0.1", 32, 3, 1, true));
        cur_lane = 1;
        xs.push_back(conv_bn(x, b + "transition1.1.0.0.weight", b + "transition1.1.0.1", 64, 3, 2, true));
        cur_lane = 0;
        xs = hr_module(xs, b + "stage2.0.", nullptr);
        for (size_t i = 0; i < xs.size(); ++i) name_view("stage2." + std::to_string(i), xs[i]);
        cur_lane = 2;
        xs.push_back(conv_bn(xs.back(), b + "transition2.2.0.0.weight", b + "transition2.2.0.1", 128, 3, 2, true));
        cur_lane = 0;
        for (int m = 0; m < 4; ++m) xs = hr_module(xs, b + "stage3." + std::to_string(m) + ".", nullptr);
        for (size_t i = 0; i < xs.size(); ++i) name_view("stage3." + std::to_string(i), xs[i]);
        cur_lane = 3;
        xs.push_back(conv_bn(xs.back(), b + "transition3.3.0.0.weight", b + "transition3.3.0.1", 256, 3, 2, true));
        cur_lane = 0;
        v_cat = new_buffer(480, 56, 56);                    // torch.cat([x0, x1, x2, x3], 1) (hrnet.py:524)
        for (int m = 0; m < 3; ++m) {
            View o0 = slice(v_cat, 0, 32);
            xs = hr_module(xs, b + "stage4." + std::to_string(m) + ".", m == 2 ? &o0 : nullptr);
        }
        for (size_t i = 0; i < xs.size(); ++i) name_view("stage4." + std::to_string(i), xs[i]);
        int coff = 32;
        for (int idx = 2; idx <= 4; ++idx) {                // upsample heads (hrnet.py:440-453,521-523)
            const int br = idx - 1, c = kBranchCh[br], n_layers = idx - 1;
            cur_lane = br;                                  // the three upsample heads are independent
            View t = xs[br];
            for (int l = 0; l < n_layers; ++l) {
                const std::string q = b + "upsample_stage_" + std::to_string(idx) + ".";
                View up = add_bilinear(t);
                name_view("up" + std::to_string(idx) + "." + std::to_string(l) + ".bilinear", up);
                View dst = slice(v_cat, coff, c);
                t = conv_bn(up, q + std::to_string(4 * l + 1) + ".weight", q + std::to_string(4 * l + 2), c, 3, 1, true, {},
                            l == n_layers - 1 ? &dst : nullptr);
                name_view("up" + std::to_string(idx) + "." + std::to_string(l) + ".conv", t);
            }
            coff += c;
        }
        cur_lane = 0;
        // PARE head (pare.py:305-336).  The two 480->128 first convolutions read the same input and are
        // issued as one 480->256 convolution writing both halves of one buffer.
        const std::string hd = "head.";
        solo_region = true;
        View first = add_conv(v_cat,
                              {ConvSeg{hd + "keypoint_deconv_layers.0.weight", hd + "keypoint_deconv_layers.1", "", 128},
                               ConvSeg{hd + "smpl_deconv_layers.0.weight", hd + "smpl_deconv_layers.1", "", 128}},
                              3, 1, true);
        View part_feats = conv_bn(slice(first, 0, 128), hd + "keypoint_deconv_layers.3.weight", hd + "keypoint_deconv_layers.4", 128, 3, 1, true);
        v_heat = add_conv(part_feats, {ConvSeg{hd + "keypoint_final_layer.weight", "", hd + "keypoint_final_layer.bias", 25}}, 1, 1, false);
        cur_lane = 1;                                       // the 3D branch runs beside the 2D branch
        v_smpl_feats = conv_bn(slice(first, 128, 128), hd + "smpl_deconv_layers.3.weight", hd + "smpl_deconv_layers.4", 128, 3, 1, true);
        v_csmap = add_conv(v_smpl_feats, {ConvSeg{hd + "smpl_final_layer.weight", "", hd + "smpl_final_layer.bias", 64}}, 1, 1, false);
        cur_lane = 0;
        solo_region = false;
        Op op;
        op.kind = Op::POOL; ops.push_back(op);
        op.kind = Op::TAIL; ops.push_back(op);
        op.kind = Op::SMPL; ops.push_back(op);
    }

    float* resolve_ptr(float* tag) const {
        const uintptr_t t = reinterpret_cast<uintptr_t>(tag);
        if (t == ~(uintptr_t)0 || t == 0) return tag;
        auto it = view_slot.find((int)(t - 1));
        return *slots[it->second];
    }
    void resolve(View& v) const { v.p = resolve_ptr(v.p); }

    int dev_alloc(float** p, size_t floats) {
        void* q = nullptr;
        if (hipMalloc(&q, floats * sizeof(float)) != hipSuccess) return fail(GRNET_ENOMEM, "hipMalloc failed");
        dev_allocs.push_back(q);
        *p = static_cast<float*>(q);
        return 0;
    }

    int allocate() {
        size_t total = 64;   // leading zero block
        std::vector<size_t> offs;
        for (auto& pr : pending) {
            offs.push_back(total);
            size_t fl = pr.second * (size_t)max_frames;
            total += (fl + 63) / 64 * 64;               // 256-byte aligned buffers
        }
        total += 64;                                    // 256 bytes of tail: conv_wino4s_f32's 16-byte row loads on 7-wide maps touch (and mask) one float past a row,
                                                        // i.e. 4 bytes past the LAST buffer's end for its last row -- they stay inside the arena
        arena_floats = total;
        void* q = nullptr;
        if (hipMalloc(&q, total * sizeof(float)) != hipSuccess)
            return fail(GRNET_ENOMEM, "hipMalloc of the activation arena (" + std::to_string(total * 4 >> 20) + " MiB) failed");
        arena = static_cast<float*>(q);
        if (hipMemset(arena, 0, 64 * sizeof(float)) != hipSuccess) return fail(GRNET_EHIP, "hipMemset failed");
        zeros = arena;
        for (size_t i = 0; i < pending.size(); ++i) *pending[i].first = arena + offs[i];
        for (auto& L : convs) {
            resolve(L.in); resolve(L.out);
            if (L.in2.c) resolve(L.in2);
            for (auto& a : L.adds) resolve(a.v);
        }
        for (auto& op : ops) { resolve(op.bin); resolve(op.bout); }
        for (auto& sv : sum_views) { resolve(sv.first); for (auto& a : sv.second) resolve(a.v); }
        for (auto& fp : fuse_ups) {
            for (auto& v : fp.xs) resolve(v);
            for (auto& v : fp.outs) resolve(v);
            for (auto& e : fp.extra) for (auto& v : e) resolve(v);
        }
        for (auto& nv : named) resolve(nv.second);
        resolve(v_cat); resolve(v_heat); resolve(v_smpl_feats); resolve(v_csmap);
        if (dtype == 1 && !bf16_stem) resolve(v_in8);
        // streams / events of the parallel lanes are created here, never inside a stream capture
        if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) return fail(GRNET_EHIP, "hipEventCreate failed");
        if (int rc = install_schedule(max_frames)) return rc;
        if (hipStreamCreateWithFlags(&capture_stream, hipStreamNonBlocking) != hipSuccess) return fail(GRNET_EHIP, "hipStreamCreate failed");
        const size_t n = max_frames;
        int rc;
        if ((rc = dev_alloc(&d_plf, n * 128 * 24))) return rc;
        if ((rc = dev_alloc(&d_csf, n * 64 * 24))) return rc;
        if ((rc = dev_alloc(&d_stats, softmax_pool_ws_floats((int)n)))) return rc;
        if ((rc = dev_alloc(&d_rot6d, n * 144))) return rc;
        if ((rc = dev_alloc(&d_shape, n * 10))) return rc;
        if ((rc = dev_alloc(&d_cam, n * 3))) return rc;
        if ((rc = dev_alloc(&d_rotmat, n * 216))) return rc;
        if ((rc = dev_alloc(&d_theta, n * 85))) return rc;
        if ((rc = dev_alloc(&d_A, n * kSmplWsFloatsPerFrame))) return rc;
        if ((rc = dev_alloc(&d_verts, n * 6890 * 3))) return rc;
        if ((rc = dev_alloc(&d_kp3d, n * 87))) return rc;
        if ((rc = dev_alloc(&d_kp2d, n * 58))) return rc;
        return 0;
    }

    // Read-after-write edges between lanes.  Every op writes a buffer nobody has written before (no
    // buffer reuse; the writers of the concat buffer own disjoint channel slices), so RAW edges are the
    // only hazards inside one forward; forwards are separated by the join at the end of enqueue().
    // Buffers an op reads / the buffer it writes (nullptr: caller-owned outputs).
    void op_reads(const Op& op, std::vector<const float*>& r) const {
        r.clear();
        switch (op.kind) {
            case Op::CONV: {
                const ConvLayer& L = convs[op.conv_idx];
                r.push_back(L.in.p);
                if (L.in2.c) r.push_back(L.in2.p);
                for (auto& a : L.adds) r.push_back(a.v.p);
                break;
            }
            case Op::SUM:
                for (auto& a : sum_views[op.conv_idx].second) r.push_back(a.v.p);
                break;
            case Op::BILINEAR: r.push_back(op.bin.p); break;
            case Op::FUSEUP:
            {
                const FuseUpPlan& fp = fuse_ups[op.conv_idx];
                for (size_t j = fp.only < 0 ? 0 : fp.only; j < fp.xs.size(); ++j) r.push_back(fp.xs[j].p);
                for (size_t i = 0; i < fp.extra.size(); ++i)
                    if (fp.only < 0 || fp.only == (int)i) for (auto& v : fp.extra[i]) r.push_back(v.p);
            }
                break;
            case Op::POOL: r.push_back(v_heat.p); r.push_back(v_smpl_feats.p); r.push_back(v_csmap.p); break;
            default: break;                                     // TAIL / SMPL follow POOL on lane 0
        }
    }
    void op_writes(const Op& op, std::vector<const float*>& w) const {
        w.clear();
        if (op.kind == Op::CONV) w.push_back(convs[op.conv_idx].out.p);
        else if (op.kind == Op::SUM) w.push_back(sum_views[op.conv_idx].first.p);
        else if (op.kind == Op::BILINEAR) w.push_back(op.bout.p);
        else if (op.kind == Op::CONVERT) w.push_back(v_in8.p);
        else if (op.kind == Op::FUSEUP) {
            const FuseUpPlan& fp = fuse_ups[op.conv_idx];
            for (size_t i = 0; i < fp.outs.size(); ++i) if (fp.only < 0 || fp.only == (int)i) w.push_back(fp.outs[i].p);
        }
    }

    // Static list scheduling of the op list onto the kLanes streams.  The plan writes "branch b on lane b",
    // which leaves the fuse layer of an HR module as a chain of small launches on the lane of the slowest branch
    // (measured: ~210 us per stage-4 module in which mostly one small kernel runs at a time).  Here every op gets an
    // estimated duration, and ops are placed earliest-start-first (ties: longest remaining path first) on the lane
    // that lets them start first, preferring the lane of their latest producer (no cross-lane event).  Streams are FIFO,
    // so the resulting list is both the enqueue order and a topological order; analyze_dependencies() then derives
    // the cross-lane events from it exactly as for the hand-written lanes.
    // Build ops_flat from the plan: lane placement, cross-lane events, the streams the schedule uses.
    // (Round 4 also re-placed the lanes at tune time from the durations grnet_op_timeline measures in company: 3.728 -> 3.784 ms and
    // 3.735 -> 3.757 ms per step, i.e. no better than the calibrated estimates below; removed.)
    int install_schedule(int n) {
        drop_graphs();
        seen_once.clear();
        for (hipEvent_t e : op_events_flat) if (e) (void)hipEventDestroy(e);
        op_events_flat.clear();
        ops_flat = ops;
        static const int sched_env = GRNET_AB(LANE_SCHED, 1);   // 0: lanes as written in the plan
        if (sched_env) schedule_lanes(ops_flat, n);
        analyze_dependencies(ops_flat, op_events_flat);
        int used = 1;                                          // only the streams the schedule really uses are forked / joined
        for (const Op& op : ops_flat) used = std::max(used, op.lane + 1);
        for (int l = 1; l < used; ++l) {
            if (!side[l] && hipStreamCreateWithFlags(&side[l], hipStreamNonBlocking) != hipSuccess) return fail(GRNET_EHIP, "hipStreamCreate failed");
            if (!ev_join[l] && hipEventCreateWithFlags(&ev_join[l], hipEventDisableTiming) != hipSuccess) return fail(GRNET_EHIP, "hipEventCreate failed");
        }
        lanes_used = used;
        for (size_t i = 0; i < ops_flat.size(); ++i)
            if (ops_flat[i].record && hipEventCreateWithFlags(&op_events_flat[i], hipEventDisableTiming) != hipSuccess)
                return fail(GRNET_EHIP, "hipEventCreate failed");
        return 0;
    }

    void schedule_lanes(std::vector<Op>& list, int n) const {
        const int m = (int)list.size();
        std::vector<double> est(m), blevel(m, 0.0);
        std::vector<std::vector<int>> deps(m), users(m);
        std::map<const float*, std::vector<int>> writers;
        std::vector<const float*> r, wr;
        int prev_tail = -1;
        static const double fix_us = GRNET_AB_F(SCHED_FIX, 6.0);
        static const double hop_us = GRNET_AB_F(SCHED_HOP, 4.0);
        for (int i = 0; i < m; ++i) {
            const Op& op = list[i];
            switch (op.kind) {
                case Op::CONV: {
                    const double gf = 2.0 * convs[op.conv_idx].macs_per_frame * n / 1e9;
                    est[i] = fix_us + gf / (gf > 20 ? 0.105 : gf > 3 ? 0.085 : 0.060);     // us; GFLOP per us = TFLOP/s / 1000
                    // launches that run beside three others (everything between transition1 and the heads): measured in company at 16
                    // frames (grnet_op_timeline) the four branch convolutions of a module take 19 / 23 / 22 / 32 us for the SAME FLOPs
                    // (56x56 ... 7x7: the 7x7 chain is the long pole of stage 4), the stride-2 and small launches 17-23 us
                    if (!convs[op.conv_idx].solo && gf < 3) {
                        const ConvLayer& L = convs[op.conv_idx];
                        est[i] = std::max(est[i], 17.0);
                        if (L.ks == 3 && L.stride == 1 && L.in.c == L.cout) est[i] *= L.in.w == 7 ? 1.45 : L.in.w == 56 ? 0.9 : 1.05;
                    }
                    break;
                }
                case Op::FUSEUP: est[i] = 18; break;
                case Op::POOL: est[i] = 50; break;
                case Op::TAIL: est[i] = 50; break;
                case Op::SMPL: est[i] = 60; break;
                default: est[i] = 8; break;
            }
            op_reads(op, r);
            for (const float* b : r) {
                auto it = writers.find(b);
                if (it == writers.end()) continue;
                for (int w : it->second)
                    if (std::find(deps[i].begin(), deps[i].end(), w) == deps[i].end()) deps[i].push_back(w);
            }
            if (op.kind == Op::POOL || op.kind == Op::TAIL || op.kind == Op::SMPL) {   // the tail is a chain on the caller's stream
                if (prev_tail >= 0) deps[i].push_back(prev_tail);
                prev_tail = i;
            }
            op_writes(op, wr);
            for (const float* o : wr) writers[o].push_back(i);
        }
        for (int i = 0; i < m; ++i)
            for (int d : deps[i]) users[d].push_back(i);
        for (int i = m - 1; i >= 0; --i) {
            double b = 0;
            for (int u : users[i]) b = std::max(b, blevel[u]);
            blevel[i] = b + est[i];
        }
        std::vector<int> pending(m), lane_of(m, 0), order;
        std::vector<double> finish(m, 0.0);
        std::vector<char> done(m, 0);
        for (int i = 0; i < m; ++i) pending[i] = (int)deps[i].size();
        double lane_free[kLanes] = {};
        static const int n_lanes = std::min(kLanes, std::max(1, GRNET_AB(LANES, 4)));
        order.reserve(m);
        for (int step = 0; step < m; ++step) {
            int best = -1, best_lane = 0;
            double best_start = 0;
            for (int i = 0; i < m; ++i) {
                if (done[i] || pending[i]) continue;
                double ready = 0;
                int from = -1;
                for (int d : deps[i])
                    if (finish[d] >= ready) { ready = finish[d]; from = d; }
                const bool pinned = list[i].kind == Op::POOL || list[i].kind == Op::TAIL || list[i].kind == Op::SMPL;
                int lane = 0;
                double start = std::max(ready, lane_free[0]);
                if (!pinned && list[i].follow >= 0) {             // shares the stream of the op it follows
                    lane = lane_of[list[i].follow];
                    start = std::max(ready, lane_free[lane]);
                } else if (!pinned) {
                    const int pref = from >= 0 ? lane_of[from] : 0;
                    lane = pref;
                    start = std::max(ready, lane_free[pref]);
                    for (int l = 0; l < n_lanes; ++l) {
                        const double st = std::max(ready, lane_free[l]);
                        if (st + hop_us < start) { start = st; lane = l; }   // a cross-lane hop costs an event
                    }
                }
                if (best < 0 || start < best_start - 1e-9 || (start < best_start + 1e-9 && blevel[i] > blevel[best])) {
                    best = i; best_lane = lane; best_start = start;
                }
            }
            done[best] = 1;
            lane_of[best] = best_lane;
            finish[best] = best_start + est[best];
            lane_free[best_lane] = finish[best];
            for (int u : users[best]) --pending[u];
            order.push_back(best);
        }
        std::vector<Op> out;
        out.reserve(m);
        for (int i : order) {
            Op op = list[i];
            op.lane = lane_of[i];
            op.waits.clear();
            op.record = false;
            out.push_back(std::move(op));
        }
        if (getenv("GRNET_TRACE")) fprintf(stderr, "[grnet] lane schedule: %d ops, estimated makespan %.0f us (sum of estimates %.0f us)\n", m,
                                           *std::max_element(lane_free, lane_free + kLanes), [&] { double t = 0; for (double e : est) t += e; return t; }());
        list.swap(out);
    }

    void analyze_dependencies(std::vector<Op>& ops, std::vector<hipEvent_t>& op_events) {
        std::map<const float*, std::vector<int>> writers;      // buffer base -> ops that wrote (part of) it
        std::vector<const float*> r, wr;
        for (int i = 0; i < (int)ops.size(); ++i) {
            Op& op = ops[i];
            op_reads(op, r);
            for (const float* buf : r) {
                auto it = writers.find(buf);
                if (it == writers.end()) continue;              // the caller's frames
                for (int w : it->second)
                    if (ops[w].lane != op.lane) {
                        bool dup = false;
                        for (int x : op.waits) dup |= x == w;
                        if (!dup) op.waits.push_back(w);
                        ops[w].record = true;
                    }
            }
            op_writes(op, wr);
            for (const float* out : wr) writers[out].push_back(i);
        }
        op_events.assign(ops.size(), nullptr);
        if (getenv("GRNET_TRACE")) {
            size_t waits = 0, records = 0;
            for (const Op& op : ops) { waits += op.waits.size(); records += op.record; }
            fprintf(stderr, "[grnet] dependencies: %zu ops, %zu cross-lane waits, %zu recorded events\n", ops.size(), waits, records);
        }
    }

    // ------------------------------------------------------------------ weights
    const HostTensor* find(const std::string& k) const {
        auto it = tensors.find(k);
        return it == tensors.end() ? nullptr : &it->second;
    }

    int upload(const std::vector<float>& h, float** d) {
        int rc = dev_alloc(d, h.size());
        if (rc) return rc;
        if (hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return fail(GRNET_EHIP, "hipMemcpy H2D failed");
        return 0;
    }

    // Fold BN (fp64) and pack to [tap][CinPad][CoutPad].
    int pack_conv(ConvLayer& L) {
        const int cin = L.cin_w, ks = L.ks, taps = ks * ks;
        const int TC = conv_pick_tc(L.cout);
        const bool bf = dtype == 1;                            // bf16: [CinPad/32][tap][CoutPad][32]: a chunk's rows are contiguous for LDS-DMA
        L.cin_pad = bf ? (L.in.c + L.in2.c + 31) / 32 * 32 : (cin + kConvCK - 1) / kConvCK * kConvCK;
        if (L.in2.c && (!bf || ks != 1 || L.in.c % 32 != 0 || L.segs.size() != 1 || L.seg2.cout != L.cout)) return fail(GRNET_ESTATE, "a two-input launch is a bf16 1x1 convolution with one weight segment per input");
        L.cout_pad = bf ? (L.cout + 31) / 32 * 32 : (L.cout + TC - 1) / TC * TC;
        std::vector<float> wp((size_t)taps * L.cin_pad * L.cout_pad, 0.f), bp(L.cout_pad, 0.f);
        // Every eligible 3x3 stride-1 layer takes a Winograd F(4x4,3x3) kernel: on 56x56 maps layer1, upsample heads, PARE head, transition1's
        // 256 -> 32 and the 32 -> 32 convolutions of the HR branch; on 28x28 maps the upsample-head layers and the 64 -> 64 convolutions of
        // the HR branch (conv_wino4.hip); on 14x14 / 7x7 maps the 128- / 256-channel HR branches and the 256 -> 256 upsample-head layer
        // (conv_wino4s.hip).  GRNET_WINO4=0 leaves every layer on the direct kernels (as GRNET_OPT_WINOGRAD = 0 does at run time).
        static const int wino4_env = GRNET_AB(WINO4, 2);
        const bool wino4 = !bf && wino4_env && conv_wino4_eligible(L.in.c, L.cout, L.ks, L.stride, L.in.h, L.in.w, (int)L.adds.size()) && L.cin_pad % 8 == 0 &&
                           L.cout_pad % (L.cout % 64 == 0 ? 64 : 32) == 0 && (L.adds.empty() || L.adds[0].shift == 0) &&
                           (L.in.w == 56 || (L.in.c >= 64 && L.cout % 64 == 0));
        const bool wino4s = !bf && wino4_env && cin == L.in.c && conv_wino4s_eligible(L.in.c, L.cout, L.ks, L.stride, L.in.h, L.in.w, (int)L.adds.size()) &&
                            (L.adds.empty() || L.adds[0].shift == 0);          // the small maps: conv_wino4s.hip
        static const int stem_env = GRNET_AB(STEM, 1);
        const bool stem_shape = cin == L.in.c && L.segs.size() == 1 && conv_stem_eligible(L.in.c, L.cout, L.ks, L.stride, L.in.h, L.in.w, (int)L.adds.size());
        // GRNET_STEM is the fp32 A/B switch only: a bf16 plan built for conv_bf16_stem (GRNET_BF16_STEM) has no NHWC copy of the frames, so its first
        // convolution MUST get the stem kernel's weights whatever GRNET_STEM says (round-4 advice: the generic kernel then read fp32 NCHW frames as NHWC bf16)
        const bool stem = !bf && stem_env && stem_shape, stem_bf = bf && bf16_stem && stem_shape;
        if (bf && bf16_stem && reinterpret_cast<uintptr_t>(L.in.p) == ~(uintptr_t)0 && !stem_bf)
            return fail(GRNET_ESTATE, "bf16 plan without a conversion launch, but its first convolution is not eligible for conv_bf16_stem");
        std::vector<double> wfold(wino4 || wino4s || stem || stem_bf ? (size_t)L.cout * cin * 9 : 0);     // BN-folded weights (cout, cin, 3, 3) for the filter transform
        int co0 = 0;
        for (auto& s : L.segs) {
            const HostTensor* w = find(s.wkey);
            if (!w) return fail(GRNET_ENOENT, "missing tensor " + s.wkey);
            if (w->shape.size() != 4 || w->shape[0] != s.cout || w->shape[1] != cin || w->shape[2] != ks || w->shape[3] != ks)
                return fail(GRNET_EINVAL, "bad shape for " + s.wkey);
            std::vector<double> scale(s.cout, 1.0), shift(s.cout, 0.0);
            if (!s.biaskey.empty()) {
                const HostTensor* bt = find(s.biaskey);
                if (!bt || (int)bt->numel() != s.cout) return fail(GRNET_ENOENT, "missing tensor " + s.biaskey);
                for (int c = 0; c < s.cout; ++c) shift[c] = bt->data[c];
            }
            if (!s.bnprefix.empty()) {
                const HostTensor *g = find(s.bnprefix + ".weight"), *be = find(s.bnprefix + ".bias"),
                                 *m = find(s.bnprefix + ".running_mean"), *v = find(s.bnprefix + ".running_var");
                if (!g || !be || !m || !v) return fail(GRNET_ENOENT, "missing BatchNorm tensors " + s.bnprefix + ".*");
                if ((int)g->numel() != s.cout) return fail(GRNET_EINVAL, "bad BatchNorm size " + s.bnprefix);
                for (int c = 0; c < s.cout; ++c) {
                    const double sc = (double)g->data[c] / std::sqrt((double)v->data[c] + kBnEps);
                    shift[c] = (double)be->data[c] + (shift[c] - (double)m->data[c]) * sc;
                    scale[c] = sc;
                }
            }
            for (int co = 0; co < s.cout; ++co) {
                bp[co0 + co] = (float)shift[co];
                for (int ci = 0; ci < cin; ++ci)
                    for (int t = 0; t < taps; ++t) {
                        const double wv = (double)w->data[((size_t)co * cin + ci) * taps + t] * scale[co];
                        wp[bf ? ((((size_t)(ci / 32) * taps + t) * L.cout_pad + co0 + co) * 32 + ci % 32) : ((size_t)t * L.cin_pad + ci) * L.cout_pad + co0 + co] = (float)wv;
                        if (wino4 || wino4s || stem || stem_bf) wfold[((size_t)(co0 + co) * cin + ci) * 9 + t] = wv;
                    }
            }
            co0 += s.cout;
        }
        if (L.in2.c) {                                         // the second input's 1x1 weights behind the first's input channels, its BatchNorm shift added to the bias
            const ConvSeg& s2 = L.seg2;
            const int cin2 = L.in2.c;
            const HostTensor* w = find(s2.wkey);
            if (!w) return fail(GRNET_ENOENT, "missing tensor " + s2.wkey);
            if (w->shape.size() != 4 || w->shape[0] != s2.cout || w->shape[1] != cin2 || w->shape[2] != 1 || w->shape[3] != 1) return fail(GRNET_EINVAL, "bad shape for " + s2.wkey);
            const HostTensor *g = find(s2.bnprefix + ".weight"), *be = find(s2.bnprefix + ".bias"), *m = find(s2.bnprefix + ".running_mean"), *v = find(s2.bnprefix + ".running_var");
            if (!g || !be || !m || !v) return fail(GRNET_ENOENT, "missing BatchNorm tensors " + s2.bnprefix + ".*");
            if ((int)g->numel() != s2.cout) return fail(GRNET_EINVAL, "bad BatchNorm size " + s2.bnprefix);
            for (int co = 0; co < s2.cout; ++co) {
                const double sc = (double)g->data[co] / std::sqrt((double)v->data[co] + kBnEps);
                bp[co] = (float)((double)bp[co] + (double)be->data[co] - (double)m->data[co] * sc);
                for (int ci = 0; ci < cin2; ++ci) {
                    const int cc = L.in.c + ci;
                    wp[(((size_t)(cc / 32) * taps + 0) * L.cout_pad + co) * 32 + cc % 32] = (float)((double)w->data[(size_t)co * cin2 + ci] * sc);
                }
            }
        }
        int rc;
        if (bf) {                                              // round the folded weights to bf16 (nearest even), two per float slot
            std::vector<float> packed((wp.size() + 1) / 2, 0.f);
            uint16_t* h16 = reinterpret_cast<uint16_t*>(packed.data());
            for (size_t i = 0; i < wp.size(); ++i) h16[i] = f32_to_bf16(wp[i]);
            if ((rc = upload(packed, &L.w_dev))) return rc;
        } else if ((rc = upload(wp, &L.w_dev))) {
            return rc;
        }
        if ((rc = upload(bp, &L.b_dev))) return rc;
        if (stem) {
            std::vector<float> sw(7 * 4 * 64);
            pack_stem_weights(wfold.data(), sw.data());
            if ((rc = upload(sw, &L.stem_dev))) return rc;
        }
        if (stem_bf) {                                          // conv_bf16_stem: 4 x 64 x 8 bf16, two per float slot
            std::vector<float> sw(4 * 64 * 8 / 2);
            pack_stem_weights_bf16(wfold.data(), reinterpret_cast<unsigned short*>(sw.data()));
            if ((rc = upload(sw, &L.stem_dev))) return rc;
        }
        if (wino4s) {                                          // U = G g G^T of the folded filter, fp64 -> fp32
            std::vector<float> uws((size_t)36 * cin * L.cout);
            pack_wino4r_weights(wfold.data(), L.cout, cin, uws.data());
            if ((rc = upload(uws, &L.wino4s_dev))) return rc;
        }
        if (wino4) {
            std::vector<float> uw4((size_t)36 * L.cin_pad * L.cout_pad);
            pack_wino4_weights(wfold.data(), L.cout, cin, L.cin_pad, L.cout_pad, uw4.data(), L.in.w);
            if ((rc = upload(uw4, &L.wino4_dev))) return rc;
        }
        return 0;
    }

    int upload_key(const std::string& k, size_t numel, const float** d) {
        const HostTensor* t = find(k);
        if (!t) return fail(GRNET_ENOENT, "missing tensor " + k);
        if (t->numel() != numel) return fail(GRNET_EINVAL, "bad size for " + k);
        float* p = nullptr;
        int rc = upload(t->data, &p);
        *d = p;
        return rc;
    }

    // GRU weights are optional: loaded when every tensor is present under "gru." (standalone) or
    // "pfeat_corrector.featnet." (inside a MAX-GRNet checkpoint, feature_correction.py:44).
    int finalize_gru() {
        std::string pre;
        if (find("gru.rnn.weight_ih_l0")) pre = "gru.";
        else if (find("pfeat_corrector.featnet.rnn.weight_ih_l0")) pre = "pfeat_corrector.featnet.";
        else return 0;
        int rc;
        if ((rc = upload_key(pre + "cparam_mpl.weight", 128 * 3 * 24, &gruw.cparam_w))) return rc;
        for (int l = 0; l < 2; ++l)
            for (int d = 0; d < 2; ++d) {
                const std::string suf = "_l" + std::to_string(l) + (d ? "_reverse" : "");
                const size_t insz = l == 0 ? 3072 : 600;
                if ((rc = upload_key(pre + "rnn.weight_ih" + suf, 900 * insz, &gruw.w_ih[l][d]))) return rc;
                if ((rc = upload_key(pre + "rnn.bias_ih" + suf, 900, &gruw.b_ih[l][d]))) return rc;
                if ((rc = upload_key(pre + "rnn.bias_hh" + suf, 900, &gruw.b_hh[l][d]))) return rc;
                const HostTensor* whh = find(pre + "rnn.weight_hh" + suf);
                if (!whh || whh->numel() != 900 * 300) return fail(GRNET_ENOENT, "missing tensor " + pre + "rnn.weight_hh" + suf);
                std::vector<float> tr(900 * 300);
                for (int g = 0; g < 900; ++g)
                    for (int k = 0; k < 300; ++k) tr[(size_t)k * 900 + g] = whh->data[(size_t)g * 300 + k];
                float* p = nullptr;
                if ((rc = upload(tr, &p))) return rc;
                gruw.w_hh[l][d] = p;
            }
        struct { const char* name; const float** w0; const float** b0; const float** w2; const float** b2; int in, out; } heads[3] = {
            {"speed_mlp", &gruw.speed_w0, &gruw.speed_b0, &gruw.speed_w2, &gruw.speed_b2, 1200, 1},
            {"step_mlp", &gruw.step_w0, &gruw.step_b0, &gruw.step_w2, &gruw.step_b2, 1200, 2},
            {"phase_mlp", &gruw.phase_w0, &gruw.phase_b0, &gruw.phase_w2, &gruw.phase_b2, 600, 4}};
        for (auto& hd : heads) {
            const std::string q = pre + hd.name;
            if ((rc = upload_key(q + ".0.weight", (size_t)100 * hd.in, hd.w0))) return rc;
            if ((rc = upload_key(q + ".0.bias", 100, hd.b0))) return rc;
            if ((rc = upload_key(q + ".2.weight", (size_t)hd.out * 100, hd.w2))) return rc;
            if ((rc = upload_key(q + ".2.bias", hd.out, hd.b2))) return rc;
        }
        gru_ready = true;
        return 0;
    }

    // The attention block of the pose-feature corrector is optional as well: "tsattn." (standalone) or
    // "pfeat_corrector.featTencoder.0." (inside a MAX-GRNet checkpoint, feature_correction.py:95).
    int finalize_tsattn() {
        std::string pre;
        if (find("tsattn.mulattn.qkv_t.weight")) pre = "tsattn.";
        else if (find("pfeat_corrector.featTencoder.0.mulattn.qkv_t.weight")) pre = "pfeat_corrector.featTencoder.0.";
        else return 0;
        const size_t D = 3072, E = 1000;
        struct { const char* key; size_t n; const float** dst; } items[] = {
            {"norm1.gamma", D, &tsw.n1_g}, {"norm1.beta", D, &tsw.n1_b}, {"norm2.gamma", D, &tsw.n2_g}, {"norm2.beta", D, &tsw.n2_b},
            {"mulattn.qkv_t.weight", 3 * E * D, &tsw.qkv_t_w}, {"mulattn.qkv_t.bias", 3 * E, &tsw.qkv_t_b},
            {"mulattn.ts_attn.weight", 4 * E * E, &tsw.ts_w}, {"mulattn.ts_attn.bias", 2 * E, &tsw.ts_b},
            {"mulattn.qkv_s.weight", 3 * E * (D + 128), &tsw.qkv_s_w}, {"mulattn.qkv_s.bias", 3 * E, &tsw.qkv_s_b},
            {"mulattn.fc_s.weight", D * E, &tsw.fc_s_w}, {"mulattn.fc_s.bias", D, &tsw.fc_s_b},
            {"mulattn.fc_t.weight", D * E, &tsw.fc_t_w}, {"mulattn.fc_t.bias", D, &tsw.fc_t_b},
            {"ffn.jwff_layer1.weight", 64 * 128 * 24, &tsw.jw1}, {"ffn.jwff_layer2.weight", 128 * 64 * 24, &tsw.jw2}};
        for (auto& it : items) {
            int rc = upload_key(pre + it.key, it.n, it.dst);
            if (rc) return rc;
        }
        tsattn_ready = true;
        return 0;
    }

    // The rest of the pose-feature corrector (feature_correction.py:66-91): the two gait-token MLPs and the two input BatchNorm1d
    // (eval: folded to scale / shift in fp64).  Optional, under the keys of a MAX-GRNet checkpoint.
    int finalize_featcorr() {
        const std::string pre = "pfeat_corrector.";
        if (!find(pre + "gfeat_mpl_t.0.weight")) return 0;
        int rc;
        if ((rc = upload_key(pre + "gfeat_mpl_t.0.weight", 1536 * 7, &fcw.t0_w))) return rc;
        if ((rc = upload_key(pre + "gfeat_mpl_t.0.bias", 1536, &fcw.t0_b))) return rc;
        if ((rc = upload_key(pre + "gfeat_mpl_t.3.weight", (size_t)3072 * 1536, &fcw.t3_w))) return rc;
        if ((rc = upload_key(pre + "gfeat_mpl_t.3.bias", 3072, &fcw.t3_b))) return rc;
        if ((rc = upload_key(pre + "gfeat_mpl_s.0.weight", 64 * 7, &fcw.s0_w))) return rc;
        if ((rc = upload_key(pre + "gfeat_mpl_s.0.bias", 64, &fcw.s0_b))) return rc;
        if ((rc = upload_key(pre + "gfeat_mpl_s.3.weight", 128 * 64, &fcw.s3_w))) return rc;
        if ((rc = upload_key(pre + "gfeat_mpl_s.3.bias", 128, &fcw.s3_b))) return rc;
        struct { const char* name; size_t c; const float** scale; const float** shift; } bns[2] = {
            {"bn_in", 3072, &fcw.bn_scale, &fcw.bn_shift}, {"bn_in_s", 3200, &fcw.bns_scale, &fcw.bns_shift}};
        for (auto& bn : bns) {
            const HostTensor *g = find(pre + bn.name + ".weight"), *be = find(pre + bn.name + ".bias"),
                             *m = find(pre + bn.name + ".running_mean"), *v = find(pre + bn.name + ".running_var");
            if (!g || !be || !m || !v) return fail(GRNET_ENOENT, "missing BatchNorm1d tensors " + pre + bn.name + ".*");
            if (g->numel() != bn.c || be->numel() != bn.c || m->numel() != bn.c || v->numel() != bn.c)
                return fail(GRNET_EINVAL, "bad BatchNorm1d size " + pre + bn.name);
            std::vector<float> sc(bn.c), sh(bn.c);
            for (size_t c = 0; c < bn.c; ++c) {
                const double k = (double)g->data[c] / std::sqrt((double)v->data[c] + kBnEps);
                sc[c] = (float)k;
                sh[c] = (float)((double)be->data[c] - (double)m->data[c] * k);
            }
            float* p = nullptr;
            if ((rc = upload(sc, &p))) return rc;
            *bn.scale = p;
            if ((rc = upload(sh, &p))) return rc;
            *bn.shift = p;
        }
        featcorr_ready = true;
        return 0;
    }

    // The 1x1 fuse terms of one HR module (hrnet.py:199-210: Conv2d 1x1 + BatchNorm2d; the nearest upsampling commutes with both):
    // BatchNorm folded in fp64, weights in the MFMA B-fragment order of hr_fuse.hip, the shifts of an output's terms summed into one bias.
    int pack_fuse_up(FuseUpPlan& fp) {
        for (int i = 0; i < fp.nb - 1; ++i) {
            if (fp.only >= 0 && fp.only != i) continue;
            const int co = kBranchCh[i];
            std::vector<double> bias(co, 0.0);
            for (int j = i + 1; j < fp.nb; ++j) {
                const int ci = kBranchCh[j];
                const std::string q = fp.prefix + "fuse_layers." + std::to_string(i) + "." + std::to_string(j) + ".";
                const HostTensor* w = find(q + "0.weight");
                if (!w) return fail(GRNET_ENOENT, "missing tensor " + q + "0.weight");
                if (w->shape.size() != 4 || w->shape[0] != co || w->shape[1] != ci || w->shape[2] != 1 || w->shape[3] != 1) return fail(GRNET_EINVAL, "bad shape for " + q + "0.weight");
                const HostTensor *g = find(q + "1.weight"), *be = find(q + "1.bias"), *m = find(q + "1.running_mean"), *v = find(q + "1.running_var");
                if (!g || !be || !m || !v) return fail(GRNET_ENOENT, "missing BatchNorm tensors " + q + "1.*");
                if ((int)g->numel() != co || (int)be->numel() != co || (int)m->numel() != co || (int)v->numel() != co) return fail(GRNET_EINVAL, "bad BatchNorm size " + q + "1");
                std::vector<double> wf((size_t)co * ci);
                for (int c = 0; c < co; ++c) {
                    const double sc = (double)g->data[c] / std::sqrt((double)v->data[c] + kBnEps);
                    bias[c] += (double)be->data[c] - (double)m->data[c] * sc;
                    for (int k = 0; k < ci; ++k) wf[(size_t)c * ci + k] = (double)w->data[(size_t)c * ci + k] * sc;
                }
                std::vector<float> packed((size_t)co * ci / (dtype == 1 ? 2 : 1));
                if (dtype == 1) pack_fuse_up_weights_bf16(wf.data(), co, ci, reinterpret_cast<unsigned short*>(packed.data()));
                else pack_fuse_up_weights(wf.data(), co, ci, packed.data());
                if (int rc = upload(packed, &fp.w_dev[i][j - i - 1])) return rc;
            }
            std::vector<float> bf(bias.begin(), bias.end());
            if (int rc = upload(bf, &fp.b_dev[i])) return rc;
        }
        return 0;
    }

    int finalize() {
        if (finalized) return fail(GRNET_ESTATE, "weights already finalized");
        for (auto& L : convs) {
            int rc = pack_conv(L);
            if (rc) return rc;
        }
        for (auto& fp : fuse_ups) {
            int rc = pack_fuse_up(fp);
            if (rc) return rc;
        }
        int rc;
        {   // per-joint 128 -> 6 weights (locallyconnected2d.py:43-46), stored (6,128,24) = [o][c][j]; the tail kernel walks c with one
            // thread per (j, o): re-order to [c][j][o] so every step reads 144 contiguous floats instead of 144 lines
            const HostTensor* t = find("head.pose_mlp.weight");
            if (!t) return fail(GRNET_ENOENT, "missing tensor head.pose_mlp.weight");
            if (t->numel() != 6 * 128 * 24) return fail(GRNET_EINVAL, "bad size for head.pose_mlp.weight");
            std::vector<float> tr(6 * 128 * 24);
            for (int o = 0; o < 6; ++o)
                for (int c = 0; c < 128; ++c)
                    for (int j = 0; j < 24; ++j) tr[(size_t)c * 144 + j * 6 + o] = t->data[((size_t)o * 128 + c) * 24 + j];
            float* p = nullptr;
            if ((rc = upload(tr, &p))) return rc;
            tailw.pose_w = p;
        }
        if ((rc = upload_key("head.shape_mlp.weight", 10 * 1536, &tailw.shape_w))) return rc;
        if ((rc = upload_key("head.shape_mlp.bias", 10, &tailw.shape_b))) return rc;
        if ((rc = upload_key("head.cam_mlp.weight", 3 * 1536, &tailw.cam_w))) return rc;
        if ((rc = upload_key("head.cam_mlp.bias", 3, &tailw.cam_b))) return rc;
        if ((rc = finalize_gru())) return rc;
        if ((rc = finalize_tsattn())) return rc;
        if ((rc = finalize_featcorr())) return rc;
        if (!smpl_loaded) return fail(GRNET_ESTATE, "grnet_load_smpl must be called before grnet_finalize_weights");
        tensors.clear();                                    // host copies no longer needed
        finalized = true;
        return 0;
    }

    // ------------------------------------------------------------------ tuning
    int hint_for(const ConvLayer& L, int n) const {
        if (conv_tile_hint) return conv_tile_hint;
        auto m = tuned_mode.find(n);
        if (m == tuned_mode.end() || !(m->second & 1)) return 0;      // cost model
        auto it = L.tuned.find(n);
        return it == L.tuned.end() ? 0 : it->second;
    }

    // Measure, don't guess: time every launch configuration of every distinct convolution shape on this GPU
    // for n frames (3 launches each, HIP events) and keep the fastest; then time whole forwards as a replayed hipGraph and as eager
    // launches on the lane streams, with the cost model's and the measured table, and keep the fastest.  Activation buffers are used as
    // scratch (contents are garbage afterwards, like after any forward).
    // conv_wino4s_f32 on this layer in a call of n frames?  A 7x7 row tile is four images: below three row tiles (n < 12) a launch is
    // 16-32 workgroups whose waves each walk 16 k-steps, and the direct split-K kernel with its 8-wave workgroups is the shorter chain
    // link (measured at 1 / 2 / 4 / 8 / 12 frames: -4 % / -3 % / -5 % / -4 % / +0.5 % with the 7x7 layers on it; 14x14: +2 ... +4 % throughout)
    bool wino4s_runs(const ConvLayer& L, int n) const { return L.wino4s_dev && wino_mode && (L.in.w != 7 || n >= 12); }
    // layer1's 64 -> 256 1x1 convolutions and the PARE head's 128 -> 25 heat-map layer on 56x56 maps: the register-resident kernel of
    // conv_pw.hip (fp32 handles; GRNET_PW: bit 0 64 -> 256, bit 1 128 -> 25, bit 2 the rest of the eligible shapes -- 64 -> 64 and
    // 128 -> 64 measure within 1 us of the generic kernel either way and stay on it; 0: the generic kernel everywhere)
    bool pw_on(const ConvLayer& L) const {
        static const int pw_env = GRNET_AB(PW, 3);
        return dtype == 0 && L.in.w == 56 && L.segs.size() == 1 && L.cin_w == L.in.c && (L.adds.empty() || L.adds[0].shift == 0) &&
               conv_pw_eligible(L.in.c, L.cout, L.ks, L.stride, L.in.h, L.in.w, (int)L.adds.size()) && L.cout_pad >= (L.cout > 32 ? (L.cout + 63) / 64 * 64 : 32) &&
               (pw_env & (L.in.c == 64 && L.cout >= 128 ? 1 : L.in.c == 128 && L.cout <= 32 ? 2 : 4));
    }
    int last_n = 16;                   // frame count of the latest forward (grnet_conv_executed_flops_per_frame reports for it)
    std::map<int, int> tuned_mode;     // n -> bit 0: measured per-shape configurations (else cost model), bit 2: eager launches on the lane streams even if graphs are enabled
    int tune(int n, hipStream_t s, int level = 1) {
        if (!finalized) return fail(GRNET_ESTATE, "grnet_tune before grnet_finalize_weights");
        if (n < 1 || n > max_frames) return fail(GRNET_EINVAL, "n_frames outside [1, max_frames]");
        static const int cands[] = {0, 14, 7, 1071, 1072, 1041, 1042, 1171, 1141};
        hipEvent_t e0 = nullptr, e1 = nullptr;
        // whatever way this function is left: events destroyed, half-built graphs dropped, the caller's schedule switches restored,
        // and -- unless the tuning completed -- no partial entry for n left behind
        struct Restore {
            grnet* g; int n; bool use_graph, done = false; hipEvent_t *e0, *e1;
            ~Restore() {
                if (*e0) (void)hipEventDestroy(*e0);
                if (*e1) (void)hipEventDestroy(*e1);
                g->drop_graphs();
                g->use_graph = use_graph;
                if (!done) { g->tuned_mode.erase(n); for (auto& L : g->convs) L.tuned.erase(n); }
            }
        } restore{this, n, use_graph, false, &e0, &e1};
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        std::map<std::tuple<int, int, int, int, int, int, int>, int> by_shape;
        for (auto& L : convs) {
            if (dtype == 1) { L.tuned[n] = 0; continue; }          // the bf16 kernel picks its tile by map width; only the schedule is timed
            const auto key = std::make_tuple(L.in.c, L.cout, L.ks, L.stride, L.in.h, (int)L.adds.size() + (L.solo ? 100 : 0), L.out.ctot);
            auto it = by_shape.find(key);
            if (it != by_shape.end()) { L.tuned[n] = it->second; continue; }
            float best = 1e30f, t_model = 1e30f;
            int best_hint = 0;
            for (int hint : cands) {
                ConvArgs a = conv_args(L, v_cat.p, n);        // any readable buffer stands in for the caller's frames
                if (launch_conv(a, s, hint) != hipSuccess) { (void)hipGetLastError(); continue; }
                HIP_TRY(hipEventRecord(e0, s));
                for (int r = 0; r < 3; ++r) (void)launch_conv(a, s, hint);
                HIP_TRY(hipEventRecord(e1, s));
                HIP_TRY(hipEventSynchronize(e1));
                float ms = 0;
                HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
                if (hint == 0) t_model = ms;
                if (ms < best) { best = ms; best_hint = hint; }
            }
            // keep the cost model's choice unless a measured configuration is clearly (1.3x) faster in isolation:
            // close calls measured alone do not predict behaviour when several lanes share the CUs
            // (layers that run alone -- stem, layer1, PARE head -- take any measured gain above noise)
            if (!(t_model > (L.solo ? 1.06f : 1.3f) * best)) best_hint = 0;
            L.tuned[n] = best_hint;
            by_shape[key] = best_hint;
        }
        // schedule: {cost model, measured table} x {replayed hipGraph, eager launches on the four lane streams} -- the graph executor of
        // ROCm 7.2 maps parallel branches to fewer hardware queues than explicit streams do, so eager multi-stream launching can win
        // although it costs CPU time per launch.  Mode bits: 1 = measured per-shape table, 4 = eager.
        float t_mode[8];
        for (float& t : t_mode) t = 1e30f;
        const bool keep_graph = use_graph;
        for (int mode : {0, 1, 4, 5}) {
            if ((mode & 4) == 0 && !keep_graph) continue;                               // graphs not enabled by the caller
            if (dtype == 1 && (mode & 1)) continue;                                     // bf16: no per-shape table
            use_graph = (mode & 4) == 0;
            tuned_mode[n] = mode;
            drop_graphs();
            seen_once.clear();
            int rc = forward(v_cat.p, n, nullptr, s);          // first sight of the key: eager
            if (!rc) rc = forward(v_cat.p, n, nullptr, s);     // second: builds the graph, first replay
            if (rc) return rc;
            HIP_TRY(hipEventRecord(e0, s));
            for (int r = 0; r < 3; ++r) if ((rc = forward(v_cat.p, n, nullptr, s))) return rc;
            HIP_TRY(hipEventRecord(e1, s));
            HIP_TRY(hipEventSynchronize(e1));
            HIP_TRY(hipEventElapsedTime(&t_mode[mode], e0, e1));
        }
        use_graph = keep_graph;
        int best_mode = -1;
        for (int mode : {0, 1, 4, 5})
            if (t_mode[mode] < 1e30f && (best_mode < 0 || t_mode[mode] < t_mode[best_mode])) best_mode = mode;
        // three forwards per mode are a noisy clock (+-3 % from run to run on a shared node): a replayed graph has to win by more than that over the eager
        // launches of the same table to be taken (it never has: ROCm 7.2's executor deals the branches to fewer queues than the four lane streams)
        if (best_mode >= 0 && !(best_mode & 4) && t_mode[best_mode | 4] < 1e30f && t_mode[best_mode] > 0.97f * t_mode[best_mode | 4]) best_mode |= 4;
        if (best_mode < 0) return fail(GRNET_ESTATE, "no schedule could be timed");
        tuned_mode[n] = best_mode;
        // in-context refinement (level 2): greedy coordinate descent on the time of the whole replayed forward --
        // a configuration that wins alone can lose when four lanes share the CUs.  Shapes in order of their FLOP share.
        if (level >= 2) {
            use_graph = (best_mode & 4) == 0;
            tuned_mode[n] = best_mode | 1;                      // refine the measured table under the winning schedule
            auto time_forward = [&](float* out_ms) -> int {
                drop_graphs();
                seen_once.clear();
                int rc = forward(v_cat.p, n, nullptr, s);
                if (!rc) rc = forward(v_cat.p, n, nullptr, s);
                if (rc) return rc;
                float best_ms = 1e30f;
                for (int rep2 = 0; rep2 < 2; ++rep2) {
                    HIP_TRY(hipEventRecord(e0, s));
                    for (int r = 0; r < 2; ++r) if ((rc = forward(v_cat.p, n, nullptr, s))) return rc;
                    HIP_TRY(hipEventRecord(e1, s));
                    HIP_TRY(hipEventSynchronize(e1));
                    float ms = 0;
                    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best_ms) best_ms = ms;
                }
                *out_ms = best_ms / 2;
                return 0;
            };
            typedef std::tuple<int, int, int, int, int, int, int> Key;
            std::map<Key, double> share;
            auto key_of = [](const ConvLayer& L) { return std::make_tuple(L.in.c, L.cout, L.ks, L.stride, L.in.h, (int)L.adds.size(), L.out.ctot); };
            for (auto& L : convs) share[key_of(L)] += L.macs_per_frame;
            std::vector<std::pair<double, Key>> order;
            for (auto& kv : share) order.push_back({kv.second, kv.first});
            std::sort(order.begin(), order.end(), [](const std::pair<double, Key>& x, const std::pair<double, Key>& y) { return x.first > y.first; });
            float cur_ms = 0;
            int rc = time_forward(&cur_ms);
            if (rc) return rc;
            const float start_ms = cur_ms;
            for (auto& ok : order) {
                int keep_hint = 0;
                for (auto& L : convs) if (key_of(L) == ok.second) { keep_hint = L.tuned[n]; break; }
                int best_hint = keep_hint;
                for (int hint : cands) {
                    if (hint == keep_hint) continue;
                    bool valid = true;
                    for (auto& L : convs)
                        if (key_of(L) == ok.second) {
                            ConvArgs a = conv_args(L, v_cat.p, n);
                            if (launch_conv(a, s, hint) != hipSuccess) { (void)hipGetLastError(); valid = false; }
                            break;
                        }
                    if (!valid) continue;
                    for (auto& L : convs) if (key_of(L) == ok.second) L.tuned[n] = hint;
                    float ms = 0;
                    if ((rc = time_forward(&ms))) return rc;
                    if (ms < cur_ms * 0.995f) { cur_ms = ms; best_hint = hint; }
                }
                for (auto& L : convs) if (key_of(L) == ok.second) L.tuned[n] = best_hint;
            }
            if (getenv("GRNET_TRACE")) fprintf(stderr, "[grnet] in-context tuning n=%d: %.3f -> %.3f ms\n", n, start_ms, cur_ms);
            use_graph = keep_graph;
        }
        restore.done = true;
        seen_once.clear();
        if (getenv("GRNET_TRACE"))
            fprintf(stderr, "[grnet] tuned n=%d: forward ms graph[model %.3f measured %.3f] eager[model %.3f measured %.3f] -> mode %d\n",
                    n, t_mode[0] / 3, t_mode[1] / 3, t_mode[4] / 3, t_mode[5] / 3, best_mode);
        return 0;
    }
    // ------------------------------------------------------------------ execution
    static const void* bf16_at(const View& v) { return reinterpret_cast<const uint16_t*>(v.p) + v.coff; }   // first channel of an NHWC bf16 view
    ConvArgs conv_args(const ConvLayer& L, const float* frames, int n) const {
        ConvArgs a{};
        a.in = reinterpret_cast<uintptr_t>(L.in.p) == ~(uintptr_t)0 ? frames : L.in.p;
        a.in_ctot = L.in.ctot; a.in_coff = L.in.coff;
        a.N = n; a.Cin = L.in.c; a.H = L.in.h; a.W = L.in.w;
        a.out = L.out.p; a.out_ctot = L.out.ctot; a.out_coff = L.out.coff;
        a.Cout = L.cout; a.Ho = L.out.h; a.Wo = L.out.w;
        a.w = L.w_dev; a.bias = L.b_dev; a.CinPad = L.cin_pad; a.CoutPad = L.cout_pad;
        a.ks = L.ks; a.stride = L.stride; a.relu = L.relu; a.relu_from = L.relu_from;
        a.n_add = (int)L.adds.size();
        for (int k = 0; k < a.n_add; ++k) {
            a.add[k] = L.adds[k].v.p; a.add_ctot[k] = L.adds[k].v.ctot; a.add_coff[k] = L.adds[k].v.coff;
            a.add_shift[k] = L.adds[k].shift;
        }
        a.zeros = zeros;
        a.pw_stream = !(chain_mode & 128) ? 0 : bf16_min_frames ? 2 : 1;
        if (L.in2.c) { a.in2 = L.in2.p; a.in2_ctot = L.in2.ctot; a.in2_coff = L.in2.coff; a.cin_split = L.in.c; a.Cin = L.in.c + L.in2.c; }
        if (L.pair_next >= 0 && pair_active(n)) {
            const ConvLayer& F = convs[L.pair_next];
            a.w2 = F.w_dev; a.bias2 = F.b_dev; a.out2 = F.out.p; a.out2_ctot = F.out.ctot; a.out2_coff = F.out.coff; a.relu2 = F.relu;
        }
        return a;
    }
    // bf16 layer1: expansion + next reduction as one launch from 19 frames per call on (the 256-channel tile needs >= 512 workgroups of 112 pixels); bit 6 of the
    // GRNET_OPT_BF16_CHAIN mask.  A forced tile switches it off.
    bool pair_active(int n) const { return dtype == 1 && (chain_mode & 64) && !conv_tile_hint && (long)n * 56 * 56 >= 256L * 112 * 2; }      // (geometric: GRNET_OPT_BF16_MIN_FRAMES does not lower it)
    int bf16_min_frames = 0;                         // GRNET_OPT_BF16_MIN_FRAMES: 0 = every kernel group of chain_mode from its own smallest call (64 / 32 / 64 / 19 / 42 frames), else from this many

    // Which kernel runs convolution L in a call of n frames: ONE place, used by the launcher, by the executed-FLOP report and by the
    // per-kernel table of bench.py (round-3 review: the report read a hidden "latest n" and ignored the environment masks).
    enum ConvKernel { K_BF16, K_BF16_STEM, K_BF16_ROLL, K_BF16_ROLL_MEMBER, K_BF16_CHAIN, K_BF16_CHAIN_MEMBER, K_BF16_PAIR, K_BF16_PAIR_MEMBER, K_BF16_WIDE, K_BF16_S2, K_WINO4S, K_PW, K_STEM, K_WINO4, K_DIRECT };
    // bf16: does chain `c` run as ONE conv_bf16_chain launch in a call of n frames?  A chain workgroup is one frame on one CU: from about a
    // quarter of the chip's CUs on it beats eight launches (GRNET_BF16_CHAIN: bit 0 64 ch @28x28, bit 1 128 ch @14x14, bit 2 256 ch @7x7, bit 3 32 ch @56x56 --
    // there a launch per BasicBlock with 19-row bands resident;
    // GRNET_BF16_CHAIN_MIN: smallest call that takes it).  A forced tile (tests / tuning) switches it off like every special kernel.
    bool chain_active(const ChainPlan& c, int n) const {
        const int chain_min = bf16_min_frames ? bf16_min_frames : 64;
        return dtype == 1 && !conv_tile_hint && n >= chain_min && (chain_mode & (c.w == 28 ? 1 : c.w == 14 ? 2 : c.w == 7 ? 4 : 8));
    }
    // bf16: the wide 3x3 stride-1 layers (upsample heads, PARE head, layer1's 3x3) on conv_bf16_wide_band.  A workgroup is a band of 7 / 14 rows of one
    // frame x 128 (64) output channels: from 32 frames per call on a launch has at least one workgroup per CU (bit 4 of the GRNET_OPT_BF16_CHAIN mask;
    // GRNET_BF16_WIDE_MIN: smallest call).  A forced tile switches it off like every special kernel.
    bool wide_runs(const ConvLayer& L, int n) const {
        const int wide_min = bf16_min_frames ? bf16_min_frames : 32;
        if (dtype != 1 || !(chain_mode & 16) || conv_tile_hint || n < wide_min || L.stem_dev || !L.w_dev) return false;
        return conv_bf16_wide_eligible(conv_args(L, nullptr, n));
    }
    // bf16: the 3x3 stride-2 layers (fuse-layer down paths, transitions, the stem's second convolution) on conv_bf16_s2_band, from 64 frames per call on
    // (a workgroup is a band of one frame; GRNET_BF16_S2_MIN).  Bit 5 of the GRNET_OPT_BF16_CHAIN mask.
    bool s2_runs(const ConvLayer& L, int n) const {
        const int s2_min = bf16_min_frames ? bf16_min_frames : 64;
        if (dtype != 1 || !(chain_mode & 32) || conv_tile_hint || n < s2_min || L.stem_dev || !L.w_dev || L.in2.c) return false;
        return conv_bf16_s2_eligible(conv_args(L, nullptr, n));
    }
    static constexpr int kChainModeAll = 1023;
    // bf16: the stem pair (bit 9 of the mask) / a layer1 Bottleneck (bit 8) as ONE row-walking launch (conv_bf16_roll.hip), from 64 frames per call on (a workgroup is a
    // frame, or a quarter of one): HBM sees the launch's input and output once.  A forced tile switches it off like every special kernel.
    bool roll_active(const RollPlan& r, int n) const {
        return dtype == 1 && !conv_tile_hint && n >= (bf16_min_frames ? bf16_min_frames : 64) && (chain_mode & (r.kind == 0 ? 512 : 256));
    }
    int chain_mode = (getenv("GRNET_BF16_CHAIN") ? atoi(getenv("GRNET_BF16_CHAIN")) : kChainModeAll) & kChainModeAll;     // GRNET_OPT_BF16_CHAIN: bits 0-3 BasicBlock chains by branch, 4 wide bands, 5 stride-2 bands, 6 layer1 1x1 pairs, 7 1x1 stream kernel
    ConvKernel kernel_for(const ConvLayer& L, int n) const {
        static const int w4s_env = GRNET_AB(WINO4S, 7);      // bit 0: 128 @14x14, bit 1: 256 @7x7, bit 2: 256 @14x14
        if (dtype == 1 && L.roll >= 0 && roll_active(rolls[L.roll], n)) return L.roll_pos == 0 ? K_BF16_ROLL : K_BF16_ROLL_MEMBER;
        if (dtype == 1 && L.chain >= 0 && chain_active(chains[L.chain], n)) return L.chain_pos == 0 ? K_BF16_CHAIN : K_BF16_CHAIN_MEMBER;
        if (dtype == 1 && L.pair_next >= 0 && pair_active(n)) return K_BF16_PAIR;
        if (dtype == 1 && L.pair_of >= 0 && pair_active(n)) return K_BF16_PAIR_MEMBER;
        if (dtype == 1 && wide_runs(L, n)) return K_BF16_WIDE;
        if (dtype == 1 && s2_runs(L, n)) return K_BF16_S2;
        if (dtype == 1) return L.stem_dev ? K_BF16_STEM : K_BF16;      // (a plan built for conv_bf16_stem has no NHWC copy of the frames for the generic kernel)
        if (conv_tile_hint) return K_DIRECT;                   // a forced tile also switches every special kernel off (tests / tuning)
        if (wino4s_runs(L, n) && (w4s_env & (L.in.w == 7 ? 2 : L.in.c == 128 ? 1 : 4))) return K_WINO4S;
        if (pw_on(L)) return K_PW;
        if (L.stem_dev) return K_STEM;
        if (L.wino4_dev && wino_mode) return K_WINO4;
        return K_DIRECT;
    }
    // multiplies the matrix cores execute per algorithmic multiply of L: F(4x4,3x3) does 36 per 4x4 tile instead of 144; the small maps pay
    // for their padding (14 -> 16, 7 -> 8 per side)
    double executed_ratio(const ConvLayer& L, int n) const {
        switch (kernel_for(L, n)) {
            case K_WINO4S: return 0.25 * (L.in.w == 14 ? 256.0 / 196.0 : 64.0 / 49.0);
            case K_WINO4: return 0.25;
            default: return 1.0;
        }
    }
    std::string kernel_name(const ConvLayer& L, int n) const {
        char b[96];
        switch (kernel_for(L, n)) {
            case K_BF16: return "conv_bf16";
            case K_BF16_STEM: return "conv_bf16_stem";
            case K_BF16_ROLL: return rolls[L.roll].kind == 0 ? "conv_bf16_stem_pair" : "conv_bf16_bneck";
            case K_BF16_ROLL_MEMBER: return rolls[L.roll].kind == 0 ? "conv_bf16_stem_pair+" : "conv_bf16_bneck+";      // runs inside the launch of the group's first member
            case K_BF16_PAIR: return "conv_bf16_pair";
            case K_BF16_PAIR_MEMBER: return "conv_bf16_pair+";     // runs inside the pair's launch
            case K_BF16_WIDE: snprintf(b, sizeof b, "conv_bf16_wide<%d,%d>", L.in.c >= 128 ? 128 : 64, L.in.w); return b;
            case K_BF16_S2: snprintf(b, sizeof b, "conv_bf16_s2<%d>", L.out.w); return b;
            case K_BF16_CHAIN: snprintf(b, sizeof b, "conv_bf16_chain<%d,%d>", L.in.c, L.in.w); return b;
            case K_BF16_CHAIN_MEMBER: snprintf(b, sizeof b, "conv_bf16_chain<%d,%d>+", L.in.c, L.in.w); return b;      // runs inside the chain's launch: no launch, no time of its own
            case K_WINO4S: snprintf(b, sizeof b, "conv_wino4s_f32<%d,%d>", L.in.w, L.in.c); return b;
            case K_PW: snprintf(b, sizeof b, "conv_pw_f32<%d>", L.in.c); return b;
            case K_STEM: return "conv_stem_f32";
            case K_WINO4: {
                const int npw = conv_wino4_wide(L.cout, L.in.w);
                if (npw && L.cin_pad % 16 == 0 && L.cout_pad % (npw * 32) == 0) snprintf(b, sizeof b, "conv_wino4w_f32<%d,%d>", L.in.w, npw);
                else snprintf(b, sizeof b, "conv_wino4_f32<%d,%d>", conv_wino4_blocks(L.cout, L.in.w), L.in.w);
                return b;
            }
            default: snprintf(b, sizeof b, "conv_direct_f32 %dx%d s%d", L.ks, L.ks, L.stride); return b;
        }
    }
    int launch_conv_op(const ConvLayer& L, const float* frames, int n, hipStream_t s, int* n_launches) {
        static const int w4s_ks = GRNET_AB(WINO4S_KS, 0);
        *n_launches = 1;
        switch (kernel_for(L, n)) {
            case K_BF16: HIP_TRY(launch_conv_bf16(conv_args(L, frames, n), s, hint_for(L, n))); break;
            case K_BF16_STEM: HIP_TRY(launch_conv_bf16_stem(frames, L.stem_dev, L.b_dev, L.out.p, L.out.ctot, L.out.coff, n, L.relu, s)); break;
            case K_BF16_CHAIN: {
                const ChainPlan& cp = chains[L.chain];
                const ConvLayer& last = convs[cp.convs.back()];
                ChainArgs ca{};
                ca.in = L.in.p; ca.in_ctot = L.in.ctot; ca.in_coff = L.in.coff;
                ca.out = last.out.p; ca.out_ctot = last.out.ctot; ca.out_coff = last.out.coff;
                ca.N = n; ca.nconv = (int)cp.convs.size();
                for (int i = 0; i < ca.nconv; ++i) { ca.w[i] = convs[cp.convs[i]].w_dev; ca.bias[i] = convs[cp.convs[i]].b_dev; }
                for (int k = 0; k + 1 < ca.nconv / 2; ++k) {          // the 56x56 branch runs one launch per BasicBlock: the blocks' own output buffers carry the hand-over
                    const View& m = convs[cp.convs[2 * k + 1]].out;
                    ca.mid[k] = m.p; ca.mid_ctot[k] = m.ctot; ca.mid_coff[k] = m.coff;
                }
                HIP_TRY(launch_conv_bf16_chain(ca, cp.c, cp.w, s));
                *n_launches = conv_bf16_chain_launches(cp.c, cp.w, ca.nconv);
                break;
            }
            case K_BF16_CHAIN_MEMBER: *n_launches = 0; break;       // its work is in the launch of the chain's first member
            case K_BF16_ROLL: {
                const RollPlan& rp = rolls[L.roll];
                const ConvLayer& last = convs[rp.convs.back()];
                if (rp.kind == 0) {
                    const ConvLayer& c2 = convs[rp.convs[1]];
                    HIP_TRY(launch_conv_bf16_stem_pair(frames, last.out.p, last.out.ctot, last.out.coff, n, L.stem_dev, L.b_dev, c2.w_dev, c2.b_dev, s));
                } else {
                    const ConvLayer &c2 = convs[rp.convs[1]], &c3 = convs[rp.convs[2]];
                    HIP_TRY(launch_conv_bf16_bneck(L.in.p, L.in.ctot, L.in.coff, last.out.p, last.out.ctot, last.out.coff, n, rp.kind == 1, L.w_dev, L.b_dev, c2.w_dev, c2.b_dev,
                                                   c3.w_dev, c3.b_dev, s));
                }
                break;
            }
            case K_BF16_ROLL_MEMBER: *n_launches = 0; break;        // its work is in the launch of the group's first member
            case K_BF16_PAIR: HIP_TRY(launch_conv_bf16(conv_args(L, frames, n), s, 0)); break;
            case K_BF16_PAIR_MEMBER: *n_launches = 0; break;        // its work is the second stage of the expansion's launch
            case K_BF16_WIDE: HIP_TRY(launch_conv_bf16_wide(conv_args(L, frames, n), s)); break;
            case K_BF16_S2: HIP_TRY(launch_conv_bf16_s2(conv_args(L, frames, n), s)); break;
            case K_WINO4S: {
                ConvArgs wa = conv_args(L, frames, n);
                wa.w = L.wino4s_dev;
                static const int w4s_prio = GRNET_AB(WINO4S_PRIO, 3);   // bit 0: 14x14 layers, bit 1: 7x7 layers at wave priority 1
                wa.prio = (w4s_prio & (L.in.w == 7 ? 2 : 1)) ? 1 : 0;
                HIP_TRY(launch_conv_wino4s(wa, s, w4s_ks));
                break;
            }
            case K_PW: HIP_TRY(launch_conv_pw(conv_args(L, frames, n), s)); break;
            case K_STEM: {
                ConvArgs wa = conv_args(L, frames, n);
                wa.w = L.stem_dev;
                HIP_TRY(launch_conv_stem(wa, s));
                break;
            }
            case K_WINO4: {
                ConvArgs wa = conv_args(L, frames, n);
                wa.w = L.wino4_dev;
                static const int chain_prio4 = GRNET_AB(WINO_PRIO, 1);
                // the BasicBlock chains of the 56x56 and 28x28 HR branches (32-channel workgroups): wave priority 1.  Worth +1 % when
                // only the 56x56 chain ran on a Winograd kernel; with both on F(4x4,3x3) every combination is within 0.5 %
                wa.prio = (L.in.c == L.cout && L.cout <= 64 && !L.solo) ? chain_prio4 : 0;
                HIP_TRY(launch_conv_wino4(wa, s, n_launches));
                break;
            }
            case K_DIRECT: HIP_TRY(launch_conv(conv_args(L, frames, n), s, hint_for(L, n))); break;
        }
        return 0;
    }
    int launch_fuse_up_op(const FuseUpPlan& fp, int n, hipStream_t s) {
        FuseUpArgs a{};
        a.N = n; a.nb = fp.nb; a.only = fp.only;
        for (int i = 0; i < fp.nb - 1; ++i) {
            if (fp.only >= 0 && fp.only != i) continue;
            FuseUpOut& fo = a.o[i];
            fo.out = fp.outs[i].p; fo.out_ctot = fp.outs[i].ctot; fo.out_coff = fp.outs[i].coff;
            fo.base = fp.xs[i].p; fo.base_ctot = fp.xs[i].ctot; fo.base_coff = fp.xs[i].coff;
            fo.bias = fp.b_dev[i];
            fo.relu = 1;
            fo.n_extra = (int)fp.extra[i].size();
            for (int k = 0; k < fo.n_extra; ++k) { fo.extra[k] = fp.extra[i][k].p; fo.extra_ctot[k] = fp.extra[i][k].ctot; fo.extra_coff[k] = fp.extra[i][k].coff; }
            for (int j = i + 1; j < fp.nb; ++j) fo.src[j - i - 1] = FuseUpSrc{fp.xs[j].p, fp.xs[j].ctot, fp.xs[j].coff, fp.w_dev[i][j - i - 1]};
        }
        HIP_TRY(dtype == 1 ? launch_hr_fuse_up_bf16(a, s) : launch_hr_fuse_up(a, s));
        return 0;
    }

    int enqueue(const float* frames, int n, const grnet_outputs_t& o, hipStream_t s, bool convs_only = false) {
        int launches = 0;
        last_n = n;
        float* plf = o.point_local_feat ? o.point_local_feat : d_plf;
        float* csf = o.cam_shape_feats ? o.cam_shape_feats : d_csf;
        float* rot6d = o.pred_rot6d ? o.pred_rot6d : d_rot6d;
        float* rotmat = o.rotmat ? o.rotmat : d_rotmat;
        float* theta = o.theta ? o.theta : d_theta;
        float* verts = o.verts ? o.verts : d_verts;
        float* kp3d = o.kp_3d ? o.kp_3d : d_kp3d;
        float* kp2d = o.kp_2d ? o.kp_2d : d_kp2d;
        const std::vector<Op>& ops = ops_flat;
        const std::vector<hipEvent_t>& op_events = op_events_flat;
        GraphRecorder* rec = g_recorder;                          // non-null: build graph nodes instead of launching
        const bool lanes = multi_lane && !rec;
        std::vector<hipGraphNode_t> lane_last(kLanes, nullptr), op_node(rec ? ops.size() : 0, nullptr);
        hipStream_t lane_stream[kLanes];
        for (int l = 0; l < kLanes; ++l) lane_stream[l] = s;
        if (lanes) {
            HIP_TRY(hipEventRecord(ev_fork, s));                 // fork: side lanes start after everything before this forward
            for (int l = 1; l < lanes_used; ++l) {
                lane_stream[l] = side[l];
                HIP_TRY(hipStreamWaitEvent(side[l], ev_fork, 0));
            }
        }
        hipStream_t caller = s;
        for (size_t oi = 0; oi < ops.size(); ++oi) {
            const Op& op = ops[oi];
            if (convs_only && op.kind != Op::CONV && op.kind != Op::FUSEUP) continue;
            s = lane_stream[op.lane];
            const int lane = multi_lane ? op.lane : 0;
            if (lanes)
                for (int w : op.waits) HIP_TRY(hipStreamWaitEvent(s, op_events[w], 0));
            if (tl_start && !rec) HIP_TRY(hipEventRecord((*tl_start)[oi], s));
            if (rec) {                                            // dependencies: previous node of the lane + cross-lane producers
                rec->deps.clear();
                rec->n_chain = lane_last[lane] ? 1 : 0;
                if (lane_last[lane]) rec->deps.push_back(lane_last[lane]);
                if (multi_lane)
                    for (int w : op.waits)                          // several waited ops can be ONE node (the members of a chain launch): an edge is added once
                        if (op_node[w] && std::find(rec->deps.begin(), rec->deps.end(), op_node[w]) == rec->deps.end()) rec->deps.push_back(op_node[w]);
            }
            // timing-only ablation (results are garbage): GRNET_ABL_SKIP=<substring of a weight key>[,<substring>...] drops the matching
            // convolution launches and "fuse_up" the grouped fuse launches, events and dependencies stay -- what is a group of launches worth?
            static const char* abl_skip = GRNET_AB_STR(ABL_SKIP);          // diagnostic builds only (make ABLATION=1): a stray variable must not make the product drop launches
            if (abl_skip && (op.kind == Op::CONV || op.kind == Op::FUSEUP)) {
                const std::string lbl = op_label(op);
                bool skip = false;
                for (const char* q = abl_skip; *q;) {
                    const char* e = strchr(q, ',');
                    const std::string pat = e ? std::string(q, e) : std::string(q);
                    if (!pat.empty() && lbl.find(pat) != std::string::npos) skip = true;
                    q = e ? e + 1 : q + strlen(q);
                }
                if (skip) {
                    if (tl_end && !rec) HIP_TRY(hipEventRecord((*tl_end)[oi], s));
                    if (lanes && op.record) HIP_TRY(hipEventRecord(op_events[oi], s));
                    continue;
                }
            }
            switch (op.kind) {
                case Op::CONVERT:
                    HIP_TRY(launch_nchw_f32_to_nhwc_bf16(frames, v_in8.p, n, 3, 224, 224, 8, s));
                    ++launches;
                    break;
                case Op::CONV: {
                    int nl = 1;
                    if (int rc = launch_conv_op(convs[op.conv_idx], frames, n, s, &nl)) return rc;
                    launches += nl;
                    break;
                }
                case Op::SUM: {
                    const auto& sv = sum_views[op.conv_idx];
                    SumArgs a = op.sum;
                    a.N = n;
                    a.out = sv.first.p; a.out_ctot = sv.first.ctot; a.out_coff = sv.first.coff;
                    for (int k = 0; k < a.n_add; ++k) {
                        a.add[k] = sv.second[k].v.p; a.add_ctot[k] = sv.second[k].v.ctot; a.add_coff[k] = sv.second[k].v.coff;
                        a.add_shift[k] = sv.second[k].shift;
                    }
                    if (dtype == 1) HIP_TRY(launch_fuse_sum_bf16(a, s));
                    else HIP_TRY(launch_fuse_sum(a, s));
                    ++launches;
                    break;
                }
                case Op::FUSEUP:
                    if (int rc = launch_fuse_up_op(fuse_ups[op.conv_idx], n, s)) return rc;
                    ++launches;
                    break;
                case Op::BILINEAR:
                    if (dtype == 1) HIP_TRY(launch_bilinear2x_bf16(op.bin.p, op.bout.p, n, op.bin.c, op.bin.h, op.bin.w, s));
                    else HIP_TRY(launch_bilinear2x(op.bin.p, op.bout.p, n, op.bin.c, op.bin.h, op.bin.w, s));
                    ++launches;
                    break;
                case Op::POOL:
                    if (dtype == 1)
                        HIP_TRY(launch_softmax_pool_bf16(v_heat.p, v_heat.ctot, bf16_at(v_smpl_feats), 128, v_smpl_feats.ctot, bf16_at(v_csmap), 64,
                                                         v_csmap.ctot, d_stats, n, 56 * 56, s));
                    else
                        HIP_TRY(launch_softmax_pool(v_heat.p, 25, v_smpl_feats.p, 128, v_csmap.p, 64, plf, csf, d_stats, n, 56 * 56, s));
                    ++launches;
                    break;
                case Op::TAIL:
                    HIP_TRY(launch_head_tail(d_stats, true, plf, csf, tailw, rot6d, d_shape, d_cam, rotmat, theta, n, s));
                    ++launches;
                    break;
                case Op::SMPL:
                    HIP_TRY(launch_smpl(d_shape, rotmat, d_cam, smpl, d_A, verts, kp3d, kp2d, n, s));
                    launches += 4;
                    break;
            }
            if (tl_end && !rec) HIP_TRY(hipEventRecord((*tl_end)[oi], s));
            if (lanes && op.record) HIP_TRY(hipEventRecord(op_events[oi], s));
            if (rec && !rec->deps.empty()) lane_last[lane] = op_node[oi] = rec->deps[0];
        }
        s = caller;
        if (lanes)
            for (int l = 1; l < lanes_used; ++l) {                // join: the caller's stream continues after every lane
                HIP_TRY(hipEventRecord(ev_join[l], side[l]));
                HIP_TRY(hipStreamWaitEvent(s, ev_join[l], 0));
            }
        if (!convs_only) {
            auto copy_out = [&](float* dst, const float* src, size_t bytes) -> int {
                if (!dst) return 0;
                if (!rec) { HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s)); return 0; }
                std::vector<hipGraphNode_t> deps;
                for (hipGraphNode_t nd : lane_last) if (nd) deps.push_back(nd);
                hipGraphNode_t node = nullptr;
                HIP_TRY(hipGraphAddMemcpyNode1D(&node, rec->graph, deps.data(), deps.size(), dst, src, bytes, hipMemcpyDeviceToDevice));
                return 0;
            };
            int rc;
            if (dtype == 1) {                                   // the optional map outputs stay (N,C,56,56) fp32 for the caller
                auto conv_out = [&](float* dst, const View& v) -> int {
                    if (!dst) return 0;
                    if (rec) {
                        rec->deps.clear();
                        for (hipGraphNode_t nd : lane_last) if (nd) rec->deps.push_back(nd);
                    }
                    HIP_TRY(launch_nhwc_bf16_to_nchw_f32(v.p, dst, n, v.c, v.h, v.w, v.ctot, v.coff, s));
                    return 0;
                };
                if ((rc = conv_out(o.features, v_cat))) return rc;
                if ((rc = conv_out(o.part_attn, v_heat))) return rc;
                if ((rc = conv_out(o.smpl_feats, v_smpl_feats))) return rc;
            } else {
            if ((rc = copy_out(o.features, v_cat.p, (size_t)n * 480 * 3136 * 4))) return rc;
            if ((rc = copy_out(o.part_attn, v_heat.p, (size_t)n * 25 * 3136 * 4))) return rc;
            if ((rc = copy_out(o.smpl_feats, v_smpl_feats.p, (size_t)n * 128 * 3136 * 4))) return rc;
            }
            launches_last = launches;
        }
        return 0;
    }

    // Test hook on a bf16 handle: (n,cin,h,w) f32 NCHW in / out, converted to and from NHWC bf16 around ONE conv launch.
    int op_conv2d_bf16(const float* in_dev, int n, int cin, int hgt, int wid, const float* w_host, const float* bias_host, int cout, int ks,
                       int stride, int relu, const float* add_dev, float* out_dev, int tile_hint, hipStream_t s) {
        const int taps = ks * ks, pad = ks / 2, cin8 = (cin + 7) / 8 * 8, cin_pad = (cin + 31) / 32 * 32, cout_pad = (cout + 31) / 32 * 32;
        const int ho = (hgt + 2 * pad - ks) / stride + 1, wo = (wid + 2 * pad - ks) / stride + 1, cout8 = (cout + 7) / 8 * 8;
        std::vector<uint16_t> wp((size_t)taps * cout_pad * cin_pad, 0);
        std::vector<float> bp(cout_pad, 0.f);
        for (int co = 0; co < cout; ++co) {
            if (bias_host) bp[co] = bias_host[co];
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < taps; ++t)
                    wp[(((size_t)(ci / 32) * taps + t) * cout_pad + co) * 32 + ci % 32] = f32_to_bf16(w_host[((size_t)co * cin + ci) * taps + t]);
        }
        if (tile_hint == 3001) {                               // conv_bf16_stem on this one convolution: fp32 NCHW in, (n,cout,112,112) f32 out
            if (!conv_stem_eligible(cin, cout, ks, stride, hgt, wid, add_dev ? 1 : 0)) return fail(GRNET_EINVAL, "shape not eligible for the bf16 stem kernel");
            std::vector<double> wf((size_t)cout * cin * 9);
            for (size_t i = 0; i < wf.size(); ++i) wf[i] = w_host[i];
            std::vector<unsigned short> sw(4 * 64 * 8);
            pack_stem_weights_bf16(wf.data(), sw.data());
            std::vector<float> bh(64, 0.f);
            if (bias_host) for (int c = 0; c < 64; ++c) bh[c] = bias_host[c];
            void *swd = nullptr, *bhd = nullptr, *od = nullptr;
            if (hipMalloc(&swd, sw.size() * 2) != hipSuccess || hipMalloc(&bhd, 256) != hipSuccess || hipMalloc(&od, (size_t)n * ho * wo * 64 * 2) != hipSuccess) return fail(GRNET_ENOMEM, "hipMalloc failed");
            hipError_t e = hipMemcpy(swd, sw.data(), sw.size() * 2, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(bhd, bh.data(), 256, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = launch_conv_bf16_stem(in_dev, swd, static_cast<const float*>(bhd), od, 64, 0, n, relu, s);
            if (e == hipSuccess) e = launch_nhwc_bf16_to_nchw_f32(od, out_dev, n, 64, ho, wo, 64, 0, s);
            hipError_t e2 = hipStreamSynchronize(s);
            hipFree(swd); hipFree(bhd); hipFree(od);
            if (e != hipSuccess || e2 != hipSuccess) return fail(GRNET_EHIP, std::string("bf16 stem conv: ") + hipGetErrorString(e != hipSuccess ? e : e2));
            return 0;
        }
        void *wd = nullptr, *bd = nullptr, *xin = nullptr, *xadd = nullptr, *xout = nullptr;
        const size_t in_b = (size_t)n * hgt * wid * cin8 * 2, out_b = (size_t)n * ho * wo * cout8 * 2;
        if (hipMalloc(&wd, wp.size() * 2) != hipSuccess || hipMalloc(&bd, bp.size() * 4) != hipSuccess || hipMalloc(&xin, in_b) != hipSuccess ||
            hipMalloc(&xout, out_b) != hipSuccess || (add_dev && hipMalloc(&xadd, out_b) != hipSuccess))
            return fail(GRNET_ENOMEM, "hipMalloc failed");
        if (hipMemcpy(wd, wp.data(), wp.size() * 2, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(bd, bp.data(), bp.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            hipFree(wd); hipFree(bd); hipFree(xin); hipFree(xout);
            if (xadd) hipFree(xadd);
            return fail(GRNET_EHIP, "hipMemcpy of the test weights failed");
        }
        hipError_t e = launch_nchw_f32_to_nhwc_bf16(in_dev, xin, n, cin, hgt, wid, cin8, s);
        if (e == hipSuccess && add_dev) e = launch_nchw_f32_to_nhwc_bf16(add_dev, xadd, n, cout, ho, wo, cout8, s);
        ConvArgs a{};
        a.in = static_cast<const float*>(xin); a.in_ctot = cin8; a.in_coff = 0; a.N = n; a.Cin = cin8; a.H = hgt; a.W = wid;
        a.Cout = cout; a.Ho = ho; a.Wo = wo;
        a.out = static_cast<float*>(xout); a.out_ctot = cout8; a.out_coff = 0;
        a.w = static_cast<const float*>(wd); a.bias = static_cast<const float*>(bd); a.CinPad = cin_pad; a.CoutPad = cout_pad;
        a.ks = ks; a.stride = stride; a.relu = relu;
        if (add_dev) { a.n_add = 1; a.add[0] = static_cast<const float*>(xadd); a.add_ctot[0] = cout8; a.add_coff[0] = 0; a.add_shift[0] = 0; }
        a.zeros = zeros;
        a.pw_stream = 1;
        if (const char* d = GRNET_AB_STR(CONV_DBG)) a.dbg = atoi(d);
        const bool wide = tile_hint == 3003, s2 = tile_hint == 3004;      // conv_bf16_wide_band / conv_bf16_s2_band on this one convolution
        if ((wide && !conv_bf16_wide_eligible(a)) || (s2 && !conv_bf16_s2_eligible(a))) {
            hipFree(wd); hipFree(bd); hipFree(xin); hipFree(xout);
            if (xadd) hipFree(xadd);
            return fail(GRNET_EINVAL, "shape not eligible for the band kernel");
        }
        auto launch_one = [&]() { return wide ? launch_conv_bf16_wide(a, s) : s2 ? launch_conv_bf16_s2(a, s) : launch_conv_bf16(a, s, tile_hint); };
        if (e == hipSuccess) e = launch_one();
        if (const char* r = GRNET_AB_STR(CONV_REPS)) {       // timing loop for tools/bf16_micro.py
            const int reps = atoi(r);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, s);
            for (int i = 0; i < reps && e == hipSuccess; ++i) e = launch_one();
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double mb = (in_b + out_b * (add_dev ? 2 : 1)) / 1e6;
            fprintf(stderr, "[bf16_micro] cin %d cout %d k %d s %d hw %d n %d hint %d add %d: %.2f us/launch, %.0f MB algorithmic = %.2f TB/s\n", cin, cout, ks, stride,
                    hgt, n, tile_hint, add_dev ? 1 : 0, ms * 1e3f / reps, mb, mb / (ms * 1e3 / reps));
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        if (e == hipSuccess) e = launch_nhwc_bf16_to_nchw_f32(xout, out_dev, n, cout, ho, wo, cout8, 0, s);
        hipError_t e2 = hipStreamSynchronize(s);
        hipFree(wd); hipFree(bd); hipFree(xin); hipFree(xout);
        if (xadd) hipFree(xadd);
        if (e != hipSuccess) return fail(GRNET_EHIP, std::string("bf16 conv: ") + hipGetErrorString(e));
        if (e2 != hipSuccess) return fail(GRNET_EHIP, std::string("bf16 conv kernel: ") + hipGetErrorString(e2));
        return 0;
    }

    // Test / timing hook on a bf16 handle: a chain of nconv 3x3 convolutions c -> c on (n,c,w,w) f32 NCHW in / out (converted to and from NHWC
    // bf16 around ONE conv_bf16_chain launch).  w_host: nconv x (c,c,3,3), bias_host: nconv x (c).  reps > 0: also times `reps` back-to-back
    // launches with HIP events (*us_out: us per launch).
    int op_conv_chain_bf16(const float* in_dev, int n, int c, int wid, int nconv, const float* w_host, const float* bias_host, float* out_dev, int reps,
                           float* us_out, hipStream_t s) {
        if (dtype != 1) return fail(GRNET_ESTATE, "grnet_op_conv_chain needs a bf16 handle");
        if (!conv_bf16_chain_eligible(c, wid) || nconv < 2 || nconv > kMaxChain || (nconv & 1) || n < 1) return fail(GRNET_EINVAL, "shape not eligible for the chain kernel");
        const size_t wel = (size_t)9 * c * c;
        std::vector<uint16_t> wp(wel * nconv, 0);
        for (int i = 0; i < nconv; ++i)
            for (int co = 0; co < c; ++co)
                for (int ci = 0; ci < c; ++ci)
                    for (int t = 0; t < 9; ++t)
                        wp[i * wel + (((size_t)(ci / 32) * 9 + t) * c + co) * 32 + ci % 32] = f32_to_bf16(w_host[i * wel + ((size_t)co * c + ci) * 9 + t]);
        void *wd = nullptr, *bd = nullptr, *xin = nullptr, *xout = nullptr;
        const size_t act_b = (size_t)n * wid * wid * c * 2;
        auto cleanup = [&]() { if (wd) hipFree(wd); if (bd) hipFree(bd); if (xin) hipFree(xin); if (xout) hipFree(xout); };
        if (hipMalloc(&wd, wp.size() * 2) != hipSuccess || hipMalloc(&bd, (size_t)nconv * c * 4) != hipSuccess || hipMalloc(&xin, act_b) != hipSuccess ||
            hipMalloc(&xout, act_b) != hipSuccess) { cleanup(); return fail(GRNET_ENOMEM, "hipMalloc failed"); }
        hipError_t e = hipMemcpy(wd, wp.data(), wp.size() * 2, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(bd, bias_host, (size_t)nconv * c * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = launch_nchw_f32_to_nhwc_bf16(in_dev, xin, n, c, wid, wid, c, s);
        ChainArgs ca{};
        ca.in = xin; ca.in_ctot = c; ca.in_coff = 0; ca.out = xout; ca.out_ctot = c; ca.out_coff = 0; ca.N = n; ca.nconv = nconv;
        void* mids[kMaxChain / 2 - 1] = {};
        if (conv_bf16_chain_launches(c, wid, nconv) > 1)
            for (int k = 0; k + 1 < nconv / 2; ++k) {
                if (hipMalloc(&mids[k], act_b) != hipSuccess) { for (void* m : mids) if (m) hipFree(m); cleanup(); return fail(GRNET_ENOMEM, "hipMalloc failed"); }
                ca.mid[k] = mids[k]; ca.mid_ctot[k] = c; ca.mid_coff[k] = 0;
            }
        for (int i = 0; i < nconv; ++i) { ca.w[i] = static_cast<const uint16_t*>(wd) + i * wel; ca.bias[i] = static_cast<const float*>(bd) + (size_t)i * c; }
        if (e == hipSuccess) e = launch_conv_bf16_chain(ca, c, wid, s);
        if (e == hipSuccess && reps > 0 && us_out) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, s);
            for (int i = 0; i < reps && e == hipSuccess; ++i) e = launch_conv_bf16_chain(ca, c, wid, s);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            *us_out = ms * 1e3f / reps;
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
        if (e == hipSuccess) e = launch_nhwc_bf16_to_nchw_f32(xout, out_dev, n, c, wid, wid, c, 0, s);
        hipError_t e2 = hipStreamSynchronize(s);
        cleanup();
        for (void* m : mids) if (m) hipFree(m);
        if (e != hipSuccess) return fail(GRNET_EHIP, std::string("bf16 chain: ") + hipGetErrorString(e));
        if (e2 != hipSuccess) return fail(GRNET_EHIP, std::string("bf16 chain kernel: ") + hipGetErrorString(e2));
        return 0;
    }

    // tail + SMPL from given pooled features; outputs as in enqueue() (NULL -> internal buffer)
    int head_from_feats(const float* plf, const float* csf, int n, const grnet_outputs_t& o, hipStream_t s) {
        float* rot6d = o.pred_rot6d ? o.pred_rot6d : d_rot6d;
        float* rotmat = o.rotmat ? o.rotmat : d_rotmat;
        float* theta = o.theta ? o.theta : d_theta;
        float* verts = o.verts ? o.verts : d_verts;
        float* kp3d = o.kp_3d ? o.kp_3d : d_kp3d;
        float* kp2d = o.kp_2d ? o.kp_2d : d_kp2d;
        HIP_TRY(launch_head_tail_from_feats(plf, csf, tailw, rot6d, d_shape, d_cam, rotmat, theta, n, s));
        HIP_TRY(launch_smpl(d_shape, rotmat, d_cam, smpl, d_A, verts, kp3d, kp2d, n, s));
        if (o.point_local_feat && o.point_local_feat != plf) HIP_TRY(hipMemcpyAsync(o.point_local_feat, plf, (size_t)n * 3072 * 4, hipMemcpyDeviceToDevice, s));
        if (o.cam_shape_feats && o.cam_shape_feats != csf) HIP_TRY(hipMemcpyAsync(o.cam_shape_feats, csf, (size_t)n * 1536 * 4, hipMemcpyDeviceToDevice, s));
        return 0;
    }

    // The use_gait_feat branch of GRNet.forward after the first head pass (grnet.py:154-173): cparams, FeatCorrector, second head pass,
    // regressor.  plf (b*T,128,24), csf (b*T,64,24), cam (b*T rows of stride cam_ld: pred_cam, or theta with cam_ld = 85) are the first
    // pass's results for the WHOLE clip(s); the second head pass runs in chunks of max_frames.
    int gait_correct(const float* plf, const float* csf, const float* cam, int cam_ld, const float* bbox, const float* cimg, int b, int T,
                     const grnet_outputs_t& o, const grnet_gait_outputs_t& g, hipStream_t s) {
        if (int rc = gru_fault_check()) return rc;
        const size_t M = (size_t)b * T;
        const size_t gru_need = M * 3072 * 2 + 2 * M * 900 + 2 * M * 600 + (size_t)b * 1200 + (size_t)b * 2 * kGruXbufU64PerSeq + 1024;
        auto al = [](size_t f) { return (f + 63) & ~(size_t)63; };         // every sub-buffer starts 256-byte aligned (16-byte vector loads, 8-byte granules)
        const size_t own = al(M * 3) + al((size_t)b * 3) + al(M * 4) + al(M * 3072);
        float* ws = nullptr;
        if (int rc = temporal_scratch(kGemmWsFloats + gru_need + featcorr_ws_floats(b, T) + own, &ws)) return rc;
        GemmWorkspaceLease lease(ws, kGemmWsFloats);          // handed back on EVERY way out of this function
        float* p = ws + kGemmWsFloats;
        float* cparams = g.pred_cparam ? g.pred_cparam : p;   p += al(M * 3);
        float* avg = g.pred_avg ? g.pred_avg : p;             p += al((size_t)b * 3);
        float* phase = g.pred_phase ? g.pred_phase : p;       p += al(M * 4);
        float* new_plf = g.point_local_feat ? g.point_local_feat : p;   p += al(M * 3072);
        GruWorkspace w;
        w.xin = p;
        float* xc_buf = p + M * 3072;
        w.gi = p + M * 3072 * 2;
        w.l0 = w.gi + 2 * M * 900;
        w.l1 = w.l0 + M * 600;
        w.hfin = w.l1 + M * 600;
        w.xbuf = reinterpret_cast<unsigned long long*>(w.hfin + (((size_t)b * 1200 + 63) & ~(size_t)63));
        w.mode = gru_mode; w.fault = gru_fault_dev;
        float* fws = p + gru_need;
        HIP_TRY(launch_gait_cparams(cam, cam_ld, bbox, cimg, cparams, (int)M, s));
        HIP_TRY(launch_gru(plf, cparams, gruw, w, avg, phase, xc_buf, b, T, s));
        HIP_TRY(launch_featcorr(plf, avg, phase, fcw, tsw, fws, new_plf, b, T, s));
        for (size_t s0 = 0; s0 < M; s0 += (size_t)max_frames) {
            const int m = (int)std::min<size_t>((size_t)max_frames, M - s0);
            grnet_outputs_t oc{};
            oc.theta = o.theta ? o.theta + s0 * 85 : nullptr;
            oc.verts = o.verts ? o.verts + s0 * 6890 * 3 : nullptr;
            oc.kp_2d = o.kp_2d ? o.kp_2d + s0 * 58 : nullptr;
            oc.kp_3d = o.kp_3d ? o.kp_3d + s0 * 87 : nullptr;
            oc.rotmat = o.rotmat ? o.rotmat + s0 * 216 : nullptr;
            oc.pred_rot6d = o.pred_rot6d ? o.pred_rot6d + s0 * 144 : nullptr;
            if (int rc = head_from_feats(new_plf + s0 * 3072, csf + s0 * 1536, m, oc, s)) return rc;
        }
        return 0;
    }

    std::string op_label(const Op& op) const {
        switch (op.kind) {
            case Op::CONV: {
                const ConvLayer& L = convs[op.conv_idx];
                char b[256];
                snprintf(b, sizeof b, "conv %dx%d s%d %d->%d @%d %s", L.ks, L.ks, L.stride, L.in.c, L.cout, L.in.w, L.segs.empty() ? "" : L.segs[0].wkey.c_str());
                return b;
            }
            case Op::FUSEUP: return "fuse_up " + fuse_ups[op.conv_idx].prefix;
            case Op::SUM: return "fuse_sum";
            case Op::BILINEAR: return "bilinear2x c" + std::to_string(op.bin.c) + " @" + std::to_string(op.bin.w);
            case Op::POOL: return "attn_pool";
            case Op::TAIL: return "head_tail";
            case Op::SMPL: return "smpl";
            case Op::CONVERT: return "convert";
        }
        return "?";
    }

    // Diagnostic: one eager forward on the lane streams with a timing event in front of and behind every op (after its cross-lane waits),
    // un-traced -- rocprofv3's per-dispatch cost distorts a step of ~300 launches of 5-25 us.  Text: one line per op in enqueue order,
    // "index lane start_us end_us label", times relative to the first op's start.  The events cost ~1 us of queue time each.
    int op_timeline(const float* frames, int n, hipStream_t s, std::string& text) {
        if (!finalized) return fail(GRNET_ESTATE, "grnet_op_timeline before grnet_finalize_weights");
        if (!frames || n < 1 || n > max_frames) return fail(GRNET_EINVAL, "n_frames outside [1, max_frames]");
        if (!multi_lane) return fail(GRNET_ESTATE, "grnet_op_timeline needs GRNET_OPT_MULTI_LANE");
        const size_t m = ops_flat.size();
        std::vector<hipEvent_t> st(m, nullptr), en(m, nullptr);
        struct Cleanup {
            grnet* g; std::vector<hipEvent_t>*a, *b;
            ~Cleanup() { g->tl_start = g->tl_end = nullptr; for (auto e : *a) if (e) (void)hipEventDestroy(e); for (auto e : *b) if (e) (void)hipEventDestroy(e); }
        } cleanup{this, &st, &en};
        for (size_t i = 0; i < m; ++i) { HIP_TRY(hipEventCreate(&st[i])); HIP_TRY(hipEventCreate(&en[i])); }
        grnet_outputs_t o{};
        for (int rep = 0; rep < 3; ++rep) {                      // two warm passes, the third is reported
            tl_start = rep == 2 ? &st : nullptr;
            tl_end = rep == 2 ? &en : nullptr;
            int rc = enqueue(frames, n, o, s);
            tl_start = tl_end = nullptr;
            if (rc) return rc;
        }
        HIP_TRY(hipStreamSynchronize(s));
        text.clear();
        for (size_t i = 0; i < m; ++i) {
            float a = 0, b = 0;
            HIP_TRY(hipEventElapsedTime(&a, st[0], st[i]));
            HIP_TRY(hipEventElapsedTime(&b, st[0], en[i]));
            char line[400];
            snprintf(line, sizeof line, "%zu %d %.2f %.2f %s\n", i, ops_flat[i].lane, a * 1e3f, b * 1e3f, op_label(ops_flat[i]).c_str());
            text += line;
        }
        return 0;
    }

    int forward(const float* frames, int n, const grnet_outputs_t* out, hipStream_t s) {
        if (!finalized) return fail(GRNET_ESTATE, "grnet_forward before grnet_finalize_weights");
        if (!frames || n < 1 || n > max_frames)
            return fail(GRNET_EINVAL, "n_frames " + std::to_string(n) + " outside [1, max_frames=" + std::to_string(max_frames) + "]");
        grnet_outputs_t o{};
        if (out) o = *out;
        last_n = n;
        {
            auto tm = tuned_mode.find(n);
            const bool eager_tuned = tm != tuned_mode.end() && (tm->second & 4);
            if (!use_graph || eager_tuned) return enqueue(frames, n, o, s);
        }
        GraphKey key{n, frames, o};
        auto it = graphs.find(key);
        // A caller that passes fresh output buffers every call (the Python shim does) rarely repeats a key, so a key is only
        // captured the SECOND time it is seen (first sight: eager launch, remembered in `seen_once`), and the cache keeps the
        // kMaxGraphs most recently used captured forwards: a steady-state key always ends up captured, one-off keys cost nothing.
        constexpr size_t kMaxGraphs = 16;
        if (it == graphs.end()) {
            bool seen = false;
            for (const GraphKey& k : seen_once) seen |= !(k < key) && !(key < k);
            if (!seen) {
                if (seen_once.size() >= 64) seen_once.erase(seen_once.begin());
                seen_once.push_back(key);
                return enqueue(frames, n, o, s);
            }
            if (graphs.size() >= kMaxGraphs) {
                auto lru = graphs.begin();
                for (auto g = graphs.begin(); g != graphs.end(); ++g)
                    if (g->second.last_use < lru->second.last_use) lru = g;
                (void)hipGraphExecDestroy(lru->second.exec);
                graphs.erase(lru);
            }
        }
        if (it == graphs.end()) {
            hipGraph_t g = nullptr;
            HIP_TRY(hipGraphCreate(&g, 0));
            GraphRecorder recorder;
            recorder.graph = g;
            // GRNET_GRAPH_EDGES (diagnostic): 0 = dependencies given at node creation (edges in plan order; default), 1 = lane-chain edges first, 2 = cross-lane
            // edges first.  The order changes how ROCm 7.2's executor deals the nodes over its queues (108 / 26 / 142 / 16, 105 / 20 / 159 / 8, 116 / 90 / 70 / 16)
            // but none of them replays faster than 4.16 ms against 3.5 ms for the eager lane streams (profiles/r04_graph_vs_eager_timeline.txt)
            static const int edges_env = GRNET_AB(GRAPH_EDGES, 0);
            recorder.edge_order = edges_env;
            g_recorder = &recorder;
            int rc = enqueue(frames, n, o, s);
            g_recorder = nullptr;
            if (rc) { hipGraphDestroy(g); return rc; }
            hipError_t e = hipSuccess;
            if (recorder.edge_order) {
                auto add = [&](std::vector<hipGraphNode_t>& from, std::vector<hipGraphNode_t>& to) {
                    if (e == hipSuccess && !from.empty()) e = hipGraphAddDependencies(g, from.data(), to.data(), from.size());
                };
                if (recorder.edge_order == 2) { add(recorder.cross_from, recorder.cross_to); add(recorder.chain_from, recorder.chain_to); }
                else { add(recorder.chain_from, recorder.chain_to); add(recorder.cross_from, recorder.cross_to); }
                if (e != hipSuccess) { hipGraphDestroy(g); return fail(GRNET_EHIP, std::string("hipGraphAddDependencies: ") + hipGetErrorString(e)); }
            }
            hipGraphExec_t ge = nullptr;
            e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            hipGraphDestroy(g);
            if (e != hipSuccess) return fail(GRNET_EHIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
            it = graphs.emplace(key, GraphEntry{ge, 0}).first;
        }
        it->second.last_use = ++graph_clock;
        HIP_TRY(hipGraphLaunch(it->second.exec, s));
        return 0;
    }
};

// ================================================================================ C ABI
extern "C" {

const char* grnet_version(void) { return "grnet_hip 0.1 (gfx950, fp32 MFMA)"; }

int grnet_create(grnet_t** out_handle, int device_id, int dtype, int max_frames) {
    if (!out_handle || max_frames < 1 || max_frames > 2048 || (dtype != 0 && dtype != 1)) return GRNET_EINVAL;
    *out_handle = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return GRNET_EHIP;
    DeviceGuard guard(device_id);                          // the caller's current device is restored on return
    std::unique_ptr<grnet> h(new grnet());
    h->device = device_id;
    h->max_frames = max_frames;
    h->dtype = dtype;
    if (const char* ml = getenv("GRNET_MULTI_LANE")) h->multi_lane = atoi(ml) != 0;   // profiling: per-kernel times without overlap
    if (const char* wn = getenv("GRNET_WINO")) h->wino_mode = atoi(wn) != 0;
    h->build_plan();
    int rc = h->allocate();
    if (rc) { fprintf(stderr, "grnet_create: %s\n", h->err.c_str()); return rc; }
    if (conv_init() != hipSuccess || conv_bf16_init() != hipSuccess || conv_bf16_chain_init() != hipSuccess || conv_bf16_roll_init() != hipSuccess) { fprintf(stderr, "grnet_create: conv_init failed\n"); return GRNET_EHIP; }
    *out_handle = h.release();
    return 0;
}

int grnet_load_tensor(grnet_t* h, const char* key, const void* host_ptr, const int64_t* shape, int ndim, int dtype) {
    if (!h || !key || (!host_ptr && ndim >= 0 && dtype == GRNET_DTYPE_F32) || ndim < 0 || ndim > 8) return GRNET_EINVAL;
    if (h->finalized) return h->fail(GRNET_ESTATE, "grnet_load_tensor after grnet_finalize_weights");
    if (dtype == GRNET_DTYPE_I64) return 0;                 // num_batches_tracked: irrelevant in eval
    if (dtype != GRNET_DTYPE_F32) return h->fail(GRNET_EINVAL, std::string("unsupported dtype for ") + key);
    HostTensor t;
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); numel *= (size_t)shape[i]; }
    t.data.assign(static_cast<const float*>(host_ptr), static_cast<const float*>(host_ptr) + numel);
    h->tensors[key] = std::move(t);
    return 0;
}

int grnet_load_smpl(grnet_t* h, const float* v_template, const float* shapedirs, const float* posedirs, const float* J_regressor,
                    const float* lbs_weights, const int32_t* parents, const float* J_regressor_extra) {
    if (!h || !v_template || !shapedirs || !posedirs || !J_regressor || !lbs_weights || !parents || !J_regressor_extra)
        return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    const int V = 6890;
    for (int i = 0; i < 24; ++i)
        if (parents[i] >= i || (i > 0 && parents[i] < 0)) return h->fail(GRNET_EINVAL, "SMPL parents must be topologically ordered");
    auto up = [&](const float* src, size_t n, const float** dst) {
        std::vector<float> tmp(src, src + n);
        float* p = nullptr;
        int rc = h->upload(tmp, &p);
        *dst = p;
        return rc;
    };
    int rc;
    {   // blend-shape table of the MFMA GEMM: [posedirs (207 rows) ; shapedirs^T (10) ; v_template (1) ; 0 0], row-major (220, 20670)
        const size_t C = (size_t)V * 3;
        std::vector<float> blend((size_t)kBlendK * C, 0.f);
        memcpy(blend.data(), posedirs, (size_t)207 * C * sizeof(float));
        for (size_t c = 0; c < C; ++c) {
            for (int l = 0; l < 10; ++l) blend[(size_t)(207 + l) * C + c] = shapedirs[c * 10 + l];
            blend[(size_t)217 * C + c] = v_template[c];
        }
        float* p = nullptr;
        if ((rc = h->upload(blend, &p))) return rc;
        h->smpl.blend = p;
    }
    {   // skinning weights as a padded (joint, weight) list per vertex: non-zero entries in ascending joint order
        int kmax = 1;
        for (int v = 0; v < V; ++v) {
            int c = 0;
            for (int j = 0; j < 24; ++j) c += lbs_weights[(size_t)v * 24 + j] != 0.f;
            kmax = std::max(kmax, c);
        }
        std::vector<float> w((size_t)V * kmax, 0.f), idx_f((size_t)V * kmax);
        int32_t* idx = reinterpret_cast<int32_t*>(idx_f.data());
        for (int v = 0; v < V; ++v) {
            int c = 0;
            for (int j = 0; j < 24; ++j) {
                const float wj = lbs_weights[(size_t)v * 24 + j];
                if (wj != 0.f) { idx[(size_t)v * kmax + c] = j; w[(size_t)v * kmax + c] = wj; ++c; }
            }
            for (; c < kmax; ++c) idx[(size_t)v * kmax + c] = -1;
        }
        float* p = nullptr;
        if ((rc = h->upload(w, &p))) return rc;
        h->smpl.skin_w = p;
        if ((rc = h->upload(idx_f, &p))) return rc;          // int32 payload moved as raw 4-byte words
        h->smpl.skin_idx = reinterpret_cast<const int*>(p);
        h->smpl.skin_k = kmax;
    }
    if ((rc = up(lbs_weights, (size_t)V * 24, &h->smpl.lbs_weights))) return rc;
    {   // the one extra joint the path uses (smpl.py:117: JOINT_MAP 'Thorax (MPII)' = 50 -> row 5): sparse row
        std::vector<float> w, idx_f;
        for (int v = 0; v < V; ++v) {
            const float x = J_regressor_extra[(size_t)5 * V + v];
            if (x != 0.f) { w.push_back(x); int32_t i = v; float f; memcpy(&f, &i, 4); idx_f.push_back(f); }
        }
        h->smpl.thorax_n = (int)w.size();
        if (w.empty()) { w.push_back(0.f); idx_f.push_back(0.f); }
        float* p = nullptr;
        if ((rc = h->upload(w, &p))) return rc;
        h->smpl.thorax_w = p;
        if ((rc = h->upload(idx_f, &p))) return rc;
        h->smpl.thorax_idx = reinterpret_cast<const int*>(p);
    }
    // the joint regressor is linear: apply it to the tables once, in fp64
    std::vector<float> Jt(72), Js(720);
    for (int j = 0; j < 24; ++j)
        for (int d = 0; d < 3; ++d) {
            double a = 0;
            double s[10] = {0};
            for (int v = 0; v < V; ++v) {
                const double w = J_regressor[(size_t)j * V + v];
                if (w == 0.0) continue;
                a += w * v_template[v * 3 + d];
                for (int l = 0; l < 10; ++l) s[l] += w * shapedirs[((size_t)v * 3 + d) * 10 + l];
            }
            Jt[j * 3 + d] = (float)a;
            for (int l = 0; l < 10; ++l) Js[(j * 3 + d) * 10 + l] = (float)s[l];
        }
    float* p = nullptr;
    if ((rc = h->upload(Jt, &p))) return rc;
    h->smpl.J_template = p;
    if ((rc = h->upload(Js, &p))) return rc;
    h->smpl.J_shapedirs = p;
    void* q = nullptr;
    if (hipMalloc(&q, 24 * sizeof(int)) != hipSuccess) return h->fail(GRNET_ENOMEM, "hipMalloc failed");
    h->dev_allocs.push_back(q);
    if (hipMemcpy(q, parents, 24 * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return h->fail(GRNET_EHIP, "hipMemcpy failed");
    h->smpl.parents = static_cast<const int*>(q);
    h->smpl_loaded = true;
    return 0;
}

int grnet_finalize_weights(grnet_t* h) {
    if (!h) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    return h->finalize();
}

int grnet_forward(grnet_t* h, const float* frames_dev, int n_frames, const grnet_outputs_t* out, void* stream) {
    if (!h) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    return h->forward(frames_dev, n_frames, out, static_cast<hipStream_t>(stream));
}

int grnet_gru_forward(grnet_t* h, const float* x, const float* cp, int b, int T, float* y, float* phase, float* xc, void* stream) {
    if (!h || !x || !cp || !y || !phase || b < 1 || T < 1) return GRNET_EINVAL;
    if (!h->gru_ready) return h->fail(GRNET_ESTATE, "GRU weights were not loaded (keys gru.* or pfeat_corrector.featnet.*)");
    DeviceGuard guard(h->device);
    if (int rc = h->gru_fault_check()) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t rows = (size_t)b * T;
    float* ws = nullptr;                                   // handle-owned scratch: no allocation once a size has been seen
    const size_t need = rows * 3072 * 2 + 2 * rows * 900 + 2 * rows * 600 + (size_t)b * 1200 + (size_t)b * 2 * kGruXbufU64PerSeq + 1024;
    if (int rc = h->temporal_scratch(kGemmWsFloats + need, &ws)) return rc;
    GemmWorkspaceLease lease(ws, kGemmWsFloats);
    ws += kGemmWsFloats;
    GruWorkspace w;
    w.xin = ws;
    float* xc_buf = xc ? xc : ws + rows * 3072;
    w.gi = ws + rows * 3072 * 2;
    w.l0 = w.gi + 2 * rows * 900;
    w.l1 = w.l0 + rows * 600;
    w.hfin = w.l1 + rows * 600;
    w.xbuf = reinterpret_cast<unsigned long long*>(w.hfin + (((size_t)b * 1200 + 63) & ~(size_t)63));
    w.mode = h->gru_mode; w.fault = h->gru_fault_dev;
    hipError_t e = launch_gru(x, cp, h->gruw, w, y, phase, xc_buf, b, T, s);
    if (e != hipSuccess) return h->fail(GRNET_EHIP, std::string("launch_gru: ") + hipGetErrorString(e));
    return 0;
}

int grnet_tsattn_forward(grnet_t* h, const float* x, const float* xs, int b, int n, float* y, void* stream) {
    if (!h || !x || !xs || !y || b < 1 || n < 1) return GRNET_EINVAL;
    DeviceGuard guard(h->device);                          // the limit below is the handle's device's
    if (n > tsattn_max_frames()) return h->fail(GRNET_EINVAL, "a clip of " + std::to_string(n) + " frames exceeds the attention block's limit of " + std::to_string(tsattn_max_frames()) +
                                                             " frames per clip (its softmax row over the clip lives in LDS): split the sequence into clips");
    if (!h->tsattn_ready)
        return h->fail(GRNET_ESTATE, "attention-block weights were not loaded (keys tsattn.* or pfeat_corrector.featTencoder.0.*)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* ws = nullptr;                                   // handle-owned scratch, like the GRU's
    if (int rc = h->temporal_scratch(kGemmWsFloats + tsattn_ws_floats(b, n), &ws)) return rc;
    GemmWorkspaceLease lease(ws, kGemmWsFloats);
    hipError_t e = launch_tsattn(x, xs, h->tsw, ws + kGemmWsFloats, y, b, n, s);
    if (e != hipSuccess) return h->fail(GRNET_EHIP, std::string("launch_tsattn: ") + hipGetErrorString(e));
    return 0;
}

int grnet_set_option(grnet_t* h, int option, int value) {
    if (!h) return GRNET_EINVAL;
    if (option == GRNET_OPT_USE_GRAPH) { h->use_graph = value != 0; return 0; }
    if (option == GRNET_OPT_CONV_TILE) {
        if (value != 0 && value != 7 && value != 14 && value != 1071 && value != 1072 && value != 1041 && value != 1042 && value != 1171 && value != 1141)
            return h->fail(GRNET_EINVAL, "conv tile must be 0, 7, 14 or a split-K code 1071/1072/1041/1042/1171/1141");
        h->conv_tile_hint = value;
        h->drop_graphs();
        return 0;
    }
    if (option == GRNET_OPT_WINOGRAD) { h->wino_mode = value != 0; h->drop_graphs(); return 0; }
    if (option == GRNET_OPT_BF16_CHAIN) { h->chain_mode = value & grnet::kChainModeAll; h->drop_graphs(); return 0; }
    if (option == GRNET_OPT_GRU_MODE) {
        if (value < 0 || (value & 15) > 3 || (value & ~31)) return h->fail(GRNET_EINVAL, "GRU mode is 0 .. 3, + 16 for agent-scope stores");
        h->gru_mode = value;
        return 0;
    }
    if (option == GRNET_OPT_BF16_MIN_FRAMES) {
        if (value < 0) return h->fail(GRNET_EINVAL, "the smallest call of the bf16 kernel groups is >= 1 frame (0: each group's own default)");
        h->bf16_min_frames = value;
        h->drop_graphs();
        return 0;
    }
    if (option == GRNET_OPT_MULTI_LANE) {
        h->multi_lane = value != 0;
        h->drop_graphs();
        return 0;
    }
    return h->fail(GRNET_EINVAL, "unknown option");
}

int grnet_tune(grnet_t* h, int n_frames, void* stream, int level) {
    if (!h) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    return h->tune(n_frames, static_cast<hipStream_t>(stream), level);
}

// Tuned table <-> text ("mode" line + one line per convolution: index hint), so a table measured once on a GPU
// can be stored next to the model and re-applied without re-measuring.
int grnet_get_tuning(grnet_t* h, int n_frames, char* buf, int buf_size) {
    if (!h || !buf || buf_size < 16) return GRNET_EINVAL;
    auto m = h->tuned_mode.find(n_frames);
    if (m == h->tuned_mode.end()) return h->fail(GRNET_ESTATE, "no tuning for this n_frames");
    std::string out = "mode " + std::to_string(m->second) + "\n";
    for (size_t i = 0; i < h->convs.size(); ++i) {
        auto it = h->convs[i].tuned.find(n_frames);
        out += std::to_string(i) + " " + std::to_string(it == h->convs[i].tuned.end() ? 0 : it->second) + "\n";
    }
    if ((int)out.size() + 1 > buf_size) return h->fail(GRNET_EINVAL, "buffer too small");
    memcpy(buf, out.c_str(), out.size() + 1);
    return (int)out.size();
}

int grnet_set_tuning(grnet_t* h, int n_frames, const char* text) {
    if (!h || !text) return GRNET_EINVAL;
    int mode = 0, consumed = 0;
    if (sscanf(text, "mode %d\n%n", &mode, &consumed) != 1) return h->fail(GRNET_EINVAL, "bad tuning text");
    const char* p = text + consumed;
    int idx, hint, used;
    while (sscanf(p, "%d %d\n%n", &idx, &hint, &used) == 2) {
        if (idx < 0 || idx >= (int)h->convs.size()) return h->fail(GRNET_EINVAL, "tuning text does not match this plan");
        h->convs[idx].tuned[n_frames] = hint;
        p += used;
    }
    h->tuned_mode[n_frames] = mode;
    h->drop_graphs();
    return 0;
}

int grnet_num_kernel_launches(grnet_t* h) { return h ? h->launches_last : GRNET_EINVAL; }

int grnet_num_conv_launches(grnet_t* h) { return h ? (int)(h->convs.size() + h->fuse_ups.size()) : GRNET_EINVAL; }

double grnet_conv_flops_per_frame(grnet_t* h) {
    if (!h) return 0;
    double m = 0;
    for (auto& L : h->convs) m += L.macs_per_frame;
    for (auto& fp : h->fuse_ups) m += fp.macs_per_frame;            // the 1x1 fuse terms computed by hr_fuse_up_f32
    return 2.0 * m;
}

double grnet_conv_executed_flops_per_frame_n(grnet_t* h, int n_frames) {
    if (!h) return 0;
    double m = 0;
    for (auto& L : h->convs) m += L.macs_per_frame * h->executed_ratio(L, n_frames);
    for (auto& fp : h->fuse_ups) m += fp.macs_per_frame;
    return 2.0 * m;
}

double grnet_conv_executed_flops_per_frame(grnet_t* h) { return h ? grnet_conv_executed_flops_per_frame_n(h, h->last_n) : 0; }

int grnet_describe_conv(grnet_t* h, int pos, int32_t* info, char* name, int name_size) {
    if (!h || !info || !h->finalized || pos < 0) return GRNET_EINVAL;
    int seen = 0;
    for (const Op& op : h->ops_flat) {
        if (op.kind != Op::CONV && op.kind != Op::FUSEUP) continue;
        if (seen++ != pos) continue;
        if (op.kind == Op::FUSEUP) {                       // the grouped 1x1 up terms of one HR module: Cin = 0 marks the entry
            const FuseUpPlan& fp = h->fuse_ups[op.conv_idx];
            int cout = 0;
            int64_t rd = 0;
            for (int i = 0; i < fp.nb - 1; ++i) cout += kBranchCh[i];
            for (int j = 0; j < fp.nb; ++j) rd += (int64_t)kBranchCh[j] * fp.xs[j].h * fp.xs[j].w;       // every branch output is read once
            const int32_t v[12] = {0, cout, 1, 1, fp.xs[0].h, fp.xs[0].w, fp.xs[0].h, fp.xs[0].w, fp.nb, 1, op.lane, (int32_t)rd};
            memcpy(info, v, sizeof(v));
            if (name && name_size > 0) snprintf(name, name_size, "%sfuse_layers(up)", fp.prefix.c_str());
            return 0;
        }
        const ConvLayer& L = h->convs[op.conv_idx];
        int64_t add_elems = 0;
        for (const AddRef& r : L.adds) add_elems += (int64_t)L.cout * (L.out.h >> r.shift) * (L.out.w >> r.shift);
        const int32_t v[12] = {L.in.c + L.in2.c, L.cout, L.ks, L.stride, L.in.h, L.in.w, L.out.h, L.out.w, (int32_t)L.adds.size(), L.relu,
                               op.lane, (int32_t)add_elems};
        memcpy(info, v, sizeof(v));
        if (name && name_size > 0) snprintf(name, name_size, "%s", L.segs.empty() ? "" : L.segs[0].wkey.c_str());
        return 0;
    }
    return GRNET_EINVAL;
}

double grnet_describe_conv_macs(grnet_t* h, int pos) {
    if (!h || pos < 0) return -1.0;
    int seen = 0;
    for (const Op& op : h->ops_flat) {
        if (op.kind != Op::CONV && op.kind != Op::FUSEUP) continue;
        if (seen++ != pos) continue;
        return op.kind == Op::FUSEUP ? h->fuse_ups[op.conv_idx].macs_per_frame : h->convs[op.conv_idx].macs_per_frame;
    }
    return -1.0;
}

static const Op* nth_conv_op(grnet_t* h, int pos) {
    int seen = 0;
    for (const Op& op : h->ops_flat) {
        if (op.kind != Op::CONV && op.kind != Op::FUSEUP) continue;
        if (seen++ == pos) return &op;
    }
    return nullptr;
}

int grnet_conv_kernel_info(grnet_t* h, int pos, int n_frames, char* name, int name_size, double* executed_macs_per_frame) {
    if (!h || !h->finalized || pos < 0 || n_frames < 1) return GRNET_EINVAL;
    const Op* op = nth_conv_op(h, pos);
    if (!op) return GRNET_EINVAL;
    if (op->kind == Op::FUSEUP) {
        if (name && name_size > 0) snprintf(name, name_size, h->dtype == 1 ? "hr_fuse_up_bf16<%d>" : "hr_fuse_up_f32<%d>", h->fuse_ups[op->conv_idx].nb);
        if (executed_macs_per_frame) *executed_macs_per_frame = h->fuse_ups[op->conv_idx].macs_per_frame;
        return 0;
    }
    const ConvLayer& L = h->convs[op->conv_idx];
    if (name && name_size > 0) snprintf(name, name_size, "%s", h->kernel_name(L, n_frames).c_str());
    if (executed_macs_per_frame) *executed_macs_per_frame = L.macs_per_frame * h->executed_ratio(L, n_frames);
    return 0;
}

int grnet_time_conv(grnet_t* h, int pos, int n_frames, int reps, void* stream, float* us_out) {
    if (!h || !us_out || !h->finalized || pos < 0 || reps < 1 || n_frames < 1 || n_frames > h->max_frames) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    const Op* op = nth_conv_op(h, pos);
    if (!op) return GRNET_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { if (e0) (void)hipEventDestroy(e0); return GRNET_EHIP; }
    int rc = 0, nl = 0;
    auto once = [&]() { return op->kind == Op::FUSEUP ? h->launch_fuse_up_op(h->fuse_ups[op->conv_idx], n_frames, s) : h->launch_conv_op(h->convs[op->conv_idx], h->v_cat.p, n_frames, s, &nl); };
    for (int r = 0; r < 2 && !rc; ++r) rc = once();              // warm: weights and inputs in the caches, as between two steps
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < reps && !rc; ++r) rc = once();
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *us_out = ms * 1e3f / reps;
    return rc;
}

int grnet_op_timeline(grnet_t* h, const float* frames_dev, int n_frames, void* stream, char* buf, int buf_size) {
    if (!h || !buf || buf_size < 1) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    std::string text;
    if (int rc = h->op_timeline(frames_dev, n_frames, static_cast<hipStream_t>(stream), text)) return rc;
    if ((int)text.size() + 1 > buf_size) return h->fail(GRNET_EINVAL, "grnet_op_timeline: buffer too small (" + std::to_string(text.size() + 1) + " bytes needed)");
    memcpy(buf, text.c_str(), text.size() + 1);
    return (int)text.size();
}

int grnet_time_convs(grnet_t* h, int n_frames, void* stream, float* ms_out) {
    if (!h || !ms_out || !h->finalized) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return GRNET_EHIP;
    grnet_outputs_t o{};
    hipEventRecord(e0, s);
    // the frames pointer of the first conv is only read; reuse the concat buffer as a stand-in input
    int rc = h->enqueue(h->v_cat.p, n_frames, o, s, true);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    hipEventElapsedTime(ms_out, e0, e1);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return rc;
}

int grnet_op_conv2d(grnet_t* h, const float* in_dev, int n, int cin, int hgt, int wid, const float* w_host, const float* bias_host,
                    int cout, int ks, int stride, int relu, const float* add_dev, float* out_dev, int tile_hint, void* stream) {
    if (!h || !in_dev || !w_host || !out_dev) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    if (h->dtype == 1) return h->op_conv2d_bf16(in_dev, n, cin, hgt, wid, w_host, bias_host, cout, ks, stride, relu, add_dev, out_dev, tile_hint,
                                                static_cast<hipStream_t>(stream));
    const int taps = ks * ks, TC = conv_pick_tc(cout);
    const int cin_pad = (cin + kConvCK - 1) / kConvCK * kConvCK, cout_pad = (cout + TC - 1) / TC * TC;
    std::vector<float> wp((size_t)taps * cin_pad * cout_pad, 0.f), bp(cout_pad, 0.f);
    for (int co = 0; co < cout; ++co) {
        if (bias_host) bp[co] = bias_host[co];
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < taps; ++t) wp[((size_t)t * cin_pad + ci) * cout_pad + co] = w_host[((size_t)co * cin + ci) * taps + t];
    }
    float *wd = nullptr, *bd = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&wd), wp.size() * 4) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&bd), bp.size() * 4) != hipSuccess)
        return h->fail(GRNET_ENOMEM, "hipMalloc failed");
    if (hipMemcpy(wd, wp.data(), wp.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(bd, bp.data(), bp.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        hipFree(wd); hipFree(bd);
        return h->fail(GRNET_EHIP, "hipMemcpy of the test weights failed");
    }
    const int pad = ks / 2;
    ConvArgs a{};
    a.in = in_dev; a.in_ctot = cin; a.in_coff = 0; a.N = n; a.Cin = cin; a.H = hgt; a.W = wid;
    a.Cout = cout; a.Ho = (hgt + 2 * pad - ks) / stride + 1; a.Wo = (wid + 2 * pad - ks) / stride + 1;
    a.out = out_dev; a.out_ctot = cout; a.out_coff = 0;
    a.w = wd; a.bias = bd; a.CinPad = cin_pad; a.CoutPad = cout_pad; a.ks = ks; a.stride = stride; a.relu = relu;
    if (add_dev) { a.n_add = 1; a.add[0] = add_dev; a.add_ctot[0] = cout; a.add_coff[0] = 0; a.add_shift[0] = 0; }
    a.zeros = h->zeros;
    if (const char* d = GRNET_AB_STR(CONV_DBG)) a.dbg = atoi(d);
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* ud = nullptr;
    if (tile_hint == 2003) { tile_hint = 2001; a.dbg |= 32; }   // 2003: the 4-wave F(4x4,3x3) kernel also where the 8-wave one would run
    if (tile_hint == 2001) {                                   // the F(4x4,3x3) kernel on this one convolution
        if (!conv_wino4_eligible(cin, cout, ks, stride, hgt, wid, add_dev ? 1 : 0) || cin_pad % 8 != 0 || cout_pad % (cout % 64 == 0 ? 64 : 32) != 0) {
            hipFree(wd); hipFree(bd);
            return h->fail(GRNET_EINVAL, "shape not eligible for the F(4x4,3x3) kernel");
        }
        std::vector<double> wf((size_t)cout * cin * 9);
        for (size_t i = 0; i < wf.size(); ++i) wf[i] = w_host[i];
        std::vector<float> uw((size_t)36 * cin_pad * cout_pad);
        pack_wino4_weights(wf.data(), cout, cin, cin_pad, cout_pad, uw.data(), wid);
        if (hipMalloc(reinterpret_cast<void**>(&ud), uw.size() * 4) != hipSuccess || hipMemcpy(ud, uw.data(), uw.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            hipFree(wd); hipFree(bd); if (ud) hipFree(ud);
            return h->fail(GRNET_ENOMEM, "Winograd test weights");
        }
        a.w = ud;
    }
    if (tile_hint == 3002 && !conv_pw_eligible(cin, cout, ks, stride, hgt, wid, add_dev ? 1 : 0)) {   // the register-resident 1x1 kernel on this one convolution
        hipFree(wd); hipFree(bd);
        return h->fail(GRNET_EINVAL, "shape not eligible for the 1x1 kernel");
    }
    if (tile_hint == 3001) {                                   // the flattened-K stem kernel on this one convolution
        if (!conv_stem_eligible(cin, cout, ks, stride, hgt, wid, add_dev ? 1 : 0)) {
            hipFree(wd); hipFree(bd);
            return h->fail(GRNET_EINVAL, "shape not eligible for the stem kernel");
        }
        std::vector<double> wf((size_t)cout * cin * 9);
        for (size_t i = 0; i < wf.size(); ++i) wf[i] = w_host[i];
        std::vector<float> sw(7 * 4 * 64);
        pack_stem_weights(wf.data(), sw.data());
        if (hipMalloc(reinterpret_cast<void**>(&ud), sw.size() * 4) != hipSuccess || hipMemcpy(ud, sw.data(), sw.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            hipFree(wd); hipFree(bd); if (ud) hipFree(ud);
            return h->fail(GRNET_ENOMEM, "stem test weights");
        }
        a.w = ud;
    }
    int w4s_on = 0, w4s_ks = 0;
    if (tile_hint >= 2020 && tile_hint <= 2024) {              // the small-map F(4x4,3x3) kernel, 202k: k waves split the input channels (0: default)
        w4s_on = 1; w4s_ks = tile_hint - 2020;
        if (!conv_wino4s_eligible(cin, cout, ks, stride, hgt, wid, add_dev ? 1 : 0)) {
            hipFree(wd); hipFree(bd);
            return h->fail(GRNET_EINVAL, "shape not eligible for the small-map F(4x4,3x3) kernel");
        }
        std::vector<double> wf((size_t)cout * cin * 9);
        for (size_t i = 0; i < wf.size(); ++i) wf[i] = w_host[i];
        std::vector<float> uw((size_t)36 * cin * cout);
        pack_wino4r_weights(wf.data(), cout, cin, uw.data());
        if (hipMalloc(reinterpret_cast<void**>(&ud), uw.size() * 4) != hipSuccess || hipMemcpy(ud, uw.data(), uw.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            hipFree(wd); hipFree(bd); if (ud) hipFree(ud);
            return h->fail(GRNET_ENOMEM, "Winograd test weights");
        }
        a.w = ud;
    }
    auto launch_one = [&]() {
        return w4s_on ? launch_conv_wino4s(a, s, w4s_ks) : tile_hint == 2001 ? launch_conv_wino4(a, s) : tile_hint == 3001 ? launch_conv_stem(a, s) : tile_hint == 3002 ? launch_conv_pw(a, s) : launch_conv(a, s, tile_hint);
    };
    hipError_t e = launch_one();
    if (const char* r = GRNET_AB_STR(CONV_REPS)) {           // timing loop for tools/conv_micro.py
        const int reps = atoi(r);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, s);
        for (int i = 0; i < reps; ++i) e = launch_one();
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        fprintf(stderr, "[conv_micro] cin %d cout %d k %d s %d hw %d n %d hint %d dbg %d: %.2f us/launch\n", cin, cout, ks, stride, hgt, n,
                tile_hint, a.dbg, ms * 1e3f / reps);
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    hipError_t e2 = hipStreamSynchronize(s);
    hipFree(wd);
    hipFree(bd);
    if (ud) hipFree(ud);
    if (e != hipSuccess) return h->fail(GRNET_EHIP, std::string("launch_conv: ") + hipGetErrorString(e));
    if (e2 != hipSuccess) return h->fail(GRNET_EHIP, std::string("conv kernel: ") + hipGetErrorString(e2));
    return 0;
}

int grnet_op_conv_chain(grnet_t* h, const float* in_dev, int n, int c, int wid, int nconv, const float* w_host, const float* bias_host, float* out_dev,
                        int reps, float* us_out, void* stream) {
    if (!h || !in_dev || !w_host || !bias_host || !out_dev) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    return h->op_conv_chain_bf16(in_dev, n, c, wid, nconv, w_host, bias_host, out_dev, reps, us_out, static_cast<hipStream_t>(stream));
}

int grnet_op_bilinear2x(grnet_t* h, const float* in_dev, int n, int c, int hgt, int wid, float* out_dev, void* stream) {
    if (!h || !in_dev || !out_dev) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h->dtype == 1) {                                       // bf16 handle: the bf16 NHWC kernel of the bf16 path (fp32 NCHW -> bf16 NHWC -> x2 -> fp32 NCHW)
        if (n < 1 || c < 8 || c % 8 != 0 || hgt < 1 || wid < 1) return h->fail(GRNET_EINVAL, "bilinear2x (bf16): channels must be a multiple of 8");
        const size_t nin = (size_t)n * hgt * wid * c;
        void* tmp = nullptr;
        hipError_t eb = hipMallocAsync(&tmp, nin * 2 * 5, s);
        if (eb != hipSuccess) return h->fail(GRNET_EHIP, std::string("bilinear2x (bf16) scratch: ") + hipGetErrorString(eb));
        void* up = static_cast<unsigned short*>(tmp) + nin;
        eb = launch_nchw_f32_to_nhwc_bf16(in_dev, tmp, n, c, hgt, wid, c, s);
        if (eb == hipSuccess) eb = launch_bilinear2x_bf16(tmp, up, n, c, hgt, wid, s);
        if (eb == hipSuccess) eb = launch_nhwc_bf16_to_nchw_f32(up, out_dev, n, c, 2 * hgt, 2 * wid, c, 0, s);
        (void)hipFreeAsync(tmp, s);
        if (eb != hipSuccess) return h->fail(GRNET_EHIP, std::string("bilinear2x (bf16): ") + hipGetErrorString(eb));
        return 0;
    }
    hipError_t e = launch_bilinear2x(in_dev, out_dev, n, c, hgt, wid, s);
    if (e != hipSuccess) return h->fail(GRNET_EHIP, std::string("bilinear2x: ") + hipGetErrorString(e));
    return 0;
}

int grnet_debug_tensor(grnet_t* h, const char* name, int n_frames, float* out_dev, int64_t* shape_out, void* stream) {
    if (!h || !name) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    for (auto& nv : h->named) {
        if (nv.first != name) continue;
        const View& v = nv.second;
        if (shape_out) { shape_out[0] = v.c; shape_out[1] = v.h; shape_out[2] = v.w; }
        if (!out_dev) return 0;
        const size_t plane = (size_t)v.h * v.w;
        if (h->dtype == 1) {
            hipError_t eb = launch_nhwc_bf16_to_nchw_f32(v.p, out_dev, n_frames, v.c, v.h, v.w, v.ctot, v.coff, static_cast<hipStream_t>(stream));
            if (eb != hipSuccess) return h->fail(GRNET_EHIP, std::string("debug convert: ") + hipGetErrorString(eb));
            return 0;
        }
        hipError_t e = hipMemcpy2DAsync(out_dev, (size_t)v.c * plane * 4, v.p + (size_t)v.coff * plane, (size_t)v.ctot * plane * 4,
                                        (size_t)v.c * plane * 4, n_frames, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return h->fail(GRNET_EHIP, std::string("debug copy: ") + hipGetErrorString(e));
        return 0;
    }
    return h->fail(GRNET_EINVAL, std::string("unknown debug tensor ") + name);
}

int grnet_smpl_forward(grnet_t* h, const float* betas_dev, const float* rotmat_dev, const float* cam_dev, int n, float* verts_dev,
                       float* kp3d_dev, float* kp2d_dev, void* stream) {
    if (!h || !betas_dev || !rotmat_dev || !verts_dev || !kp3d_dev || n < 1) return GRNET_EINVAL;
    if (!h->smpl_loaded) return h->fail(GRNET_ESTATE, "SMPL tables were not loaded");
    if (n > h->max_frames) return h->fail(GRNET_EINVAL, "n exceeds max_frames (the skinning-matrix workspace is sized for it)");
    DeviceGuard guard(h->device);
    hipError_t e = launch_smpl(betas_dev, rotmat_dev, cam_dev, h->smpl, h->d_A, verts_dev, kp3d_dev, kp2d_dev, n,
                               static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return h->fail(GRNET_EHIP, std::string("smpl: ") + hipGetErrorString(e));
    return 0;
}

int grnet_crop_normalise(grnet_t* h, const unsigned char* images_dev, int n, int height, int width, int one_image_for_all,
                         const float* bboxes_dev, float scale, int bgr, float* out_dev, void* stream) {
    if (!h || !images_dev || !bboxes_dev || !out_dev || n < 1 || height < 1 || width < 1 || !(scale > 0.f)) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    hipError_t e = launch_crop_normalise(images_dev, height, width, one_image_for_all ? 0 : 1, bboxes_dev, scale, bgr, out_dev, n,
                                         static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return h->fail(GRNET_EHIP, std::string("crop_normalise: ") + hipGetErrorString(e));
    return 0;
}

int grnet_crop_normalise_cv(grnet_t* h, const unsigned char* images_dev, int n, int height, int width, int one_image_for_all,
                            const double* inv_affine_dev, int bgr, float* out_dev, void* stream) {
    if (!h || !images_dev || !inv_affine_dev || !out_dev || n < 1 || height < 1 || width < 1) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    hipError_t e = launch_crop_normalise_cv(images_dev, height, width, one_image_for_all ? 0 : 1, inv_affine_dev, 6, bgr, out_dev, n,
                                            static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return h->fail(GRNET_EHIP, std::string("crop_normalise_cv: ") + hipGetErrorString(e));
    return 0;
}

int grnet_crop_normalise_cv_maps(grnet_t* h, const unsigned char* images_dev, int n, int height, int width, int one_image_for_all,
                                 const double* maps_dev, int bgr, float* out_dev, void* stream) {
    if (!h || !images_dev || !maps_dev || !out_dev || n < 1 || height < 1 || width < 1) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    hipError_t e = launch_crop_normalise_cv(images_dev, height, width, one_image_for_all ? 0 : 1, maps_dev, 10, bgr, out_dev, n,
                                            static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return h->fail(GRNET_EHIP, std::string("crop_normalise_cv_maps: ") + hipGetErrorString(e));
    return 0;
}

const char* grnet_last_error(grnet_t* h) { return h ? h->err.c_str() : "null handle"; }

void grnet_destroy(grnet_t* h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    delete h;                                              // ~grnet releases graphs, streams, events and device memory
}

// PareHead.forward + VPRegressor.forward from given pooled features -- lib/models/pare.py:271-303,52-91: the second head pass of the
// use_gait_feat branch (grnet.py:165,171) and the single-op parity hook of the tail.
int grnet_head_forward(grnet_t* h, const float* plf_dev, const float* csf_dev, int n, const grnet_outputs_t* out, void* stream) {
    if (!h || !plf_dev || !csf_dev || !out || n < 1) return GRNET_EINVAL;
    if (!h->finalized) return h->fail(GRNET_ESTATE, "grnet_head_forward before grnet_finalize_weights");
    if (n > h->max_frames) return h->fail(GRNET_EINVAL, "n exceeds max_frames");
    DeviceGuard guard(h->device);
    return h->head_from_feats(plf_dev, csf_dev, n, *out, static_cast<hipStream_t>(stream));
}

int grnet_gait_correct(grnet_t* h, const float* plf_dev, const float* csf_dev, const float* cam_dev, int cam_ld, const float* bbox_dev,
                       const float* cimg_dev, int b, int T, const grnet_outputs_t* out, const grnet_gait_outputs_t* gait, void* stream) {
    if (!h || !plf_dev || !csf_dev || !cam_dev || !bbox_dev || !cimg_dev || !out || b < 1 || T < 1 || cam_ld < 3) return GRNET_EINVAL;
    if (!h->finalized) return h->fail(GRNET_ESTATE, "grnet_gait_correct before grnet_finalize_weights");
    if (!h->gru_ready || !h->tsattn_ready || !h->featcorr_ready)
        return h->fail(GRNET_ESTATE, "pose-feature corrector weights were not loaded (keys pfeat_corrector.*)");
    if ((long)b * T > 65536) return h->fail(GRNET_EINVAL, "b*T exceeds 65536 frames");
    DeviceGuard guard(h->device);                          // the limit below is the handle's device's
    if (T > tsattn_max_frames()) return h->fail(GRNET_EINVAL, "a clip of " + std::to_string(T) + " frames exceeds the attention block's limit of " + std::to_string(tsattn_max_frames()) +
                                                             " frames per clip: split the sequence into clips (b, T)");
    grnet_gait_outputs_t g{};
    if (gait) g = *gait;
    return h->gait_correct(plf_dev, csf_dev, cam_dev, cam_ld, bbox_dev, cimg_dev, b, T, *out, g, static_cast<hipStream_t>(stream));
}

int grnet_op_rot6d_to_rotmat(grnet_t* h, const float* rot6d_dev, int m, float* rotmat_dev, void* stream) {
    if (!h || !rot6d_dev || !rotmat_dev || m < 1) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    hipError_t e = launch_rot6d_to_rotmat(rot6d_dev, rotmat_dev, m, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return h->fail(GRNET_EHIP, std::string("rot6d_to_rotmat: ") + hipGetErrorString(e));
    return 0;
}

int grnet_op_rotmat_to_aa(grnet_t* h, const float* rotmat_dev, int m, float* aa_dev, void* stream) {
    if (!h || !rotmat_dev || !aa_dev || m < 1) return GRNET_EINVAL;
    DeviceGuard guard(h->device);
    hipError_t e = launch_rotmat_to_aa(rotmat_dev, aa_dev, m, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return h->fail(GRNET_EHIP, std::string("rotmat_to_aa: ") + hipGetErrorString(e));
    return 0;
}

}  // extern "C"
