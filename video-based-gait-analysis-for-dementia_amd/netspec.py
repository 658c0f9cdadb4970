"""Topology and state-dict key map of the MAX-GRNet per-frame path.

This is the host-side description of WHAT the reference builds (HRNet-W32
backbone + PARE head + GRU gait encoder): every tensor of the reference's
``state_dict`` under its reference key name, with its shape and role.  It is
derived from the reference constructors, not copied from them:

* backbone  -- ``lib/models/hrnet.py:276-346`` (stem, layer1, transitions,
  stages 2-4 with 1/4/3 HR modules, ``upsample_stage_{2,3,4}``), config
  ``hrnet.py:584-623`` (width 32, 4 BasicBlocks per branch).
* head      -- ``lib/models/pare.py:145-243`` (two 480->128->128 conv branches,
  1x1 heads, shape/cam linears, per-joint pose weight).
* gru       -- ``lib/models/layers/gait_feat_encoder.py:10-77``.

The same enumeration order is used by the synthetic-weight generator
(``synth.py``), by the oracle and by the tests that check the C-ABI loader
accepts every reference key.
"""
from collections import OrderedDict

WIDTH = 32
BRANCH_CH = [WIDTH, WIDTH * 2, WIDTH * 4, WIDTH * 8]          # 32, 64, 128, 256
STAGES = OrderedDict([("stage2", (1, 2)), ("stage3", (4, 3)), ("stage4", (3, 4))])  # modules, branches
NUM_BLOCKS = 4
NUM_JOINTS = 24
NUM_VERTS = 6890

# SMPL kinematic tree (the published SMPL parents table).
SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
# smplx VertexJointSelector ids appended after the 24 joints.
SMPL_EXTRA_VERT_IDS = [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                       2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133]
# 49 SPIN joints out of [45 smplx joints ++ 9 extra regressed joints]: [JOINT_MAP[n] for n in JOINT_NAMES] (smpl.py:16-87,102)
SPIN49_FROM_54 = [24, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 8, 5, 45, 46, 4, 7,
                  21, 19, 17, 16, 18, 20, 47, 48, 49, 50, 51, 52, 53, 24, 35, 40, 10, 11]
# spin2 (29 joints) -> kinectv2 (25 joints) index map (kp_utils.py:211-242,904-931).
SPIN2_TO_KINECTV2 = [0, 6, 12, 15, 16, 18, 20, 22, 17, 19, 21, 23, 1, 4, 7, 10, 2, 5, 8, 11,
                     28, 25, 24, 27, 26]


def _bn(spec, prefix, c, role="bn"):
    spec[prefix + ".weight"] = ((c,), role + "_gamma")
    spec[prefix + ".bias"] = ((c,), role + "_beta")
    spec[prefix + ".running_mean"] = ((c,), "bn_mean")
    spec[prefix + ".running_var"] = ((c,), "bn_var")
    spec[prefix + ".num_batches_tracked"] = ((), "nbt")


def _conv(spec, key, cout, cin, k, role="conv_w"):
    spec[key] = ((cout, cin, k, k), role)


def backbone_spec(prefix="backbone."):
    """OrderedDict key -> (shape, role) for PoseHighResolutionNet (hrnet.py:276-346)."""
    s = OrderedDict()
    p = prefix
    _conv(s, p + "conv1.weight", 64, 3, 3)
    _bn(s, p + "bn1", 64)
    _conv(s, p + "conv2.weight", 64, 64, 3)
    _bn(s, p + "bn2", 64)
    # layer1: 4 Bottlenecks, planes 64, expansion 4 (hrnet.py:293,389-406)
    inpl = 64
    for b in range(4):
        q = f"{p}layer1.{b}."
        _conv(s, q + "conv1.weight", 64, inpl, 1)
        _bn(s, q + "bn1", 64)
        _conv(s, q + "conv2.weight", 64, 64, 3)
        _bn(s, q + "bn2", 64)
        _conv(s, q + "conv3.weight", 256, 64, 1)
        _bn(s, q + "bn3", 256, role="bnres")
        if b == 0:
            _conv(s, q + "downsample.0.weight", 256, 64, 1)
            _bn(s, q + "downsample.1", 256, role="bnskip")
        inpl = 256
    # transition1 (hrnet.py:301,348-387): [conv3x3 256->32], [[conv3x3 s2 256->64]]
    _conv(s, p + "transition1.0.0.weight", 32, 256, 3)
    _bn(s, p + "transition1.0.1", 32)
    _conv(s, p + "transition1.1.0.0.weight", 64, 256, 3)
    _bn(s, p + "transition1.1.0.1", 64)

    def stage(name, n_mod, n_br):
        for m in range(n_mod):
            for br in range(n_br):
                c = BRANCH_CH[br]
                for k in range(NUM_BLOCKS):
                    q = f"{p}{name}.{m}.branches.{br}.{k}."
                    _conv(s, q + "conv1.weight", c, c, 3)
                    _bn(s, q + "bn1", c)
                    _conv(s, q + "conv2.weight", c, c, 3)
                    _bn(s, q + "bn2", c, role="bnres")
            for i in range(n_br):
                for j in range(n_br):
                    q = f"{p}{name}.{m}.fuse_layers.{i}.{j}."
                    if j > i:
                        _conv(s, q + "0.weight", BRANCH_CH[i], BRANCH_CH[j], 1)
                        _bn(s, q + "1", BRANCH_CH[i], role="bnfuse")
                    elif j < i:
                        for k in range(i - j):
                            last = k == i - j - 1
                            co = BRANCH_CH[i] if last else BRANCH_CH[j]
                            _conv(s, q + f"{k}.0.weight", co, BRANCH_CH[j], 3)
                            _bn(s, q + f"{k}.1", co, role="bnfuse" if last else "bn")

    stage("stage2", *STAGES["stage2"])
    _conv(s, p + "transition2.2.0.0.weight", 128, 64, 3)
    _bn(s, p + "transition2.2.0.1", 128)
    stage("stage3", *STAGES["stage3"])
    _conv(s, p + "transition3.3.0.0.weight", 256, 128, 3)
    _bn(s, p + "transition3.3.0.1", 256)
    stage("stage4", *STAGES["stage4"])
    # final_layer: constructed, never called in forward (hrnet.py:327-333, 469-536)
    s[p + "final_layer.weight"] = ((24, 32, 1, 1), "conv_w")
    s[p + "final_layer.bias"] = ((24,), "bias")
    # upsample heads (hrnet.py:341-344, 440-453): Sequential[Upsample, Conv, BN, ReLU] x n
    for idx, n_layers, c in ((2, 1, 64), (3, 2, 128), (4, 3, 256)):
        for l in range(n_layers):
            _conv(s, f"{p}upsample_stage_{idx}.{4 * l + 1}.weight", c, c, 3)
            _bn(s, f"{p}upsample_stage_{idx}.{4 * l + 2}", c)
    return s


def head_spec(prefix="head."):
    """PareHead tensors (pare.py:181-243), 37 entries."""
    s = OrderedDict()
    p = prefix
    # the module's own buffers precede its sub-modules in state_dict order
    s[p + "temperature"] = ((), "one")
    s[p + "init_pose"] = ((1, 144), "small")
    s[p + "init_shape"] = ((1, 10), "small")
    s[p + "init_cam"] = ((1, 3), "small")
    for br in ("keypoint_deconv_layers", "smpl_deconv_layers"):
        _conv(s, f"{p}{br}.0.weight", 128, 480, 3)
        _bn(s, f"{p}{br}.1", 128)
        _conv(s, f"{p}{br}.3.weight", 128, 128, 3)
        _bn(s, f"{p}{br}.4", 128)
    s[p + "keypoint_final_layer.weight"] = ((25, 128, 1, 1), "heat_w")
    s[p + "keypoint_final_layer.bias"] = ((25,), "bias")
    s[p + "smpl_final_layer.weight"] = ((64, 128, 1, 1), "conv_w")
    s[p + "smpl_final_layer.bias"] = ((64,), "bias")
    s[p + "shape_mlp.weight"] = ((10, 1536), "linear_w")
    s[p + "shape_mlp.bias"] = ((10,), "bias")
    s[p + "cam_mlp.weight"] = ((3, 1536), "cam_w")
    s[p + "cam_mlp.bias"] = ((3,), "cam_b")
    s[p + "pose_mlp.weight"] = ((1, 6, 128, 24, 1, 1), "pose_w")
    return s


def gru_spec(prefix=""):
    """BidirectionalModel(use_pareFeat=True) tensors (gait_feat_encoder.py:30-77)."""
    s = OrderedDict()
    p = prefix
    H, I = 300, 128 * NUM_JOINTS
    s[p + "cparam_mpl.weight"] = ((1, 128, 3, 24, 1, 1), "cparam_w")
    for layer, insz in ((0, I), (1, 2 * H)):
        for suf in ("", "_reverse"):
            s[f"{p}rnn.weight_ih_l{layer}{suf}"] = ((3 * H, insz), "gru_w")
            s[f"{p}rnn.weight_hh_l{layer}{suf}"] = ((3 * H, H), "gru_w")
            s[f"{p}rnn.bias_ih_l{layer}{suf}"] = ((3 * H,), "gru_b")
            s[f"{p}rnn.bias_hh_l{layer}{suf}"] = ((3 * H,), "gru_b")
    for name, nout in (("speed_mlp", 1), ("step_mlp", 2)):
        s[f"{p}{name}.0.weight"] = ((100, 4 * H), "linear_w")
        s[f"{p}{name}.0.bias"] = ((100,), "bias")
        s[f"{p}{name}.2.weight"] = ((nout, 100), "linear_w")
        s[f"{p}{name}.2.bias"] = ((nout,), "bias")
    s[p + "phase_mlp.0.weight"] = ((100, 2 * H), "linear_w")
    s[p + "phase_mlp.0.bias"] = ((100,), "bias")
    s[p + "phase_mlp.2.weight"] = ((4, 100), "linear_w")
    s[p + "phase_mlp.2.bias"] = ((4,), "bias")
    return s


TSATTN = dict(in_dim=128 * NUM_JOINTS, encode_dim=1000, out_dim=128 * NUM_JOINTS, num_heads=4, num_token=NUM_JOINTS)


def tsattn_spec(prefix=""):
    """TSAttnBlock(in_dim=3072, encode_dim=1000, out_dim=3072, num_heads=4, num_token=24, use_jwff=True) tensors
    (attention_utils.py:219-259; the configuration FeatCorrector builds for one layer, feature_correction.py:92-101:
    h_size 1024 rounded down to a multiple of heads * (tokens + 1))."""
    s = OrderedDict()
    p, D, E, T = prefix, TSATTN["in_dim"], TSATTN["encode_dim"], TSATTN["num_token"]
    for n in ("norm1", "norm2"):
        s[f"{p}{n}.gamma"] = ((D,), "ln_gamma")
        s[f"{p}{n}.beta"] = ((D,), "small")
    s[p + "mulattn.qkv_t.weight"] = ((3 * E, D), "linear_w")
    s[p + "mulattn.qkv_t.bias"] = ((3 * E,), "bias")
    s[p + "mulattn.ts_attn.weight"] = ((2 * E, 2 * E), "linear_w")
    s[p + "mulattn.ts_attn.bias"] = ((2 * E,), "bias")
    s[p + "mulattn.qkv_s.weight"] = ((3 * E, D + D // T), "linear_w")
    s[p + "mulattn.qkv_s.bias"] = ((3 * E,), "bias")
    s[p + "mulattn.fc_s.weight"] = ((D, E), "linear_w")
    s[p + "mulattn.fc_s.bias"] = ((D,), "bias")
    s[p + "mulattn.fc_t.weight"] = ((D, E), "linear_w")
    s[p + "mulattn.fc_t.bias"] = ((D,), "bias")
    s[p + "ffn.jwff_layer1.weight"] = ((1, D // 2 // T, D // T, T, 1, 1), "pose_w")
    s[p + "ffn.jwff_layer2.weight"] = ((1, D // T, D // 2 // T, T, 1, 1), "pose_w")
    return s


def featcorr_spec(prefix="pfeat_corrector."):
    """FeatCorrector(x_size=128, num_avg_gfeat=3, estim_phase=True, num_layers=1, h_size=1024 -> 1000, num_joints=24,
    num_transformer_head=4, use_jwff=True) tensors in state_dict order (feature_correction.py:44-101; the configuration of
    configs/config_grnet.yaml FEAT_CORR as GRNet passes it, grnet.py:69-79): the GRU gait encoder `featnet`, the two gait-token
    MLPs, the two input BatchNorm1d and the one TSAttnBlock."""
    s = OrderedDict()
    p, D, G = prefix, 128 * NUM_JOINTS, 128
    s.update(gru_spec(p + "featnet."))
    s[p + "gfeat_mpl_t.0.weight"] = ((D // 2, 7), "linear_w")
    s[p + "gfeat_mpl_t.0.bias"] = ((D // 2,), "bias")
    s[p + "gfeat_mpl_t.3.weight"] = ((D, D // 2), "linear_w")
    s[p + "gfeat_mpl_t.3.bias"] = ((D,), "bias")
    s[p + "gfeat_mpl_s.0.weight"] = ((G // 2, 7), "linear_w")
    s[p + "gfeat_mpl_s.0.bias"] = ((G // 2,), "bias")
    s[p + "gfeat_mpl_s.3.weight"] = ((G, G // 2), "linear_w")
    s[p + "gfeat_mpl_s.3.bias"] = ((G,), "bias")
    _bn(s, p + "bn_in_s", D + G)
    _bn(s, p + "bn_in", D)
    s.update(tsattn_spec(p + "featTencoder.0."))
    return s


def grnet_spec():
    """backbone.* + head.* in the order of the reference's gen_state_dict."""
    s = OrderedDict()
    s.update(backbone_spec())
    s.update(head_spec())
    return s


SMPL_TABLE_SHAPES = OrderedDict([
    ("v_template", (NUM_VERTS, 3)),
    ("shapedirs", (NUM_VERTS, 3, 10)),
    ("posedirs", (207, NUM_VERTS * 3)),
    ("J_regressor", (24, NUM_VERTS)),
    ("lbs_weights", (NUM_VERTS, 24)),
    ("J_regressor_extra", (9, NUM_VERTS)),
])
