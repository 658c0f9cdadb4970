"""ORACLE -- CPU restatement of MAX-GRNet's per-frame path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file; the product path (the HIP library behind
``include/grnet_hip.h``) never does and fails loudly without its extension.

What it restates (reference file:line, all under /root/reference):
  * backbone   lib/models/hrnet.py:469-536 (+ blocks :43-59, :80-100, HR module
               :249-267, fuse layers :189-244, transitions :348-387, upsample
               heads :440-453)
  * PARE head  lib/models/pare.py:245-269, 305-336 (conv branches, 1x1 heads),
               layers/keypoint_attention.py:34-55 (softmax pooling),
               pare.py:338-375 + layers/locallyconnected2d.py:39-48 (tail)
  * geometry   lib/utils/geometry.py:395-410 (rot6d), :68-97, :159-293 (rotmat ->
               quaternion -> axis-angle), :427-479 (camera + projection)
  * SMPL       lib/models/smpl.py:108-130, 149-191 wrapping smplx's LBS
  * packing    lib/models/pare.py:52-91
  * GRU        lib/models/layers/gait_feat_encoder.py:79-104 (nn.GRU equations)

Pinning: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/*.npz``, which ``tools/make_goldens.py`` produced by running the
reference itself in the build container on the same seed-defined inputs.

PARITY UNPINNED at one boundary: the SMPL linear-blend-skinning arithmetic lives
in the third-party package smplx (pinned smplx==0.1.26, requirements.txt:13),
which is neither installed nor vendored and for which the reference holds no
test vectors.  ``smpl_lbs`` restates the published algorithm; the
goldens pin it only to the generator's stand-in of the same published algorithm.

Convolutions use torch's CPU kernels (the reference's own CPU path does the
same); everything else is spelled out in numpy so it is an independent
formulation of what the reference computes with torch ops.
"""
import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BRANCH_CH = [32, 64, 128, 256]
PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
EXTRA_VERT_IDS = [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                  2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133]
FOCAL = 5000.0
IMG_RES = 224


def _t(a):
    return a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))


# ----------------------------------------------------------------------------- primitives
def conv2d(x, w, stride=1, bias=None):
    """nn.Conv2d with padding = k//2 (every conv on the path: hrnet.py:24-27,67-73; pare.py:388-395)."""
    w = _t(w)
    return F.conv2d(_t(x), w, None if bias is None else _t(bias), stride=stride, padding=w.shape[-1] // 2)


def batchnorm(x, sd, prefix):
    """BatchNorm2d in eval mode: (x-mean)/sqrt(var+1e-5)*gamma+beta; torch's CPU kernel, as the
    reference's own CPU path uses (the explicit formula is ``batchnorm_explicit``, checked equal in the tests)."""
    return F.batch_norm(x, _t(sd[prefix + ".running_mean"]), _t(sd[prefix + ".running_var"]), _t(sd[prefix + ".weight"]),
                        _t(sd[prefix + ".bias"]), training=False, eps=BN_EPS)


def batchnorm_explicit(x, sd, prefix):
    g, b = _t(sd[prefix + ".weight"]), _t(sd[prefix + ".bias"])
    m, v = _t(sd[prefix + ".running_mean"]), _t(sd[prefix + ".running_var"])
    scale = g / torch.sqrt(v + BN_EPS)
    return x * scale[None, :, None, None] + (b - m * scale)[None, :, None, None]


# bf16 emulation of the HIP path's dtype=1 mode (BASELINE configs 3/5: bf16 storage, fp32 accumulation): every tensor the
# kernels keep in HBM is rounded to bf16 (nearest even) where the kernels round it -- BN-folded weights, each conv / fuse-sum /
# bilinear output -- while sums run in fp32.  Not a second reference: it shows that the bf16 path's distance from the fp32
# oracle is rounding noise of that size and nothing else (tests/test_gpu_bf16.py).
_BF16 = False


class bf16_storage:
    def __enter__(self):
        global _BF16
        self._old, _BF16 = _BF16, True
        return self

    def __exit__(self, *a):
        global _BF16
        _BF16 = self._old


def _q(t):
    return t.to(torch.bfloat16).to(torch.float32) if _BF16 else t


def conv_bn(x, sd, conv_key, bn_prefix, stride=1, relu=False, residual=None):
    if _BF16:
        g, b = _t(sd[bn_prefix + ".weight"]).double(), _t(sd[bn_prefix + ".bias"]).double()
        m, v = _t(sd[bn_prefix + ".running_mean"]).double(), _t(sd[bn_prefix + ".running_var"]).double()
        scale = g / torch.sqrt(v + BN_EPS)
        wf = _q((_t(sd[conv_key]).double() * scale[:, None, None, None]).float())
        y = conv2d(x, wf, stride) + (b - m * scale).float()[None, :, None, None]
    else:
        y = batchnorm(conv2d(x, sd[conv_key], stride), sd, bn_prefix)
    if residual is not None:
        y += residual
    return _q(torch.relu_(y) if relu else y)


def upsample_nearest(x, factor):
    """nn.Upsample(scale_factor=2**k, mode='nearest') (hrnet.py:208): out[y,x] = in[y//f, x//f]."""
    return F.interpolate(x, scale_factor=factor, mode="nearest")


def upsample_bilinear2x(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (hrnet.py:443), torch's CPU kernel
    (the explicit two-tap formula is ``upsample_bilinear2x_explicit``, checked equal in the tests)."""
    return _q(F.interpolate(_t(x), scale_factor=2, mode="bilinear", align_corners=True))


def upsample_bilinear2x_explicit(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (hrnet.py:443).

    src = dst * (in-1)/(out-1); the two taps are floor(src) and min(floor+1, in-1).
    """
    x = _t(x)
    n, c, h, w = x.shape

    def taps(n_in):
        n_out = 2 * n_in
        src = np.arange(n_out, dtype=np.float64) * ((n_in - 1) / (n_out - 1))
        i0 = np.floor(src).astype(np.int64)
        i1 = np.minimum(i0 + 1, n_in - 1)
        lam = (src - i0).astype(np.float32)
        return torch.from_numpy(i0), torch.from_numpy(i1), torch.from_numpy(lam)

    y0, y1, ly = taps(h)
    x0, x1, lx = taps(w)
    rows = x[:, :, y0, :] * (1 - ly)[None, None, :, None] + x[:, :, y1, :] * ly[None, None, :, None]
    return rows[:, :, :, x0] * (1 - lx) + rows[:, :, :, x1] * lx


# ----------------------------------------------------------------------------- backbone
def basic_block(x, sd, p):
    """BasicBlock (hrnet.py:43-59): conv-bn-relu, conv-bn, +x, relu."""
    y = conv_bn(x, sd, p + "conv1.weight", p + "bn1", relu=True)
    return conv_bn(y, sd, p + "conv2.weight", p + "bn2", relu=True, residual=x)


def bottleneck(x, sd, p, has_down):
    """Bottleneck (hrnet.py:80-100)."""
    res = conv_bn(x, sd, p + "downsample.0.weight", p + "downsample.1") if has_down else x
    y = conv_bn(x, sd, p + "conv1.weight", p + "bn1", relu=True)
    y = conv_bn(y, sd, p + "conv2.weight", p + "bn2", relu=True)
    return conv_bn(y, sd, p + "conv3.weight", p + "bn3", relu=True, residual=res)


def hr_fuse(xs, sd, p):
    """The fuse layer of HighResolutionModule.forward (hrnet.py:258-265) on the branch outputs xs; fuse terms per hrnet.py:199-241:
    j > i: conv1x1 + BN + nearest upsample, j == i: identity, j < i: i-j stride-2 conv3x3 + BN (ReLU between them); summed in the
    reference's order j = 0 .. nb-1, then ReLU."""
    nb = len(xs)
    outs = []
    for i in range(nb):
        y = None
        for j in range(nb):
            q = f"{p}fuse_layers.{i}.{j}."
            if j == i:
                t = xs[j]
            elif j > i:
                t = upsample_nearest(conv_bn(xs[j], sd, q + "0.weight", q + "1"), 2 ** (j - i))
            else:
                t = xs[j]
                for k in range(i - j):
                    t = conv_bn(t, sd, q + f"{k}.0.weight", q + f"{k}.1", stride=2, relu=(k != i - j - 1))
            y = t if y is None else y + t
        outs.append(_q(torch.relu_(y)))
    return outs


def hr_module(xs, sd, p):
    """HighResolutionModule.forward (hrnet.py:249-267): four BasicBlocks per branch, then the fuse layer."""
    nb = len(xs)
    xs = list(xs)
    for b in range(nb):
        for k in range(4):
            xs[b] = basic_block(xs[b], sd, f"{p}branches.{b}.{k}.")
    return hr_fuse(xs, sd, p)


def backbone(x, sd, p="backbone.", taps=None):
    """PoseHighResolutionNet.forward with DOWNSAMPLE=False, USE_CONV=True (hrnet.py:469-536)."""
    x = _q(_t(x))
    x = conv_bn(x, sd, p + "conv1.weight", p + "bn1", stride=2, relu=True)
    if taps is not None:
        taps["stem_conv1"] = x
    x = conv_bn(x, sd, p + "conv2.weight", p + "bn2", stride=2, relu=True)
    if taps is not None:
        taps["stem_conv2"] = x
    for b in range(4):
        x = bottleneck(x, sd, f"{p}layer1.{b}.", has_down=(b == 0))
    if taps is not None:
        taps["layer1"] = x
    xs = [conv_bn(x, sd, p + "transition1.0.0.weight", p + "transition1.0.1", relu=True),
          conv_bn(x, sd, p + "transition1.1.0.0.weight", p + "transition1.1.0.1", stride=2, relu=True)]
    xs = hr_module(xs, sd, p + "stage2.0.")
    if taps is not None:
        taps["stage2"] = list(xs)
    xs.append(conv_bn(xs[-1], sd, p + "transition2.2.0.0.weight", p + "transition2.2.0.1", stride=2, relu=True))
    for m in range(4):
        xs = hr_module(xs, sd, f"{p}stage3.{m}.")
    if taps is not None:
        taps["stage3"] = list(xs)
    xs.append(conv_bn(xs[-1], sd, p + "transition3.3.0.0.weight", p + "transition3.3.0.1", stride=2, relu=True))
    for m in range(3):
        xs = hr_module(xs, sd, f"{p}stage4.{m}.")
    if taps is not None:
        taps["stage4"] = list(xs)
    ups = [xs[0]]
    for idx, n_layers, br in ((2, 1, 1), (3, 2, 2), (4, 3, 3)):
        t = xs[br]
        for l in range(n_layers):
            q = f"{p}upsample_stage_{idx}."
            t = conv_bn(upsample_bilinear2x(t), sd, q + f"{4 * l + 1}.weight", q + f"{4 * l + 2}", relu=True)
        ups.append(t)
    return torch.cat(ups, 1)


# ----------------------------------------------------------------------------- PARE head
def head_features(feats, sd, p="head."):
    """feature_extractor up to the pooled features (pare.py:245-263, 305-336)."""
    def branch(name):
        y = conv_bn(feats, sd, f"{p}{name}.0.weight", f"{p}{name}.1", relu=True)
        return conv_bn(y, sd, f"{p}{name}.3.weight", f"{p}{name}.4", relu=True)

    part_feats = branch("keypoint_deconv_layers")
    heat = _q(conv2d(part_feats, _q(_t(sd[p + "keypoint_final_layer.weight"])), bias=sd[p + "keypoint_final_layer.bias"]))
    part_attn = heat[:, 1:]                                     # drop background channel (pare.py:316)
    smpl_feats = branch("smpl_deconv_layers")
    cam_shape = _q(conv2d(smpl_feats, _q(_t(sd[p + "smpl_final_layer.weight"])), bias=sd[p + "smpl_final_layer.bias"]))
    return {"part_feats": part_feats, "part_attn": part_attn, "smpl_feats": smpl_feats, "cam_shape_map": cam_shape}


def keypoint_attention(feat, heat):
    """softmax over H*W per (frame, joint), then attention-weighted feature sum
    (keypoint_attention.py:42-48).  numpy, float64 softmax denominators avoided on purpose:
    float32 throughout like the reference."""
    feat = np.asarray(feat, np.float32)
    heat = np.asarray(heat, np.float32)
    n, j = heat.shape[:2]
    c = feat.shape[1]
    h = heat.reshape(n, j, -1)
    e = np.exp(h - h.max(-1, keepdims=True))
    pnorm = e / e.sum(-1, keepdims=True)
    f = feat.reshape(n, c, -1)
    return np.einsum("njp,ncp->ncj", pnorm, f).astype(np.float32)      # (N, C, J)


def head_tail(plf, csf, sd, p="head."):
    """_pare_get_final_preds (pare.py:338-375) without iteration: per-joint 128->6, Linear 1536->10/3."""
    plf = np.asarray(plf, np.float32)
    csf = np.asarray(csf, np.float32)
    wp = np.asarray(sd[p + "pose_mlp.weight"])[0, :, :, :, 0, 0]          # (6,128,24)
    pose = np.einsum("ncj,ocj->njo", plf, wp)                               # (N,24,6)
    flat = csf.reshape(csf.shape[0], -1)                                    # index c*24+j
    shape = flat @ np.asarray(sd[p + "shape_mlp.weight"]).T + np.asarray(sd[p + "shape_mlp.bias"])
    cam = flat @ np.asarray(sd[p + "cam_mlp.weight"]).T + np.asarray(sd[p + "cam_mlp.bias"])
    return pose.astype(np.float32), shape.astype(np.float32), cam.astype(np.float32)


# ----------------------------------------------------------------------------- geometry
def rot6d_to_rotmat(x):
    """geometry.py:395-410.  x (...,6) viewed (3,2): a1 = elements 0,2,4; a2 = 1,3,5."""
    x = np.asarray(x, np.float32).reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = a1 / np.maximum(np.linalg.norm(a1, axis=1, keepdims=True), 1e-6)
    d = (b1 * a2).sum(1, keepdims=True)
    u = a2 - d * b1
    b2 = u / np.maximum(np.linalg.norm(u, axis=1, keepdims=True), 1e-6)
    b3 = np.cross(b1, b2)
    return np.stack([b1, b2, b3], -1).astype(np.float32)                    # columns b1 b2 b3


def rotmat_to_quat(R, eps=1e-6):
    """geometry.py:213-293 (four-branch, on the transposed matrix)."""
    R = np.asarray(R, np.float32).reshape(-1, 3, 3)
    m = np.transpose(R, (0, 2, 1))
    m00, m11, m22 = m[:, 0, 0], m[:, 1, 1], m[:, 2, 2]
    d2 = m22 < eps
    d0_d1 = m00 > m11
    d0_nd1 = m00 < -m11
    t0 = 1 + m00 - m11 - m22
    q0 = np.stack([m[:, 1, 2] - m[:, 2, 1], t0, m[:, 0, 1] + m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2]], -1)
    t1 = 1 - m00 + m11 - m22
    q1 = np.stack([m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] + m[:, 1, 0], t1, m[:, 1, 2] + m[:, 2, 1]], -1)
    t2 = 1 - m00 - m11 + m22
    q2 = np.stack([m[:, 0, 1] - m[:, 1, 0], m[:, 2, 0] + m[:, 0, 2], m[:, 1, 2] + m[:, 2, 1], t2], -1)
    t3 = 1 + m00 + m11 + m22
    q3 = np.stack([t3, m[:, 1, 2] - m[:, 2, 1], m[:, 2, 0] - m[:, 0, 2], m[:, 0, 1] - m[:, 1, 0]], -1)
    c0 = d2 & d0_d1
    c1 = d2 & ~d0_d1
    c2 = ~d2 & d0_nd1
    q = np.where(c0[:, None], q0, np.where(c1[:, None], q1, np.where(c2[:, None], q2, q3)))
    t = np.where(c0, t0, np.where(c1, t1, np.where(c2, t2, t3)))
    with np.errstate(invalid="ignore", divide="ignore"):
        return (q / np.sqrt(t)[:, None] * np.float32(0.5)).astype(np.float32)


def quat_to_aa(q):
    """geometry.py:159-210."""
    q = np.asarray(q, np.float32)
    q1, q2, q3 = q[:, 1], q[:, 2], q[:, 3]
    s2 = q1 * q1 + q2 * q2 + q3 * q3
    s = np.sqrt(s2)
    c = q[:, 0]
    two_theta = np.float32(2.0) * np.where(c < 0.0, np.arctan2(-s, -c), np.arctan2(s, c))
    with np.errstate(invalid="ignore", divide="ignore"):
        k = np.where(s2 > 0.0, two_theta / s, np.float32(2.0))
    return np.stack([q1 * k, q2 * k, q3 * k], -1).astype(np.float32)


def rotmat_to_aa(R):
    """rotation_matrix_to_angle_axis (geometry.py:68-97): NaN -> 0."""
    aa = quat_to_aa(rotmat_to_quat(R))
    aa[np.isnan(aa)] = 0.0
    return aa


def project(joints, cam):
    """convert_weak_perspective_to_perspective + perspective_projection + /112
    (geometry.py:427-479, smpl.py:172-186)."""
    joints = np.asarray(joints, np.float32)
    cam = np.asarray(cam, np.float32)
    t = np.stack([cam[:, 1], cam[:, 2],
                  np.float32(2 * FOCAL) / (np.float32(IMG_RES) * cam[:, 0] + np.float32(1e-9))], -1)
    p = joints + t[:, None, :]
    p = p / p[:, :, 2:3]
    return (np.float32(FOCAL) * p[:, :, :2] / np.float32(IMG_RES / 2.0)).astype(np.float32)


# ----------------------------------------------------------------------------- SMPL
def smpl_lbs(betas, rotmat, smpl):
    """SMPL linear blend skinning, published algorithm; PARITY UNPINNED vs smplx.

    Per-frame, per-joint loops (a deliberately different formulation from the batched one the
    golden generator's stand-in uses).  Returns verts (N,6890,3), posed joints (N,24,3).
    """
    betas = np.asarray(betas, np.float32)
    R = np.asarray(rotmat, np.float32).reshape(-1, 24, 3, 3)
    vt, sdirs, pdirs = smpl["v_template"], smpl["shapedirs"], smpl["posedirs"]
    Jr, W = smpl["J_regressor"], smpl["lbs_weights"]
    N = betas.shape[0]
    verts = np.empty((N, vt.shape[0], 3), np.float32)
    joints = np.empty((N, 24, 3), np.float32)
    for n in range(N):
        v_shaped = vt + sdirs @ betas[n]                                   # (V,3)
        J = Jr @ v_shaped                                                   # (24,3)
        pose_feat = (R[n, 1:] - np.eye(3, dtype=np.float32)).reshape(207)
        v_posed = v_shaped + (pose_feat @ pdirs).reshape(-1, 3)
        G = np.zeros((24, 4, 4), np.float32)
        for i in range(24):
            T = np.eye(4, dtype=np.float32)
            T[:3, :3] = R[n, i]
            T[:3, 3] = J[i] - (J[PARENTS[i]] if i > 0 else 0)
            G[i] = T if i == 0 else G[PARENTS[i]] @ T
        joints[n] = G[:, :3, 3]
        A = G.copy()
        for i in range(24):
            A[i, :3, 3] -= G[i, :3, :3] @ J[i]
        Tv = np.einsum("vj,jab->vab", W, A)                                 # (V,4,4)
        verts[n] = np.einsum("vab,vb->va", Tv[:, :3, :3], v_posed) + Tv[:, :3, 3]
    return verts, joints


def smpl_joints29(verts, joints24, smpl):
    """The reference wrapper's 29 'spin2' joints (smpl.py:113-118)."""
    j45 = np.concatenate([joints24, verts[:, EXTRA_VERT_IDS]], 1)
    extra = np.einsum("jv,nvk->njk", smpl["J_regressor_extra"], verts)
    return np.concatenate([j45[:, :24], j45[:, [35, 37]], j45[:, [40, 42]], extra[:, 5:6]], 1).astype(np.float32)


# ----------------------------------------------------------------------------- whole path
def grnet_forward(frames, sd, smpl, batch_size=None, return_intermediates=False):
    """GRNet.forward with use_gait_feat=False (grnet.py:129-175) + VPRegressor.forward (pare.py:52-91)."""
    frames = np.asarray(frames, np.float32)
    if frames.ndim == 5:
        b, t = frames.shape[:2]
        frames = frames.reshape(b * t, *frames.shape[2:])
    elif frames.ndim == 4:
        b, t = 1, frames.shape[0]
    else:
        raise ValueError(f"Wrong feature dimension: {frames.ndim}.")
    if batch_size is not None:
        b, t = batch_size, frames.shape[0] // batch_size
    with torch.no_grad():
        feats = backbone(frames, sd)
        hf = head_features(feats, sd)
    plf = keypoint_attention(hf["smpl_feats"].numpy(), hf["part_attn"].numpy())
    csf = keypoint_attention(hf["cam_shape_map"].numpy(), hf["part_attn"].numpy())
    rot6d, shape, cam = head_tail(plf, csf, sd)
    rotmat = rot6d_to_rotmat(rot6d).reshape(-1, 24, 3, 3)
    verts, j24 = smpl_lbs(shape, rotmat, smpl)
    kp3d = smpl_joints29(verts, j24, smpl)
    kp2d = project(kp3d, cam)
    aa = rotmat_to_aa(rotmat.reshape(-1, 3, 3)).reshape(-1, 72)
    theta = np.concatenate([cam, aa, shape], 1)
    out = {
        "theta": theta.reshape(b, t, 85), "verts": verts.reshape(b, t, -1, 3),
        "kp_2d": kp2d.reshape(b, t, -1, 2), "kp_3d": kp3d.reshape(b, t, -1, 3),
        "rotmat": rotmat.reshape(b, t, 24, 3, 3),
    }
    if return_intermediates:
        out.update(features=feats.numpy(), part_attn=hf["part_attn"].numpy(), smpl_feats=hf["smpl_feats"].numpy(),
                   part_feats=hf["part_feats"].numpy(), point_local_feat=plf, cam_shape_feats=csf,
                   pred_rot6d=rot6d, pred_shape=shape, pred_cam=cam)
    return out


# ----------------------------------------------------------------------------- GRU gait encoder
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gru_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One nn.GRU direction, gate order (r,z,n), h0 = 0."""
    b, T, _ = x.shape
    H = w_hh.shape[1]
    gi_all = x @ w_ih.T + b_ih
    h = np.zeros((b, H), np.float32)
    out = np.empty((b, T, H), np.float32)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gi = gi_all[:, t]
        gh = h @ w_hh.T + b_hh
        r = _sigmoid(gi[:, :H] + gh[:, :H])
        z = _sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = ((1 - z) * n + z * h).astype(np.float32)
        out[:, t] = h
    return out, h


def _mlp(x, sd, name, act_tanh=False):
    h = x @ sd[name + ".0.weight"].T + sd[name + ".0.bias"]
    h = np.where(h > 0, h, np.float32(0.05) * h)                            # LeakyReLU(0.05)
    y = h @ sd[name + ".2.weight"].T + sd[name + ".2.bias"]
    return np.tanh(y) if act_tanh else y


def gru_forward(x, cparams, sd):
    """BidirectionalModel.forward, use_pareFeat=True, eval (gait_feat_encoder.py:79-104).

    x (b,T,3072) laid out c*24+j; cparams (b,T,3).  Returns y (b,3), phase (b,T,4), xc (b,T,3072).
    """
    x = np.asarray(x, np.float32)
    cp = np.asarray(cparams, np.float32)
    wc = sd["cparam_mpl.weight"][0, :, :, :, 0, 0]                          # (128,3,24)
    xc = np.einsum("btf,cfj->btcj", cp, wc).reshape(*cp.shape[:2], -1).astype(np.float32)
    h_in = x + xc
    finals = []
    for layer in range(2):
        outs = []
        for suf, rev in (("", False), ("_reverse", True)):
            o, hT = gru_direction(h_in, sd[f"rnn.weight_ih_l{layer}{suf}"], sd[f"rnn.weight_hh_l{layer}{suf}"],
                                  sd[f"rnn.bias_ih_l{layer}{suf}"], sd[f"rnn.bias_hh_l{layer}{suf}"], rev)
            outs.append(o)
            finals.append(hT)
        h_in = np.concatenate(outs, -1)
    hcat = np.concatenate(finals, -1)                                       # [l0f,l0b,l1f,l1b] -> (b,1200)
    y = np.concatenate([_mlp(hcat, sd, "speed_mlp"), _mlp(hcat, sd, "step_mlp")], -1)
    phase = _mlp(h_in, sd, "phase_mlp", act_tanh=True)
    return y.astype(np.float32), phase.astype(np.float32), xc


# ----------------------------------------------------------------------------- preprocessing (row f1)
IMAGENET_MEAN = np.array([0.485, 0.456, 0.406], np.float32)
IMAGENET_STD = np.array([0.229, 0.224, 0.225], np.float32)


def crop_normalise(img_u8, bbox, scale=1.0, crop=224):
    """get_single_image_crop_demo + ToTensor + Normalize (img_utils.py:252-285, 90-113, 54-88, 355-363), rot = 0.

    warpAffine semantics restated: dst(u,v) = bilinear(src, x=(u-112)*w*s/224+cx, y=(v-112)*h*s/224+cy), integer
    coordinates are pixel centres, constant border 0, result rounded to uint8.  PARITY UNPINNED vs OpenCV's own
    fixed-point (1/32 pixel) interpolation -- cv2 is absent offline.
    """
    img = np.asarray(img_u8, np.float32)
    H, W = img.shape[:2]
    cx, cy, w, h = [np.float32(v) for v in bbox]
    u = np.arange(crop, dtype=np.float32)
    x = (u - np.float32(112.0)) * (w * np.float32(scale) / np.float32(224.0)) + cx
    y = (u - np.float32(112.0)) * (h * np.float32(scale) / np.float32(224.0)) + cy
    x0 = np.floor(x).astype(np.int64); y0 = np.floor(y).astype(np.int64)
    ax = (x - np.floor(x)).astype(np.float32); ay = (y - np.floor(y)).astype(np.float32)
    pad = np.zeros((H + 2, W + 2, 3), np.float32)
    pad[1:-1, 1:-1] = img

    def take(yy, xx):
        yy = np.clip(yy + 1, 0, H + 1); xx = np.clip(xx + 1, 0, W + 1)
        ok = ((yy >= 1) & (yy <= H))[:, None] & ((xx >= 1) & (xx <= W))[None, :]
        return pad[yy][:, xx] * ok[..., None]

    top = take(y0, x0) * (1 - ax)[None, :, None] + take(y0, x0 + 1) * ax[None, :, None]
    bot = take(y0 + 1, x0) * (1 - ax)[None, :, None] + take(y0 + 1, x0 + 1) * ax[None, :, None]
    val = np.floor(top * (1 - ay)[:, None, None] + bot * ay[:, None, None] + np.float32(0.5))
    val = np.clip(val, 0, 255) / np.float32(255.0)
    return np.ascontiguousarray(((val - IMAGENET_MEAN) / IMAGENET_STD).transpose(2, 0, 1)).astype(np.float32)


def cv_round(v):
    """cvRound / saturate_cast<int>(double): round half to even (lrint in the default rounding mode)."""
    return np.rint(np.asarray(v, np.float64)).astype(np.int64)


def warp_affine_u8(img_u8, inv_m, size=224):
    """cv2.warpAffine(img, M, (w,h), flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0) for an 8-bit image, given
    the INVERSE map `inv_m` (6 doubles) that warpAffine derives from M; `size`: an int (square) or (width, height).  OpenCV 4.1.2
    imgwarp.cpp restated (third-party source, absent offline -- restated from its published algorithm): WarpAffineInvoker computes
    1/32-pixel fixed-point positions (AB_BITS = 10, INTER_BITS = 5, round_delta = 16), remapBilinear blends the four taps with the
    15-bit table of initInterTab2D and FixedPtCast<int, uchar, 15>.  Returns the uint8 patch (height,width,C)."""
    img = np.asarray(img_u8)
    H, W = img.shape[:2]
    ow, oh = (size, size) if np.isscalar(size) else (int(size[0]), int(size[1]))
    m0, m1, m2, m3, m4, m5 = [float(v) for v in inv_m]
    x, y = np.arange(ow), np.arange(oh)
    adelta, bdelta = cv_round(m0 * x * 1024.0), cv_round(m3 * x * 1024.0)
    X0 = cv_round((m1 * y + m2) * 1024.0) + 16               # indexed by the row y
    Y0 = cv_round((m4 * y + m5) * 1024.0) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    ax, ay = X & 31, Y & 31
    pad = np.zeros((H + 2, W + 2) + img.shape[2:], np.int64)
    pad[1:-1, 1:-1] = img

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = pad[np.clip(yy + 1, 0, H + 1), np.clip(xx + 1, 0, W + 1)]
        return v * ok[..., None]

    w = [(32 - ax) * (32 - ay), ax * (32 - ay), (32 - ax) * ay, ax * ay]
    acc = (tap(sy, sx) * w[0][..., None] + tap(sy, sx + 1) * w[1][..., None] + tap(sy + 1, sx) * w[2][..., None] +
           tap(sy + 1, sx + 1) * w[3][..., None])
    return ((acc * 32 + 16384) >> 15).astype(np.uint8)


def crop_normalise_cv(img_u8, inv_m, crop=224):
    """warp_affine_u8 + ToTensor + Normalize (img_utils.py:355-363): (3,crop,crop) float32."""
    val = warp_affine_u8(img_u8, inv_m, crop).astype(np.float32) / np.float32(255.0)
    return np.ascontiguousarray(((val - IMAGENET_MEAN) / IMAGENET_STD).transpose(2, 0, 1)).astype(np.float32)


def invert_affine_cv(m):
    """The inversion cv2.warpAffine applies to M when WARP_INVERSE_MAP is not set (imgwarp.cpp, in double)."""
    m = [float(v) for v in m]
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m0, m1, m3, m4 = a11, m[1] * -d, m[3] * -d, a22
    return np.array([m0, m1, -m0 * m[2] - m1 * m[5], m3, m4, -m3 * m[2] - m4 * m[5]], np.float64)


def gen_trans_from_patch(c_x, c_y, src_width, src_height, dst_width, dst_height, scale):
    """gen_trans_from_patch_cv with rot = 0, inv = False (img_utils.py:54-88): the two triangles as FLOAT32 points, then
    cv2.getAffineTransform = the 6x6 system of the three point pairs solved in double.  Returns M (6 doubles)."""
    src_w, src_h = float(src_width) * float(scale), float(src_height) * float(scale)
    centre = np.array([c_x, c_y], np.float64)
    src = np.zeros((3, 2), np.float32)
    src[0] = centre
    src[1] = centre + np.array([0, src_h * 0.5], np.float32)
    src[2] = centre + np.array([src_w * 0.5, 0], np.float32)
    dst_center = np.array([dst_width * 0.5, dst_height * 0.5], np.float32)
    dst = np.zeros((3, 2), np.float32)
    dst[0] = dst_center
    dst[1] = dst_center + np.array([0, dst_height * 0.5], np.float32)
    dst[2] = dst_center + np.array([dst_width * 0.5, 0], np.float32)
    a, b = np.zeros((6, 6), np.float64), np.zeros(6, np.float64)
    for k in range(3):
        a[2 * k, 0:2], a[2 * k, 2] = src[k], 1.0
        a[2 * k + 1, 3:5], a[2 * k + 1, 5] = src[k], 1.0
        b[2 * k], b[2 * k + 1] = dst[k]
    return np.linalg.solve(a, b)


def patch_image_cv(img_u8, bbox, scale=1.0, patch=224):
    """generate_patch_image_cv (img_utils.py:90-113) with do_flip = False, rot = 0, as get_single_image_crop_demo calls it (:266-277):
    a square box is ONE warp into the patch; a box with bb_width != bb_height is TWO (:97-106) -- an aspect-preserving resize of the
    scaled box to (int(s*w), int(s*h)), s = patch / max(w, h), then a translation by (patch/2 - width/2, patch/2 - height/2) into the
    patch, whose border stays 0.  The uint8 result of the first warp is what the second one samples.  Returns the uint8 patch."""
    c_x, c_y, bw, bh = [float(v) for v in bbox]
    if bw != bh:
        s = patch / max(bh, bw)
        iw, ih = int(s * bw), int(s * bh)
        first = warp_affine_u8(img_u8, invert_affine_cv(gen_trans_from_patch(c_x, c_y, bw, bh, iw, ih, scale)), (iw, ih))
        dx, dy = patch / 2 - first.shape[1] / 2, patch / 2 - first.shape[0] / 2
        return warp_affine_u8(first, invert_affine_cv([1.0, 0.0, dx, 0.0, 1.0, dy]), patch)
    return warp_affine_u8(img_u8, invert_affine_cv(gen_trans_from_patch(c_x, c_y, bw, bh, patch, patch, scale)), patch)


def crop_normalise_box_cv(img_u8, bbox, scale=1.0, crop=224):
    """patch_image_cv + ToTensor + Normalize (img_utils.py:279, 355-363): (3,crop,crop) float32."""
    val = patch_image_cv(img_u8, bbox, scale, crop).astype(np.float32) / np.float32(255.0)
    return np.ascontiguousarray(((val - IMAGENET_MEAN) / IMAGENET_STD).transpose(2, 0, 1)).astype(np.float32)


# ----------------------------------------------------------------------------- temporal/spatial attention block (row f2)
def layer_normalization(z, gamma, beta, eps=1e-6):
    """The reference's own LayerNormalization (attention_utils.py:10-27): UNBIASED std and (std + eps), not nn.LayerNorm."""
    z = np.asarray(z, np.float32)
    mean = z.mean(-1, keepdims=True, dtype=np.float32)
    std = z.std(-1, keepdims=True, ddof=1, dtype=np.float32)
    return (np.asarray(gamma, np.float32) * ((z - mean) / (std + np.float32(eps))) + np.asarray(beta, np.float32)).astype(np.float32)


def _softmax(a, axis=-1):
    a = a - a.max(axis, keepdims=True)
    e = np.exp(a)
    return (e / e.sum(axis, keepdims=True)).astype(np.float32)


def _gelu(x):
    """nn.GELU() default (exact erf form)."""
    from math import sqrt
    from scipy.special import erf
    return (0.5 * x * (1.0 + erf(x / sqrt(2.0)))).astype(np.float32)


def multi_attention(x, xs, sd, p, num_heads=4):
    """MultiAttention.forward (attention_utils.py:164-217): temporal attention over the n frames of a clip and spatial
    attention over the 25 tokens of a frame run side by side and are mixed by a softmax gate computed from their clip mean.
    x (b,n,128,24), xs (b,n,128,25) -> (b,n,3072)."""
    x, xs = np.asarray(x, np.float32), np.asarray(xs, np.float32)
    b, n = x.shape[:2]
    n_tks = xs.shape[-1]
    W = lambda k: np.asarray(sd[p + k], np.float32)
    lin = lambda v, name: v @ W(name + ".weight").T + W(name + ".bias")
    E = W("qkv_t.weight").shape[0] // 3
    dh = E // num_heads
    qkv_t = lin(x.reshape(b, n, -1), "qkv_t").reshape(b, n, 3, num_heads, dh).transpose(2, 0, 3, 1, 4)      # :170-172
    qt, kt, vt = qkv_t[0], qkv_t[1], qkv_t[2]                                                               # (b,H,n,dh)
    attn = _softmax(qt @ kt.transpose(0, 1, 3, 2) / np.float32(np.sqrt(dh)))                                # :197-199
    x_t = (attn @ vt).transpose(0, 2, 1, 3).reshape(b, n, num_heads * dh)                                   # :203-204
    qkv_s = lin(xs.reshape(b, n, -1), "qkv_s").reshape(b, n, 3, num_heads, dh).transpose(2, 0, 1, 3, 4)     # :175-177
    qkv_s = qkv_s.reshape(3, b * n, num_heads, dh // n_tks, n_tks)                                          # :178  (C, tokens)
    qs, ks, vs = qkv_s[0], qkv_s[1], qkv_s[2]
    attn_s = _softmax(qs.transpose(0, 1, 3, 2) @ ks)                                                        # :209-210 (no scaling)
    x_s = (attn_s @ vs.transpose(0, 1, 3, 2)).transpose(0, 1, 3, 2).reshape(b, n, -1)                       # :214-217, :181
    alpha = np.concatenate([x_t, x_s], -1).mean(1, keepdims=True, dtype=np.float32)                         # :183-184
    alpha = _softmax(lin(alpha, "ts_attn").reshape(b, 1, -1, 2))                                            # :185-186
    return (lin(x_t * alpha[..., 0], "fc_t") + lin(x_s * alpha[..., 1], "fc_s")).astype(np.float32)         # :188


def joint_wise_ffn(x, sd, p, num_token=24):
    """JointWiseFeedForward.forward (attention_utils.py:123-130): two per-token locally connected layers
    (locallyconnected2d.py:39-48, kernel 1) with an exact GELU in between.  x (b,n,3072) index c*24+j."""
    b, n, f = x.shape
    w1 = np.asarray(sd[p + "jwff_layer1.weight"], np.float32)[0, :, :, :, 0, 0]      # (64,128,24)
    w2 = np.asarray(sd[p + "jwff_layer2.weight"], np.float32)[0, :, :, :, 0, 0]      # (128,64,24)
    v = x.reshape(b * n, f // num_token, num_token)
    h = _gelu(np.einsum("rcj,ocj->roj", v, w1))
    return np.einsum("roj,poj->rpj", h, w2).reshape(b, n, -1).astype(np.float32)


def ts_attn_block(x, xs, sd, p="", num_heads=4, num_token=24):
    """TSAttnBlock.forward with use_jwff=True, eval (attention_utils.py:261-270)."""
    x = np.asarray(x, np.float32)
    b, n = x.shape[:2]
    y = x.reshape(b, n, -1) + multi_attention(x, xs, sd, p + "mulattn.", num_heads)
    y = layer_normalization(y, sd[p + "norm1.gamma"], sd[p + "norm1.beta"])
    return layer_normalization(joint_wise_ffn(y, sd, p + "ffn.", num_token) + y, sd[p + "norm2.gamma"], sd[p + "norm2.beta"])


# ----------------------------------------------------------------------------- pose-feature corrector (row f2)
def feat_corrector(x, cparams, sd, p="pfeat_corrector."):
    """FeatCorrector.forward (feature_correction.py:104-157), eval, in the configuration GRNet builds (grnet.py:69-79 with
    configs/config_grnet.yaml: one layer, 4 heads, h_size 1024 -> 1000, use_jwff).  The reference class cannot be constructed as
    shipped (undefined names); the names are bound as DESIGN.md records: use_leff = leff_smpl_feats = False (their
    branch only unpacks shapes), `N` (:144) = n, everything else is stored and never read in forward.  Composition of the
    separately pinned module restatements: gru_forward (gait_feat_encoder.py:79-104) and ts_attn_block (attention_utils.py:261-270).

    x (b,n,3072) index c*24+j, cparams (b,n,3) -> y (b*n,128,24), pred_avg (b,3), pred_phase (b,n,4).
    """
    x = np.asarray(x, np.float32)
    b, n, _ = x.shape
    gsd = {k[len(p + "featnet."):]: v for k, v in sd.items() if k.startswith(p + "featnet.")}
    pred_avg, pred_phase, _ = gru_forward(x, cparams, gsd)
    n1 = np.linalg.norm(pred_phase[:, :, :2], axis=-1, keepdims=True)
    n2 = np.linalg.norm(pred_phase[:, :, 2:], axis=-1, keepdims=True)
    phase = pred_phase / np.concatenate([n1, n1, n2, n2], -1)
    raw = np.concatenate([np.broadcast_to(pred_avg[:, None, :], (b, n, pred_avg.shape[-1])), phase], -1).astype(np.float32)

    def mlp(name):
        h = raw @ sd[f"{p}{name}.0.weight"].T + sd[f"{p}{name}.0.bias"]
        h = np.where(h > 0, h, np.float32(0.05) * h)                        # LeakyReLU(0.05); Dropout = identity in eval
        return (h @ sd[f"{p}{name}.3.weight"].T + sd[f"{p}{name}.3.bias"]).astype(np.float32)

    def bn1d(z, name):                                                      # BatchNorm1d over the feature axis, running statistics
        g, be = sd[f"{p}{name}.weight"], sd[f"{p}{name}.bias"]
        m, v = sd[f"{p}{name}.running_mean"], sd[f"{p}{name}.running_var"]
        return ((z - m) / np.sqrt(v + np.float32(1e-5)) * g + be).astype(np.float32)

    x_wgf = x + mlp("gfeat_mpl_t")
    x_wgf_s = np.concatenate([x, mlp("gfeat_mpl_s")], -1)
    y = bn1d(x_wgf, "bn_in")
    y_s = bn1d(x_wgf_s, "bn_in_s")
    tsd = {k[len(p + "featTencoder.0."):]: v for k, v in sd.items() if k.startswith(p + "featTencoder.0.")}
    y = ts_attn_block(y.reshape(b, n, 128, -1), y_s.reshape(b, n, 128, -1), tsd)
    y = y[:, :n, :3072]
    return (y + x).reshape(b * n, -1, 24).astype(np.float32), pred_avg, pred_phase


def gait_cparams(pred_cam, bbox, cimg):
    """grnet.py:156-160: camera parameters in the full image from the crop camera and the box."""
    cam = np.asarray(pred_cam, np.float32).reshape(-1, 3)
    bbox, cimg = np.asarray(bbox, np.float32), np.asarray(cimg, np.float32)
    bs = bbox[..., 2] / np.float32(224.0)
    t_bb = bbox[..., :2] - cimg
    scale = bs.reshape(-1, 1) * cam[:, 0:1]
    return np.concatenate([scale, t_bb.reshape(-1, 2) / scale / np.float32(112.0) + cam[:, 1:]], -1).astype(np.float32)


def grnet_forward_gait(frames, bbox, cimg, sd, smpl):
    """GRNet.forward with use_gait_feat=True (grnet.py:129-175): first head pass, cparams from the predicted camera and the box,
    FeatCorrector, SECOND head pass on the corrected pose features, regressor.  frames (b,T,3,224,224), bbox (b,T,4), cimg (b,T,2)."""
    frames = np.asarray(frames, np.float32)
    b, t = frames.shape[:2]
    first = grnet_forward(frames, sd, smpl, return_intermediates=True)
    cparams = gait_cparams(first["pred_cam"], bbox, cimg)
    plf, csf = first["point_local_feat"], first["cam_shape_feats"]
    new_plf, pred_avg, pred_phase = feat_corrector(plf.reshape(b, t, -1), cparams.reshape(b, t, 3), sd)
    rot6d, shape, cam = head_tail(new_plf, csf, sd)
    rotmat = rot6d_to_rotmat(rot6d).reshape(-1, 24, 3, 3)
    verts, j24 = smpl_lbs(shape, rotmat, smpl)
    kp3d = smpl_joints29(verts, j24, smpl)
    kp2d = project(kp3d, cam)
    aa = rotmat_to_aa(rotmat.reshape(-1, 3, 3)).reshape(-1, 72)
    theta = np.concatenate([cam, aa, shape], 1)
    return {"theta": theta.reshape(b, t, 85), "verts": verts.reshape(b, t, -1, 3), "kp_2d": kp2d.reshape(b, t, -1, 2),
            "kp_3d": kp3d.reshape(b, t, -1, 3), "rotmat": rotmat.reshape(b, t, 24, 3, 3), "pred_avg": pred_avg,
            "pred_phase": pred_phase, "pred_cparam": cparams, "point_local_feat": new_plf, "first_pass": first}
