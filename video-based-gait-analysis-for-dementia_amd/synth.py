"""Seed-defined synthetic weights, SMPL tables and frames (numpy only).

No checkpoint, SMPL model file or video exists offline, so every
parity test and the bench run on tensors defined here.  Each tensor is drawn
from a counter-based Philox stream keyed by (seed, crc32(reference key name)),
so the container (where the reference itself is imported to make the goldens)
and the GPU box regenerate bit-identical values without shipping any file.

Scales are chosen so activations neither collapse (the reference's own
``init_weights`` N(0, 0.001^2), hrnet.py:543) nor saturate:
He-normal convolutions, BN gamma ~ U(0.5, 1) with a smaller gamma on the last
BN of each residual block and on fuse-layer BNs (several terms are summed),
non-trivial running statistics, and a heat-map layer scaled so the spatial
softmax is neither uniform nor one-hot.
"""
import zlib

import numpy as np

from . import netspec

WEIGHT_SEED = 20240807
SMPL_SEED = 1234
FRAME_SEED = 20240807


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[int(seed), zlib.crc32(name.encode())]))


def _draw(key, shape, role, seed):
    g = _rng(seed, key)
    f32 = np.float32
    if role == "nbt":
        return np.zeros((), np.int64)
    if role == "one":
        return np.ones((), f32)
    if role in ("conv_w", "heat_w"):
        fan_in = shape[1] * shape[2] * shape[3]
        std = np.sqrt(2.0 / fan_in)
        if role == "heat_w":
            std *= 1.5
        return (g.standard_normal(shape) * std).astype(f32)
    if role == "bn_gamma":
        return g.uniform(0.5, 1.0, shape).astype(f32)
    if role == "bnres_gamma":
        return g.uniform(0.15, 0.35, shape).astype(f32)
    if role == "bnskip_gamma":
        return g.uniform(0.5, 0.9, shape).astype(f32)
    if role == "bnfuse_gamma":
        return g.uniform(0.25, 0.5, shape).astype(f32)
    if role.endswith("_beta"):
        return (g.standard_normal(shape) * 0.1).astype(f32)
    if role == "bn_mean":
        return (g.standard_normal(shape) * 0.1).astype(f32)
    if role == "bn_var":
        return g.uniform(0.5, 1.5, shape).astype(f32)
    if role == "bias":
        return (g.standard_normal(shape) * 0.05).astype(f32)
    if role == "small":
        return (g.standard_normal(shape) * 0.1).astype(f32)
    if role == "linear_w":
        return (g.standard_normal(shape) * np.sqrt(1.0 / shape[-1])).astype(f32)
    if role == "cam_w":
        return (g.standard_normal(shape) * 0.05 * np.sqrt(1.0 / shape[-1])).astype(f32)
    if role == "cam_b":
        # weak-perspective scale s ~ 0.9 keeps tz = 2f/(224 s) well conditioned
        return np.array([0.9, 0.02, -0.03], f32)
    if role == "pose_w":
        return (g.standard_normal(shape) * np.sqrt(1.0 / shape[2])).astype(f32)
    if role == "cparam_w":
        return (g.standard_normal(shape) * 0.3).astype(f32)
    if role == "ln_gamma":
        return g.uniform(0.8, 1.2, shape).astype(f32)
    if role == "gru_w":
        return g.uniform(-1.0, 1.0, shape).astype(f32) * f32(1.0 / np.sqrt(300.0))
    if role == "gru_b":
        return g.uniform(-1.0, 1.0, shape).astype(f32) * f32(1.0 / np.sqrt(300.0))
    raise ValueError(f"unknown role {role} for {key}")


def make_state_dict(spec=None, seed=WEIGHT_SEED):
    """dict reference-key -> numpy array for ``spec`` (default: backbone.* + head.*)."""
    if spec is None:
        spec = netspec.grnet_spec()
    return {k: _draw(k, shape, role, seed) for k, (shape, role) in spec.items()}


def make_gru_state_dict(seed=WEIGHT_SEED):
    return make_state_dict(netspec.gru_spec(), seed)


def make_tsattn_state_dict(seed=WEIGHT_SEED):
    return make_state_dict(netspec.tsattn_spec(), seed)


def make_featcorr_state_dict(seed=WEIGHT_SEED, prefix="pfeat_corrector."):
    """FeatCorrector weights under the reference's keys (netspec.featcorr_spec)."""
    return make_state_dict(netspec.featcorr_spec(prefix), seed)


def make_featcorr_inputs(b, n, seed=FRAME_SEED):
    """x (b,n,3072): pooled pose features in the c*24+j layout (grnet.py:163); cparams (b,n,3) as grnet.py:157-160 forms them."""
    g = _rng(seed, f"featcorr_in_{b}_{n}")
    x = (g.standard_normal((b, n, 3072)) * 0.5).astype(np.float32)
    cp = np.concatenate([g.uniform(0.6, 1.2, (b, n, 1)), g.standard_normal((b, n, 2)) * 0.3], -1)
    return x, cp.astype(np.float32)


def make_gait_boxes(b, n, seed=FRAME_SEED, width=1920, height=1080):
    """bbox (b,n,4) [cx,cy,w,h] in image pixels and cimg (b,n,2) = half the image size, as Inference.__getitem__ returns them
    with use_gait_feat=True (inference.py:84-85)."""
    g = _rng(seed, f"gait_boxes_{b}_{n}")
    side = g.uniform(180, 420, (b, n, 1))
    bbox = np.concatenate([g.uniform(400, 1500, (b, n, 1)), g.uniform(300, 800, (b, n, 1)), side, side], -1).astype(np.float32)
    cimg = np.broadcast_to(np.array([width * 0.5, height * 0.5], np.float32), (b, n, 2)).copy()
    return bbox, cimg


def make_tsattn_inputs(b, n, seed=FRAME_SEED):
    """x (b,n,128,24): per-joint pose features; xs (b,n,128,25): the same plus the gait-feature token
    (feature_correction.py:137-148 builds them as batch-normalised features, i.e. roughly unit variance)."""
    g = _rng(seed, f"tsattn_{b}_{n}")
    x = g.standard_normal((b, n, 128, netspec.NUM_JOINTS)).astype(np.float32)
    tok = g.standard_normal((b, n, 128, 1)).astype(np.float32)
    return x, np.concatenate([x, tok], -1)


def make_smpl_tables(seed=SMPL_SEED):
    """Synthetic SMPL model with the real shapes.

    A rough body-sized point cloud; sparse row-stochastic joint regressors;
    row-stochastic skinning weights concentrated on 1-4 joints per vertex.
    """
    V = netspec.NUM_VERTS
    f32 = np.float32
    g = _rng(seed, "smpl")
    v_template = (g.uniform(-1.0, 1.0, (V, 3)) * np.array([0.45, 0.9, 0.15])).astype(f32)
    shapedirs = (g.standard_normal((V, 3, 10)) * 0.02).astype(f32)
    posedirs = (g.standard_normal((207, V * 3)) * 0.01).astype(f32)

    def sparse_rows(rows, nnz):
        m = np.zeros((rows, V), np.float64)
        for r in range(rows):
            idx = g.choice(V, size=nnz, replace=False)
            w = g.uniform(0.1, 1.0, nnz)
            m[r, idx] = w / w.sum()
        return m.astype(f32)

    J_regressor = sparse_rows(24, 32)
    J_regressor_extra = sparse_rows(9, 24)
    lbs = np.zeros((V, 24), np.float64)
    nn = g.integers(1, 5, V)
    for v in range(V):
        idx = g.choice(24, size=int(nn[v]), replace=False)
        w = g.uniform(0.1, 1.0, int(nn[v]))
        lbs[v, idx] = w / w.sum()
    return {
        "v_template": v_template,
        "shapedirs": shapedirs,
        "posedirs": posedirs,
        "J_regressor": J_regressor,
        "lbs_weights": lbs.astype(f32),
        "J_regressor_extra": J_regressor_extra,
        "parents": np.asarray(netspec.SMPL_PARENTS, np.int32),
    }


def make_frames(n, seed=FRAME_SEED, start=0):
    """``n`` pre-normalised frames (n,3,224,224) f32, i.i.d. N(0,1).

    Frame ``start+i`` depends only on (seed, start+i), so a rank that owns a
    contiguous shard of a clip draws exactly the frames the single-GPU run sees.
    """
    out = np.empty((n, 3, 224, 224), np.float32)
    for i in range(n):
        out[i] = _rng(seed, f"frame{start + i}").standard_normal((3, 224, 224), dtype=np.float32)
    return out


def make_gru_inputs(b, t, seed=FRAME_SEED):
    g = _rng(seed, f"gru_in_{b}_{t}")
    x = (g.standard_normal((b, t, 3072)) * 0.5).astype(np.float32)
    cp = np.concatenate([g.uniform(0.6, 1.2, (b, t, 1)), g.standard_normal((b, t, 2)) * 0.3], -1)
    return x, cp.astype(np.float32)
