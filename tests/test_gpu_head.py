"""The head and the geometry tail on the reference's own vectors: rot6d -> rotmat (geometry.py:395-410) incl. degenerate pairs, rotmat -> axis-angle
(geometry.py:68-97,159-293) on all four quaternion branches, the head pass fed the reference's pooled features (pare.py:338-375), stage taps of the backbone
against reference-run goldens, the element-wise form of the 1e-3 bar, and the attention pooling's merge of range softmaxes (keypoint_attention.py:42-48).
Regrouped by component in round 6; the tests themselves are unchanged."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

from .conftest import CALL_SIZE_NOISE, ROOT, elem_ratio, rel_err

pytestmark = pytest.mark.gpu

@pytest.fixture(scope="module")
def model(pkg):
    m = pkg.build_synthetic_model(max_frames=16, with_gru=True)
    yield m
    m.close()

def _geodesic(R1, R2):
    cos = (np.einsum("nij,nij->n", R1.reshape(-1, 3, 3), R2.reshape(-1, 3, 3)) - 1) / 2
    return np.arccos(np.clip(cos, -1, 1))

def test_rot6d_edge_cases_on_the_gpu(model, golden):
    """tests/golden/geometry.npz was produced by the reference's rot6d_to_rotmat (geometry.py:395-410) on random pairs PLUS the
    degenerate ones (zero vectors, a1 parallel to a2, tiny norms): the device function must reproduce every one of them."""
    g = golden["geometry"]
    got = model.op_rot6d_to_rotmat(torch.from_numpy(g["rot6d"]).cuda()).cpu().numpy()
    ref = g["rotmat"]
    assert got.shape == ref.shape
    assert np.isfinite(got).all()
    # Gram-Schmidt amplifies rounding where a1 and a2 are nearly parallel: hold generic rows to 1e-5 and all rows to the amplified bound
    a = g["rot6d"].reshape(-1, 3, 2)
    a1, a2 = a[:, :, 0], a[:, :, 1]
    n1, n2 = np.linalg.norm(a1, axis=1), np.linalg.norm(a2, axis=1)
    sin = np.linalg.norm(np.cross(a1, a2), axis=1) / np.maximum(n1 * n2, 1e-30)
    generic = (n1 > 1e-3) & (n2 > 1e-3) & (sin > 1e-2)
    assert generic.sum() > 400
    err = np.abs(got - ref).reshape(len(ref), -1).max(1)
    assert err[generic].max() < 1e-5, float(err[generic].max())
    degenerate = ~generic
    assert degenerate.sum() >= 2                                          # the fixture does hold degenerate rows (zero vectors, a1 parallel to a2)
    # degenerate rows: same clamping (eps 1e-6) as the reference, so zero / parallel inputs give the same (non-rotation) matrices
    assert err[degenerate].max() < 2e-3, (float(err[degenerate].max()), np.nonzero(degenerate)[0][:8])

def test_rotmat_to_axis_angle_all_branches_on_the_gpu(model, golden):
    """rotation_matrix_to_angle_axis (geometry.py:68-97): all four quaternion branches, near-pi rotations, the identity
    (sin^2 = 0 -> k = 2) and the NaN scrub, against the reference's own outputs."""
    g = golden["geometry"]
    R = g["rotmat_all"]
    aa = model.op_rotmat_to_aa(torch.from_numpy(R).cuda()).cpu().numpy()
    ref = g["aa"]
    assert aa.shape == ref.shape and not np.isnan(aa).any()
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]                    # branch census on the transposed matrix = same diagonal
    branches = [(m22 < 1e-6) & (m00 > m11), (m22 < 1e-6) & ~(m00 > m11), ~(m22 < 1e-6) & (m00 < -m11), ~(m22 < 1e-6) & ~(m00 < -m11)]
    assert all(b.sum() >= 5 for b in branches), [int(b.sum()) for b in branches]
    d = np.abs(aa - ref).max(1)
    ok = d < 1e-4
    # axis-angle is discontinuous at pi (aa and -aa(2pi - theta) are the same rotation): the few rows that differ element-wise
    # must be the same rotation up to 1e-3 rad
    if (~ok).any():
        from scipy.spatial.transform import Rotation
        ra = Rotation.from_rotvec(aa[~ok].astype(np.float64)).as_matrix()
        rb = Rotation.from_rotvec(ref[~ok].astype(np.float64)).as_matrix()
        assert _geodesic(ra, rb).max() < 2e-3
        assert (~ok).sum() <= 4, int((~ok).sum())

def test_head_pass_single_op_matches_reference_golden(model, golden):
    """PareHead.forward + VPRegressor from given pooled features (the second head pass of grnet.py:165): feed the REFERENCE's
    point_local_feat / cam_shape_feats and compare every output with the reference's own."""
    g = golden["grnet_n4"]
    out = model.head_forward(torch.from_numpy(g["point_local_feat"]).cuda(), torch.from_numpy(g["cam_shape_feats"]).cuda())
    torch.cuda.synchronize()
    o = {k: v.cpu().numpy() for k, v in out.items()}
    assert rel_err(o["pred_rot6d"], g["pred_rot6d"]) < 1e-5
    assert rel_err(o["theta"][:, :3], g["pred_cam"]) < 1e-5 and rel_err(o["theta"][:, 75:], g["pred_shape"]) < 1e-5
    assert rel_err(o["rotmat"], g["pred_rotmat"]) < 1e-5
    assert rel_err(o["theta"], g["theta"].reshape(4, 85)) < 1e-4
    assert rel_err(o["kp_3d"], g["kp_3d"].reshape(4, 29, 3)) < 1e-4 and rel_err(o["kp_2d"], g["kp_2d"].reshape(4, 29, 2)) < 1e-4
    assert rel_err(o["verts"][:, ::5], g["verts_s5"].reshape(4, -1, 3)) < 1e-4
    for k, ref in (("theta", g["theta"].reshape(4, 85)), ("kp_3d", g["kp_3d"].reshape(4, 29, 3)), ("rotmat", g["pred_rotmat"])):
        assert elem_ratio(o[k], ref) <= 1.0, (k, elem_ratio(o[k], ref))
    with pytest.raises(ValueError):
        model.head_forward(torch.zeros(2, 128, 23), torch.zeros(2, 64, 24))

def test_backbone_stage_taps_match_reference_golden(model, pkg, golden):
    """grnet_debug_tensor taps of the HIP backbone (hrnet.py:469-536) against the reference's stage outputs: a parity failure
    is localised to a stage instead of showing up only in `features`."""
    g = golden["grnet_n4"]
    frames = torch.from_numpy(pkg.synth.make_frames(4)).cuda()
    model(frames)
    taps = [("stem_conv1", "stem_conv1_s4", 4), ("stem_conv2", "stem_conv2_s4", 4), ("layer1", "layer1_s4", 4),
            ("stage2.0", "stage2_0_s4", 4), ("stage2.1", "stage2_1_s2", 2),
            ("stage3.0", "stage3_0_s4", 4), ("stage3.1", "stage3_1_s2", 2), ("stage3.2", "stage3_2", 1),
            ("stage4.0", "stage4_0_s4", 4), ("stage4.1", "stage4_1_s2", 2), ("stage4.2", "stage4_2", 1), ("stage4.3", "stage4_3", 1)]
    report = {}
    for name, key, s in taps:
        t = model.debug_tensor(name, 4).cpu().numpy()[..., ::s, ::s]
        assert t.shape == g[key].shape, (name, t.shape, g[key].shape)
        report[name] = rel_err(t, g[key])
    bad = {k: v for k, v in report.items() if not v < 1e-4}
    assert not bad, (bad, report)

def test_elementwise_form_of_the_bar(model, pkg, oracle, synth_weights, synth_smpl):
    """|a-b| <= 1e-3*|b| + 1e-3*rms(b) for EVERY element of every output (theta mixes camera, axis-angle and betas, so it is
    checked per part), next to the tensor-scale form the other tests use."""
    frames = pkg.synth.make_frames(8)
    out = model(torch.from_numpy(frames).cuda())[-1]
    torch.cuda.synchronize()
    ref = oracle.grnet_forward(frames, synth_weights, synth_smpl)
    th, rth = out["theta"].cpu().numpy().reshape(8, 85), np.asarray(ref["theta"]).reshape(8, 85)
    parts = {"cam": (th[:, :3], rth[:, :3]), "pose_aa": (th[:, 3:75], rth[:, 3:75]), "betas": (th[:, 75:], rth[:, 75:])}
    for k in ("kp_3d", "kp_2d", "verts", "rotmat"):
        parts[k] = (out[k].cpu().numpy(), np.asarray(ref[k]))
    ratios = {k: elem_ratio(a, b) for k, (a, b) in parts.items()}
    assert max(ratios.values()) <= 1.0, ratios

@pytest.mark.parametrize("scale", [0.0, 1.0, 60.0], ids=["uniform", "as_is", "peaked"])
def test_attention_pooling_merges_range_softmaxes(pkg, oracle, synth_smpl, scale):
    """The attention pooling computes exp(h - max) per RANGE of 448 positions and head_tail_kernel finishes the softmax over all 3136
    from the seven (max, sum) pairs.  Heat maps scaled to the extremes: all-equal (every range weighs the same), as the synthetic
    weights give them, and x 60 (a few positions carry the whole mass: most ranges' weights underflow to zero) -- pooled features and
    the outputs that follow, against the oracle with the same weights."""
    sd = {k: v.copy() for k, v in pkg.synth.make_state_dict().items()}
    keys = [k for k in sd if "keypoint_final_layer" in k]
    assert len(keys) == 2, keys
    for k in keys:
        sd[k] = (sd[k] * np.float32(scale)).astype(np.float32)
    m = pkg.GRNet(max_frames=4)
    m.load_state_dict(sd, strict=True)
    m.load_smpl(synth_smpl)
    m.finalize()
    frames = pkg.synth.make_frames(3)
    out = m(torch.from_numpy(frames).cuda(), extras=("point_local_feat", "cam_shape_feats"))[-1]
    ref = oracle.grnet_forward(frames, sd, synth_smpl, return_intermediates=True)
    for k in ("point_local_feat", "cam_shape_feats", "theta", "kp_3d"):
        a = out[k].cpu().numpy()
        assert rel_err(a, np.asarray(ref[k]).reshape(a.shape)) < 2e-4, (k, rel_err(a, np.asarray(ref[k]).reshape(a.shape)))
    m.close()
