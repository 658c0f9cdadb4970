"""Pin the oracle (CPU restatement) to outputs of the reference itself.

tests/golden/*.npz were produced by tools/make_goldens.py, which imports the reference
from /root/reference in the build container and runs it on the seed-defined inputs of synth.py.
"""
import os

import numpy as np
import pytest
import torch

from .conftest import ROOT, rel_err

TOL = 2e-5          # fp32 re-association noise through ~110 layers (SURVEY: 1e-6 self-noise)


@pytest.fixture(scope="module")
def oracle_run(pkg, oracle, synth_weights, synth_smpl):
    frames = pkg.synth.make_frames(4).reshape(2, 2, 3, 224, 224)
    taps = {}
    with torch.no_grad():
        oracle.backbone(frames.reshape(4, 3, 224, 224), synth_weights, taps=taps)
    out = oracle.grnet_forward(frames, synth_weights, synth_smpl, return_intermediates=True)
    out["taps"] = taps
    return out


def test_backbone_stages(oracle_run, golden):
    g = golden["grnet_n4"]
    t = oracle_run["taps"]
    pairs = [
        ("stem_conv1_s4", t["stem_conv1"][..., ::4, ::4]), ("stem_conv2_s4", t["stem_conv2"][..., ::4, ::4]),
        ("layer1_s4", t["layer1"][..., ::4, ::4]),
        ("stage2_0_s4", t["stage2"][0][..., ::4, ::4]), ("stage2_1_s2", t["stage2"][1][..., ::2, ::2]),
        ("stage3_0_s4", t["stage3"][0][..., ::4, ::4]), ("stage3_1_s2", t["stage3"][1][..., ::2, ::2]),
        ("stage3_2", t["stage3"][2]),
        ("stage4_0_s4", t["stage4"][0][..., ::4, ::4]), ("stage4_1_s2", t["stage4"][1][..., ::2, ::2]),
        ("stage4_2", t["stage4"][2]), ("stage4_3", t["stage4"][3]),
    ]
    for name, mine in pairs:
        assert rel_err(mine.numpy(), g[name]) < TOL, name


def test_backbone_features(oracle_run, golden):
    g = golden["grnet_n4"]
    f = oracle_run["features"]
    assert f.shape == (4, 480, 56, 56)
    assert rel_err(f[..., ::4, ::4], g["features_s4"]) < TOL
    assert rel_err(np.abs(f).mean((2, 3)), g["features_chan_absmean"]) < TOL


def test_head(oracle_run, golden):
    g = golden["grnet_n4"]
    o = oracle_run
    assert rel_err(o["part_attn"][..., ::2, ::2], g["part_attn_s2"]) < TOL
    assert rel_err(o["smpl_feats"][..., ::4, ::4], g["smpl_feats_s4"]) < TOL
    assert rel_err(o["part_feats"][..., ::4, ::4], g["part_feats_s4"]) < TOL
    for k in ("point_local_feat", "cam_shape_feats", "pred_rot6d", "pred_shape", "pred_cam"):
        assert o[k].shape == g[k].shape, k
        assert rel_err(o[k], g[k]) < 5e-5, k


def test_outputs(oracle_run, golden):
    g = golden["grnet_n4"]
    o = oracle_run
    for k in ("theta", "kp_3d", "kp_2d", "rotmat"):
        assert o[k].shape == g[k].shape, k
        assert rel_err(o[k], g[k]) < 1e-4, k
    assert o["verts"].shape == (2, 2, 6890, 3)
    assert rel_err(o["verts"][:, :, ::5], g["verts_s5"]) < 1e-4
    assert rel_err(o["verts"][0, 0], g["verts_frame0"]) < 1e-4
    mpjpe = np.linalg.norm(o["kp_3d"] - g["kp_3d"], axis=-1).mean()
    assert mpjpe < 1e-5


def test_geometry_edge_cases(oracle, golden):
    g = golden["geometry"]
    rm = oracle.rot6d_to_rotmat(g["rot6d"])
    assert np.allclose(rm, g["rotmat"], atol=2e-6, equal_nan=True)
    aa = oracle.rotmat_to_aa(g["rotmat_all"])
    ref = g["aa"]
    # discontinuous near pi: compare as rotations where the element-wise check fails
    bad = np.abs(aa - ref).max(1) > 1e-4
    assert bad.sum() <= 2, np.nonzero(bad)
    assert not np.isnan(aa).any()


def test_gru(pkg, oracle, golden):
    g = golden["gru"]
    sd = pkg.synth.make_gru_state_dict()
    for (b, t) in ((2, 6), (1, 16)):
        x, cp = pkg.synth.make_gru_inputs(b, t)
        y, ph, xc = oracle.gru_forward(x, cp, sd)
        assert rel_err(y, g[f"y_{b}_{t}"]) < 1e-5
        assert rel_err(ph, g[f"phase_{b}_{t}"]) < 1e-5
        assert rel_err(xc, g[f"xc_{b}_{t}"]) < 1e-5


def test_spec_counts(pkg):
    spec = pkg.netspec.grnet_spec()
    assert sum(k.startswith("backbone.") for k in spec) == 1868
    assert sum(k.startswith("head.") for k in spec) == 37
    n_conv = sum(1 for k, (s, r) in spec.items() if len(s) == 4 and not k.startswith("backbone.final_layer"))
    assert n_conv == 317                                               # 317 conv calls per frame


def test_explicit_formulas_equal_torch_kernels(oracle, synth_weights):
    """The oracle uses torch's CPU batch-norm / interpolate kernels for speed; their spelled-out formulas agree."""
    g = np.random.Generator(np.random.Philox(key=[5, 5]))
    x = torch.from_numpy(g.standard_normal((2, 64, 14, 14)).astype(np.float32))
    a = oracle.batchnorm(x, synth_weights, "backbone.bn1")
    b = oracle.batchnorm_explicit(x, synth_weights, "backbone.bn1")
    assert rel_err(a.numpy(), b.numpy()) < 1e-6
    assert rel_err(oracle.upsample_bilinear2x(x).numpy(), oracle.upsample_bilinear2x_explicit(x).numpy()) < 5e-6
    n = oracle.upsample_nearest(x, 4)
    assert torch.equal(n, x.repeat_interleave(4, 2).repeat_interleave(4, 3))


def test_crop_normalise_restatement(oracle):
    """Geometry of gen_trans_from_patch_cv / warpAffine (img_utils.py:54-113) on cases with a known answer."""
    img = np.zeros((300, 400, 3), np.uint8)
    img[100:200, 150:250] = (255, 128, 0)
    x = oracle.crop_normalise(img, [200.0, 150.0, 224.0, 224.0], scale=1.0)        # identity scale: crop == image window
    assert x.shape == (3, 224, 224)
    want = (np.array([255, 128, 0], np.float32) / 255 - oracle.IMAGENET_MEAN) / oracle.IMAGENET_STD
    assert np.allclose(x[:, 112, 112], want, atol=1e-6)                              # dst (112,112) <- src (cx,cy)
    u = 112 + (150 - 200)                                                            # src x = 150 is the square's first column
    assert np.allclose(x[:, 112, u], want, atol=1e-6) and not np.allclose(x[:, 112, u - 1], want, atol=1e-3)
    black = (0 - oracle.IMAGENET_MEAN) / oracle.IMAGENET_STD
    y = oracle.crop_normalise(img, [10.0, 10.0, 224.0, 224.0], scale=1.0)           # window hangs over the border: zeros
    assert np.allclose(y[:, 0, 0], black, atol=1e-6)
    z = oracle.crop_normalise(img, [200.0, 150.0, 112.0, 112.0], scale=2.0)         # scale multiplies the box
    assert np.allclose(z, x)


def test_tsattn_block_oracle_matches_reference_golden(pkg, oracle):
    """Row f2: TSAttnBlock (attention_utils.py:219-270) restated in oracle.ts_attn_block vs outputs of the reference module
    itself (tools/make_goldens.py), same seeded weights and inputs."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "tsattn.npz"))
    sd = pkg.synth.make_tsattn_state_dict()
    for (b, t) in ((2, 8), (1, 16)):
        x, xs = pkg.synth.make_tsattn_inputs(b, t)
        a = oracle.multi_attention(x, xs, sd, "mulattn.")
        assert rel_err(a[:, :, ::8], g[f"attn_{b}_{t}"]) < 2e-5
        y = oracle.ts_attn_block(x, xs, sd)
        assert y.shape == (b, t, 3072)
        assert rel_err(y, g[f"y_{b}_{t}"]) < 2e-5


def test_layer_normalization_is_the_reference_variant(oracle):
    """Unbiased std and (std + eps): differs from nn.LayerNorm by sqrt(n/(n-1)) -- 1.6e-4 relative at n = 3072."""
    rng = np.random.default_rng(3)
    z = rng.standard_normal((5, 3072)).astype(np.float32)
    out = oracle.layer_normalization(z, np.ones(3072, np.float32), np.zeros(3072, np.float32))
    ref = torch.nn.functional.layer_norm(torch.from_numpy(z), (3072,), eps=0.0).numpy() * np.sqrt(3071.0 / 3072.0)
    assert rel_err(out, ref) < 1e-5
    assert abs(out.std(-1, ddof=1).mean() - 1.0) < 1e-4


def test_feat_corrector_matches_reference(pkg, oracle):
    """Row f2: oracle.feat_corrector vs outputs of the reference's FeatCorrector.forward run with its undefined names bound
    (tools/make_goldens_featcorr.py)."""
    import os
    from .conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "featcorr.npz"))
    sd = pkg.synth.make_featcorr_state_dict()
    for (b, n) in ((2, 8), (1, 16)):
        x, cp = pkg.synth.make_featcorr_inputs(b, n)
        y, avg, ph = oracle.feat_corrector(x, cp, sd)
        assert y.shape == (b * n, 128, 24)
        assert rel_err(y, g[f"y_{b}_{n}"]) < 2e-5, rel_err(y, g[f"y_{b}_{n}"])
        assert rel_err(avg, g[f"avg_{b}_{n}"]) < 1e-5 and rel_err(ph, g[f"phase_{b}_{n}"]) < 1e-5


def test_gait_branch_matches_reference(pkg, oracle, synth_weights, synth_smpl):
    """GRNet.forward with use_gait_feat=True (grnet.py:154-173) on 4 frames: cparams, second head pass, regressor."""
    import os
    from .conftest import ROOT
    g = np.load(os.path.join(ROOT, "tests", "golden", "featcorr.npz"))
    sd = dict(synth_weights)
    sd.update(pkg.synth.make_featcorr_state_dict())
    frames = pkg.synth.make_frames(4).reshape(1, 4, 3, 224, 224)
    bbox, cimg = pkg.synth.make_gait_boxes(1, 4)
    out = oracle.grnet_forward_gait(frames, bbox, cimg, sd, synth_smpl)
    assert rel_err(out["pred_cparam"], g["gait_pred_cparam"]) < 1e-5
    assert rel_err(out["pred_avg"], g["gait_pred_avg"]) < 1e-4 and rel_err(out["pred_phase"], g["gait_pred_phase"]) < 1e-4
    for k in ("theta", "kp_3d", "kp_2d", "rotmat"):
        assert rel_err(out[k], g["gait_" + k]) < 1e-4, (k, rel_err(out[k], g["gait_" + k]))
    assert rel_err(out["verts"][:, :, ::5], g["gait_verts_s5"]) < 1e-4
    # the corrector does change the pose: the branch is not a no-op on these weights
    first = out["first_pass"]
    assert rel_err(out["theta"][..., 3:75], first["theta"][..., 3:75]) > 1e-2


# ----------------------------------------------------------------------------- row f1: OpenCV's warpAffine arithmetic, known answers
def _cv_crop(oracle, pkg, img, box, scale=1.0):
    inv = pkg.pipeline.cv_inverse_affine(np.asarray([box]), scale)[0]
    return oracle.warp_affine_u8(img, inv), inv


def test_cv_crop_known_answers(oracle, pkg):
    """The restated cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) on cases whose result is known without running OpenCV: the
    identity, integer translations (with the zero border), exact 2:1 decimation, the half-pixel blend (S0 + S1 + 1) >> 1 of the
    15-bit table, and the snap of positions to the nearest 1/32 pixel."""
    g = np.random.Generator(np.random.Philox(key=[41, 41]))
    img = g.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    out, inv = _cv_crop(oracle, pkg, img, [112.0, 112.0, 224.0, 224.0])
    assert np.allclose(inv, [1, 0, 0, 0, 1, 0], atol=1e-12) and np.array_equal(out, img)
    big = g.integers(0, 256, (300, 400, 3), dtype=np.uint8)
    out, _ = _cv_crop(oracle, pkg, big, [150.0, 120.0, 224.0, 224.0])
    assert np.array_equal(out, big[8:232, 38:262])
    out, _ = _cv_crop(oracle, pkg, big, [10.0, 290.0, 224.0, 224.0])              # mostly outside: constant border 0
    want = np.zeros((224, 224, 3), np.uint8)
    want[:122, 102:] = big[178:300, 0:122]
    assert np.array_equal(out, want)
    out, _ = _cv_crop(oracle, pkg, big, [200.0, 150.0, 448.0, 448.0])             # x_src = 2 (x - 112) + 200: pure picks, zero outside
    ys, xs = 2 * (np.arange(224) - 112) + 150, 2 * (np.arange(224) - 112) + 200
    want = np.zeros((224, 224, 3), np.uint8)
    oky, okx = (ys >= 0) & (ys < 300), (xs >= 0) & (xs < 400)
    want[np.ix_(oky, okx)] = big[np.ix_(ys[oky], xs[okx])]
    assert np.array_equal(out, want)
    out, _ = _cv_crop(oracle, pkg, big, [200.0, 150.0, 112.0, 112.0])             # x_src = (x - 112)/2 + 200: odd x sits on .5
    a = big[150 + (np.arange(0, 224, 2) - 112) // 2][:, 200 + (np.arange(1, 224, 2) - 113) // 2].astype(np.int32)
    b = big[150 + (np.arange(0, 224, 2) - 112) // 2][:, 201 + (np.arange(1, 224, 2) - 113) // 2].astype(np.int32)
    assert np.array_equal(out[0::2, 1::2], ((a + b + 1) >> 1).astype(np.uint8))
    # positions snap to 1/32 pixel: a shift of 0.01 px is the integer translation, 0.02 px is 1/32 of the way to the neighbour
    base, _ = _cv_crop(oracle, pkg, big, [150.0, 120.0, 224.0, 224.0])
    out, _ = _cv_crop(oracle, pkg, big, [150.01, 120.0, 224.0, 224.0])
    assert np.array_equal(out, base)
    out, _ = _cv_crop(oracle, pkg, big, [150.02, 120.0, 224.0, 224.0])
    s0, s1 = big[8:232, 38:262].astype(np.int64), big[8:232, 39:263].astype(np.int64)
    assert np.array_equal(out, (((s0 * 31 * 32 + s1 * 32) * 32 + 16384) >> 15).astype(np.uint8))


def test_cv_inverse_affine_follows_the_reference_steps(pkg):
    """float32 triangle points, double solve, warpAffine's inversion: for a square box the inverse map is the scale w*s/224 about
    the centre, up to the float32 rounding of the points (that rounding is part of the reference's arithmetic and is kept)."""
    boxes = np.array([[412.3, 233.7, 187.4, 187.4], [100.0, 50.0, 300.0, 300.0]], np.float64)
    inv = pkg.pipeline.cv_inverse_affine(boxes, 1.1)
    for (cx, cy, w, h), m in zip(boxes, inv):
        a = w * 1.1 / 224.0
        assert abs(m[0] - a) < 1e-6 * a and abs(m[4] - a) < 1e-6 * a and abs(m[1]) < 1e-9 and abs(m[3]) < 1e-9
        assert abs(m[2] - (cx - 112 * a)) < 1e-4 and abs(m[5] - (cy - 112 * a)) < 1e-4
    # float32 input boxes are used as float32 values (no hidden re-rounding of the centre)
    b32 = boxes.astype(np.float32)
    inv32 = pkg.pipeline.cv_inverse_affine(b32, 1.1)
    assert np.allclose(inv32, pkg.pipeline.cv_inverse_affine(b32.astype(np.float64), 1.1), rtol=0, atol=0)
    # a non-square box takes the reference's TWO-warp branch (img_utils.py:97-106), which the single inverse map does not model: refused, not stretched
    with pytest.raises(ValueError):
        pkg.pipeline.cv_inverse_affine(np.array([[100.0, 50.0, 300.0, 200.0]]), 1.1)
