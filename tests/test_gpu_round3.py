"""Round-3 GPU parity tests: the register-resident Winograd F(4x4,3x3) kernel of the 14x14 / 7x7 maps (conv_wino4s.hip) against the
oracle's direct convolution, and the attention block of the temporal branch on clips longer than 4 096 frames."""
import numpy as np
import pytest
import torch

from .conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model(pkg):
    m = pkg.build_synthetic_model(max_frames=16, with_gru=False)
    yield m
    m.close()


@pytest.mark.parametrize("case", [(1, 128, 14, 4), (3, 128, 14, 2), (16, 128, 14, 0), (1, 256, 7, 4), (5, 256, 7, 0), (16, 256, 7, 2), (16, 256, 7, 4), (3, 256, 14, 0)],
                         ids=lambda c: "x".join(map(str, c)))
def test_small_map_winograd_kernel(model, oracle, case):
    """conv_wino4s_f32 (F(4x4,3x3) on 14x14 maps = 4x4 tiles padded to 16x16, and on 7x7 maps = 2x2 tiles padded to 8x8 with four
    images per MFMA row tile) vs the oracle's direct convolution: the HR-branch shapes 128 @14x14 and 256 @7x7 and the upsample-head
    layer 256 @14x14; 1 / 3 / 5 / 16 images (5 is not a multiple of the 4 images a 7x7 row tile holds), 2 / 4 waves splitting the
    input channels; bias + ReLU, + residual, the linear form with the borders looked at separately (the right / bottom edge tiles are
    partly outside the map), and bit-identical repeats.  Bound 1e-4 of the output scale, as for the other F(4x4,3x3) kernels."""
    n, c, hw, ksplit = case
    g = np.random.Generator(np.random.Philox(key=[92, n * 100000 + c * 100 + hw + ksplit]))
    x = g.standard_normal((n, c, hw, hw)).astype(np.float32)
    w = (g.standard_normal((c, c, 3, 3)) * np.sqrt(2.0 / (c * 9))).astype(np.float32)
    b = (g.standard_normal((c,)) * 0.1).astype(np.float32)
    r = g.standard_normal((n, c, hw, hw)).astype(np.float32)
    hint = 2020 + ksplit
    conv = oracle.conv2d(x, w, stride=1, bias=b)
    xd = torch.from_numpy(x).cuda()
    got = model.op_conv2d(xd, w, b, stride=1, relu=True, tile_hint=hint).cpu().numpy()
    assert got.shape == conv.shape
    assert rel_err(got, torch.relu(conv).numpy()) < 1e-4, rel_err(got, torch.relu(conv).numpy())
    got = model.op_conv2d(xd, w, b, stride=1, relu=True, add=torch.from_numpy(r).cuda(), tile_hint=hint).cpu().numpy()
    assert rel_err(got, torch.relu(conv + torch.from_numpy(r)).numpy()) < 1e-4
    lin = oracle.conv2d(x, w, stride=1).numpy()
    got = model.op_conv2d(xd, w, None, stride=1, relu=False, tile_hint=hint).cpu().numpy()
    assert rel_err(got, lin) < 1e-4
    assert rel_err(got[:, :, [0, hw - 1]], lin[:, :, [0, hw - 1]]) < 1e-4 and rel_err(got[..., [0, hw - 1]], lin[..., [0, hw - 1]]) < 1e-4
    again = model.op_conv2d(xd, w, None, stride=1, relu=False, tile_hint=hint).cpu().numpy()
    assert np.array_equal(got, again)


def test_attention_block_takes_clips_longer_than_4096_frames(pkg, oracle):
    """The attention block of the temporal branch on ONE clip of 4 200 frames (round 2 refused n > 4096 deep inside launch_tsattn,
    after the GRU had been enqueued): same kernels, the softmax row over the clip's frames is 4 200 floats of LDS.  Checked against the
    oracle; the limit that remains (32 768 frames per clip: 128 KB of LDS) is refused UP FRONT with a message that says what to do, by
    grnet_tsattn_forward and by grnet_gait_correct alike."""
    m = pkg.build_synthetic_model(max_frames=2, with_gru=True, with_tsattn=True, use_gait_feat=False)
    tsd = pkg.synth.make_tsattn_state_dict()
    x, xs = pkg.synth.make_tsattn_inputs(1, 4200)
    y = m.tsattn_forward(torch.from_numpy(x).cuda(), torch.from_numpy(xs).cuda()).cpu().numpy()
    ref = oracle.ts_attn_block(x, xs, tsd)
    assert y.shape == ref.shape and rel_err(y, ref) < 5e-5, rel_err(y, ref)
    with pytest.raises(pkg._lib.GrnetError, match="split the sequence into clips"):
        m.tsattn_forward(torch.zeros(1, 32769, 128, 24).cuda(), torch.zeros(1, 32769, 128, 25).cuda())
    m.close()
    mg = pkg.build_synthetic_model(max_frames=2, with_gru=True, use_gait_feat=True)
    t = 32769
    with pytest.raises(pkg._lib.GrnetError, match="split the sequence into clips"):
        mg.gait_correct(torch.zeros(t, 128, 24).cuda(), torch.zeros(t, 64, 24).cuda(), torch.zeros(t, 3).cuda(), torch.zeros(1, t, 4).cuda(),
                        torch.zeros(1, t, 2).cuda(), 1, t)
    mg.close()


@pytest.mark.parametrize("case", [(1, 224), (3, 224), (16, 224), (2, 64)], ids=lambda c: "x".join(map(str, c)))
def test_stem_kernel_with_flattened_reduction(model, oracle, case):
    """conv_stem_f32 (3 -> 64, 3x3, stride 2, K = (channel, tap) flattened to 7 k-steps, operands straight from global memory) vs the
    oracle's direct convolution: the path's 224 x 224 frames at 1 / 3 / 16 frames and a 64 x 64 map; bias + ReLU and the linear form,
    the first / last rows and columns separately (the top row and the left column read the zero padding), bit-identical repeats."""
    n, hw = case
    g = np.random.Generator(np.random.Philox(key=[93, n * 1000 + hw]))
    x = g.standard_normal((n, 3, hw, hw)).astype(np.float32)
    w = (g.standard_normal((64, 3, 3, 3)) * np.sqrt(2.0 / 27)).astype(np.float32)
    b = (g.standard_normal((64,)) * 0.1).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    ref = torch.relu(oracle.conv2d(x, w, stride=2, bias=b)).numpy()
    got = model.op_conv2d(xd, w, b, stride=2, relu=True, tile_hint=3001).cpu().numpy()
    assert got.shape == ref.shape == (n, 64, hw // 2, hw // 2)
    assert rel_err(got, ref) < 1e-5, rel_err(got, ref)
    lin = oracle.conv2d(x, w, stride=2).numpy()
    got = model.op_conv2d(xd, w, None, stride=2, relu=False, tile_hint=3001).cpu().numpy()
    assert rel_err(got, lin) < 1e-5
    e = hw // 2 - 1
    assert rel_err(got[:, :, [0, e]], lin[:, :, [0, e]]) < 1e-5 and rel_err(got[..., [0, e]], lin[..., [0, e]]) < 1e-5
    assert np.array_equal(got, model.op_conv2d(xd, w, None, stride=2, relu=False, tile_hint=3001).cpu().numpy())
    with pytest.raises(Exception):
        model.op_conv2d(torch.zeros(1, 4, 64, 64).cuda(), np.zeros((64, 4, 3, 3), np.float32), None, stride=2, relu=False, tile_hint=3001)


@pytest.mark.parametrize("case", [(1, 64, 256), (3, 64, 256), (16, 64, 256), (2, 64, 64), (16, 64, 64), (5, 64, 128), (16, 128, 25), (3, 128, 25)], ids=lambda c: "x".join(map(str, c)))
def test_pointwise_kernel_of_layer1(model, oracle, case):
    """conv_pw_f32 (layer1's 64 -> 256 1x1 convolutions and the PARE head's 128 -> 25 heat-map layer on 56 x 56 maps, both operands
    straight from global memory, weights resident in registers) vs the oracle: 64 -> 256 / 128 / 64 and 128 -> 25 (a partial last
    channel block) at 1 - 16 frames (long and short runs of tiles per wave); bias + ReLU, + residual, linear form, bit-identical repeats."""
    n, cin, cout = case
    g = np.random.Generator(np.random.Philox(key=[94, n * 100000 + cin * 10 + cout]))
    x = g.standard_normal((n, cin, 56, 56)).astype(np.float32)
    w = (g.standard_normal((cout, cin, 1, 1)) * np.sqrt(2.0 / cin)).astype(np.float32)
    b = (g.standard_normal((cout,)) * 0.1).astype(np.float32)
    r = g.standard_normal((n, cout, 56, 56)).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    conv = oracle.conv2d(x, w, stride=1, bias=b)
    got = model.op_conv2d(xd, w, b, stride=1, relu=True, tile_hint=3002).cpu().numpy()
    assert got.shape == conv.shape and rel_err(got, torch.relu(conv).numpy()) < 1e-5
    got = model.op_conv2d(xd, w, b, stride=1, relu=True, add=torch.from_numpy(r).cuda(), tile_hint=3002).cpu().numpy()
    assert rel_err(got, torch.relu(conv + torch.from_numpy(r)).numpy()) < 1e-5
    lin = oracle.conv2d(x, w, stride=1).numpy()
    got = model.op_conv2d(xd, w, None, stride=1, relu=False, tile_hint=3002).cpu().numpy()
    assert rel_err(got, lin) < 1e-5
    assert np.array_equal(got, model.op_conv2d(xd, w, None, stride=1, relu=False, tile_hint=3002).cpu().numpy())


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
