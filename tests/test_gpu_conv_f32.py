"""fp32 convolution kernels against the oracle (lib/models/hrnet.py layers as the plan launches them): Winograd F(4x4,3x3) on 56x56 / 28x28 maps
(csrc/conv_wino4.hip: 4-wave and 8-wave forms, the half-size last round), on 14x14 / 7x7 maps (csrc/conv_wino4s.hip), the stem and layer1 1x1 kernels
(csrc/conv_stem.hip, csrc/conv_pw.hip), the grouped fuse launch of an HR module (csrc/hr_fuse.hip; hrnet.py:189-244, 258-265), and the whole forward with
the Winograd layers switched to the direct kernels.  (tests/test_gpu_parity.py holds the direct kernels and the end-to-end parity tests.)
Regrouped by component in round 6 from the per-round files; the tests themselves are unchanged."""
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

@pytest.mark.parametrize("case", [(1, 64, 64), (3, 128, 128), (2, 256, 256), (1, 480, 256), (3, 72, 192), (5, 64, 256), (16, 64, 256), (3, 32, 32), (2, 256, 32), (3, 40, 96)], ids=lambda c: "x".join(map(str, c)))
def test_winograd_f43_conv_kernel(model, oracle, case):
    """conv_wino4_f32 (Winograd F(4x4,3x3): 36 points per 4x4 output tile) on single convolutions vs the oracle's direct convolution:
    the layer shapes it is meant for and odd ones, 1-5 images (first / last tile rows carry the zero padding; 5 x 256 channels is
    280 workgroups); bias + ReLU, residual, and the plain linear form.  Its transforms carry the coefficients 4, 5, 2, 8: the bound
    here is 1e-4 of the output scale (measured ~1e-5), against 2e-5 for the F(2x2,3x3) kernel."""
    n, cin, cout = case
    g = np.random.Generator(np.random.Philox(key=[80, n * 100000 + cin * 1000 + cout]))
    x = g.standard_normal((n, cin, 56, 56)).astype(np.float32)
    w = (g.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
    b = (g.standard_normal((cout,)) * 0.1).astype(np.float32)
    r = g.standard_normal((n, cout, 56, 56)).astype(np.float32)
    conv = oracle.conv2d(x, w, stride=1, bias=b)
    xd = torch.from_numpy(x).cuda()
    got = model.op_conv2d(xd, w, b, stride=1, relu=True, tile_hint=2001).cpu().numpy()
    assert got.shape == conv.shape
    assert rel_err(got, torch.relu(conv).numpy()) < 1e-4, rel_err(got, torch.relu(conv).numpy())
    got = model.op_conv2d(xd, w, b, stride=1, relu=True, add=torch.from_numpy(r).cuda(), tile_hint=2001).cpu().numpy()
    assert rel_err(got, torch.relu(conv + torch.from_numpy(r)).numpy()) < 1e-4
    lin = oracle.conv2d(x, w, stride=1).numpy()
    got = model.op_conv2d(xd, w, None, stride=1, relu=False, tile_hint=2001).cpu().numpy()
    assert rel_err(got, lin) < 1e-4
    assert rel_err(got[:, :, [0, 55]], lin[:, :, [0, 55]]) < 1e-4 and rel_err(got[..., [0, 55]], lin[..., [0, 55]]) < 1e-4

@pytest.mark.parametrize("case", [(1, 128, 128), (3, 256, 256), (2, 64, 64), (3, 40, 96), (1, 32, 32), (20, 32, 256)], ids=lambda c: "x".join(map(str, c)))
def test_winograd_f43_conv_kernel_28(model, oracle, case):
    """The F(4x4,3x3) kernel on 28x28 maps: a workgroup's 14 tiles are two tile rows of 7, the image's 7 tile rows make 3.5 groups
    (the last group's lower half reads zeros and stores nothing); same cases as the F(2x2,3x3) kernel's 28x28 test."""
    n, cin, cout = case
    g = np.random.Generator(np.random.Philox(key=[81, n * 100000 + cin * 1000 + cout]))
    x = g.standard_normal((n, cin, 28, 28)).astype(np.float32)
    w = (g.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
    b = (g.standard_normal((cout,)) * 0.1).astype(np.float32)
    r = g.standard_normal((n, cout, 28, 28)).astype(np.float32)
    conv = oracle.conv2d(x, w, stride=1, bias=b)
    xd = torch.from_numpy(x).cuda()
    got = model.op_conv2d(xd, w, b, stride=1, relu=True, tile_hint=2001).cpu().numpy()
    assert got.shape == conv.shape
    assert rel_err(got, torch.relu(conv).numpy()) < 1e-4
    got = model.op_conv2d(xd, w, b, stride=1, relu=True, add=torch.from_numpy(r).cuda(), tile_hint=2001).cpu().numpy()
    assert rel_err(got, torch.relu(conv + torch.from_numpy(r)).numpy()) < 1e-4
    lin = oracle.conv2d(x, w, stride=1).numpy()
    got = model.op_conv2d(xd, w, None, stride=1, relu=False, tile_hint=2001).cpu().numpy()
    assert rel_err(got[:, :, [0, 27]], lin[:, :, [0, 27]]) < 1e-4 and rel_err(got[..., [0, 27]], lin[..., [0, 27]]) < 1e-4

@pytest.mark.parametrize("case", [(5, 32, 256), (16, 32, 256), (7, 40, 192), (6, 64, 64)], ids=lambda c: "x".join(map(str, c)))
def test_winograd_last_round_split(model, oracle, case):
    """Layers whose last round of workgroups is at most half full run it as half-size workgroups (the 32-channel kernel on the weights
    packed for the 64-channel one): 280 = 256 + 24 tiles (plain tile order), 896 = 768 + 128 (XCD-aware order, the 16-frame PARE
    layers), 294 = 256 + 38 with three channel blocks; 6 x 64 -> 64 (84 tiles) stays one launch.  With and without the residual."""
    n, cin, cout = case
    g = np.random.Generator(np.random.Philox(key=[78, n * 100000 + cin * 1000 + cout]))
    x = g.standard_normal((n, cin, 56, 56)).astype(np.float32)
    w = (g.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
    b = (g.standard_normal((cout,)) * 0.1).astype(np.float32)
    r = g.standard_normal((n, cout, 56, 56)).astype(np.float32)
    conv = oracle.conv2d(x, w, stride=1, bias=b)
    xd = torch.from_numpy(x).cuda()
    got = model.op_conv2d(xd, w, b, stride=1, relu=True, tile_hint=2001).cpu().numpy()
    assert rel_err(got, torch.relu(conv).numpy()) < 1e-4
    got = model.op_conv2d(xd, w, b, stride=1, relu=True, add=torch.from_numpy(r).cuda(), tile_hint=2001).cpu().numpy()
    assert rel_err(got, torch.relu(conv + torch.from_numpy(r)).numpy()) < 1e-4
    with pytest.raises(Exception):
        model.op_conv2d(torch.zeros(1, 64, 14, 14).cuda(), np.zeros((64, 64, 3, 3), np.float32), tile_hint=2001)   # not a 56x56 / 28x28 map: refused, no fallback

def test_winograd_layers_match_direct_layers(pkg, golden):
    """The whole forward with the eligible layers as Winograd (default) vs all-direct (GRNET_OPT_WINOGRAD = 0): same outputs to fp32
    re-association noise, both within the bar of the reference goldens."""
    m = pkg.build_synthetic_model(max_frames=16, with_gru=False)
    frames = torch.from_numpy(pkg.synth.make_frames(16)).cuda()
    m.set_option(pkg._lib.OPT_WINOGRAD, 1)
    a = {k: v.clone() for k, v in m(frames, extras=("features", "smpl_feats"))[-1].items()}
    m.set_option(pkg._lib.OPT_WINOGRAD, 0)
    b = {k: v.clone() for k, v in m(frames, extras=("features", "smpl_feats"))[-1].items()}
    torch.cuda.synchronize()
    assert not torch.equal(a["features"], b["features"])                    # the switch does select another kernel
    for k in ("features", "smpl_feats", "theta", "kp_3d", "kp_2d", "verts", "rotmat"):
        assert rel_err(a[k].cpu().numpy(), b[k].cpu().numpy()) < 5e-5, (k, rel_err(a[k].cpu().numpy(), b[k].cpu().numpy()))
    g = golden["grnet_n4"]
    for k in ("theta", "kp_3d", "kp_2d"):
        assert rel_err(a[k][0, :4].cpu().numpy().reshape(g[k].shape), g[k]) < 1e-4, k
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

MODULES = [("stage2", 0, 2)] + [("stage3", m, 3) for m in range(4)] + [("stage4", m, 4) for m in range(3)]

@pytest.mark.parametrize("n", [1, 3, 16])
def test_fuse_layer_of_every_hr_module_matches_oracle(model, pkg, oracle, synth_weights, n):
    """For each of the 8 HR modules: the module's branch outputs x_b and outputs y_i are read back from the HIP forward
    (grnet_debug_tensor), the oracle's hr_fuse runs on those x_b with the same weights, and every y_i must agree to 2e-5 of its scale
    (fp32 sums re-associated: the grouped launch adds identity, chains, bias and up terms in its own order).  Covers stage 2 (one 1x1 term, one stride-2 convolution), stage 3 (3 + 3 incl. a two-convolution chain)
    and stage 4 (6 terms, chains of one / two / three stride-2 convolutions, the merged first convolutions of the chains that start at one branch, linear and ReLU'd segments in one launch);
    1 / 3 / 16 frames."""
    frames = pkg.synth.make_frames(n)
    model(torch.from_numpy(frames).cuda().unsqueeze(0))
    torch.cuda.synchronize()
    for stage, m, nb in MODULES:
        tag = f"{stage}.{m}."
        xs = [model.debug_tensor(tag + f"x{b}", n).cpu() for b in range(nb)]
        for b in range(nb):
            assert xs[b].shape == (n, 32 << b, 56 >> b, 56 >> b)
        ref = oracle.hr_fuse(xs, synth_weights, f"backbone.{tag}")
        for i in range(nb):
            got = model.debug_tensor(tag + f"y{i}", n).cpu().numpy()
            e = rel_err(got, ref[i].numpy())
            assert got.shape == tuple(ref[i].shape) and e < 2e-5, (tag, i, e)
            assert got.min() >= 0.0                                     # the ReLU is applied exactly once, by the finishing launch

def test_fuse_layer_launch_count_and_macs(pkg):
    """The grouped fuse launch replaces 31 1x1 convolution launches and 8 elementwise sums; the merged first convolution of the
    stage-4 chains (2,0) / (3,0) three more launches: 272 convolutions + 8 grouped launches, and the MACs still add up to SURVEY
    8(d)'s 15 441 563 648 per frame (the 1x1 terms are computed, not dropped)."""
    m = pkg.build_synthetic_model(max_frames=2, with_gru=False)
    convs = m.describe_convs()
    grouped = [c for c in convs if c["cin"] == 0]
    assert len(grouped) == 8 and all(c["name"].endswith("fuse_layers(up)") for c in grouped)
    assert len(convs) == m.num_conv_launches() == 280
    assert sum(c["macs"] for c in convs) == 15441563648
    assert not any(c["ks"] == 1 and "fuse_layers" in c["name"] and c["cin"] for c in convs)      # no separate 1x1 fuse launch is left
    m.close()

@pytest.mark.parametrize("case", [(16, 480, 256, 56), (3, 256, 256, 56), (1, 128, 128, 56), (5, 32, 256, 56), (4, 48, 128, 56), (2, 64, 384, 56),
                                  (4, 64, 64, 56), (16, 64, 64, 28), (5, 256, 256, 28)],
                         ids=lambda c: "x".join(map(str, c)))
def test_wide_winograd_kernel_eight_waves(model, oracle, case):
    """conv_wino4w_f32 (F(4x4,3x3), eight waves per workgroup sharing one transformed chunk, 16-channel chunks, the two waves of a SIMD in
    opposite phases) vs the oracle's direct convolution.  128 output channels per workgroup: the three 56x56 layer shapes of the heads
    (480 -> 256 at 16 frames = 448 workgroups, 256 -> 256, 128 -> 128), two chunks (32 channels), an odd chunk count (48), three channel
    blocks (384).  The last three shapes run the eight-wave kernel only under GRNET_WINO_WIDE bits 1-3 (64 channels per workgroup, 28x28
    maps: measured and left off, conv_wino4_wide) -- by default they repeat the 4-wave kernel's check on the same data.  Bias + ReLU, + residual, the linear form with the borders looked at separately, bit-identical repeats, and agreement with
    the 4-wave kernel (hint 2003) to re-association noise.  Bound 1e-4 of the output scale as for the other F(4x4,3x3) kernels."""
    n, cin, cout, hw = case
    g = np.random.Generator(np.random.Philox(key=[94, n * 100000 + cin * 1000 + cout + hw]))
    x = g.standard_normal((n, cin, hw, hw)).astype(np.float32)
    w = (g.standard_normal((cout, cin, 3, 3)) * np.sqrt(2.0 / (cin * 9))).astype(np.float32)
    b = (g.standard_normal((cout,)) * 0.1).astype(np.float32)
    r = g.standard_normal((n, cout, hw, hw)).astype(np.float32)
    conv = oracle.conv2d(x, w, stride=1, bias=b)
    xd, rd = torch.from_numpy(x).cuda(), torch.from_numpy(r).cuda()
    got = model.op_conv2d(xd, w, b, stride=1, relu=True, tile_hint=2001).cpu().numpy()
    assert got.shape == conv.shape
    assert rel_err(got, torch.relu(conv).numpy()) < 1e-4, rel_err(got, torch.relu(conv).numpy())
    got_r = model.op_conv2d(xd, w, b, stride=1, relu=True, add=rd, tile_hint=2001).cpu().numpy()
    assert rel_err(got_r, torch.relu(conv + torch.from_numpy(r)).numpy()) < 1e-4
    old_r = model.op_conv2d(xd, w, b, stride=1, relu=True, add=rd, tile_hint=2003).cpu().numpy()
    assert rel_err(got_r, old_r) < 2e-6                                       # same transform arithmetic, the k order of the sums differs
    lin = oracle.conv2d(x, w, stride=1).numpy()
    got = model.op_conv2d(xd, w, None, stride=1, relu=False, tile_hint=2001).cpu().numpy()
    assert rel_err(got, lin) < 1e-4
    assert rel_err(got[:, :, [0, hw - 1]], lin[:, :, [0, hw - 1]]) < 1e-4 and rel_err(got[..., [0, hw - 1]], lin[..., [0, hw - 1]]) < 1e-4
    again = model.op_conv2d(xd, w, None, stride=1, relu=False, tile_hint=2001).cpu().numpy()
    assert np.array_equal(got, again)
