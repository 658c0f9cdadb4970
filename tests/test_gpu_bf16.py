"""bf16 path (grnet_create dtype = 1; BASELINE configs[2] / [4]: bf16 storage, fp32 accumulation on the bf16 matrix cores).

The reference has no bf16 mode, so the bar is stated here: (i) every conv launch equals the fp32 oracle evaluated on the SAME
bf16-rounded operands up to the one rounding of its bf16 output (2^-9 relative); (ii) over the whole network the distance from
the fp32 oracle is no larger than what an independent emulation of bf16 storage in the oracle shows (oracle.bf16_storage), and
the pose error stays in millimetres.  The 1e-3 bar of the north star belongs to the fp32 path (test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

from .conftest import rel_err

pytestmark = pytest.mark.gpu


def _rb(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.fixture(scope="module")
def bmodel(pkg):
    m = pkg.build_synthetic_model(max_frames=16, with_gru=False, dtype="bf16")
    yield m
    m.close()


CASES = [(3, 64, 3, 2, 224), (64, 64, 3, 2, 112), (64, 256, 1, 1, 56), (256, 64, 1, 1, 56), (32, 32, 3, 1, 56), (64, 64, 3, 1, 28),
         (128, 128, 3, 1, 14), (256, 256, 3, 1, 7), (32, 64, 3, 2, 56), (128, 256, 3, 2, 14), (256, 32, 1, 1, 7), (128, 25, 1, 1, 56),
         (480, 256, 3, 1, 56), (64, 64, 3, 1, 56), (32, 32, 3, 1, 28)]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("tile", [0, 7, 14])
def test_bf16_conv_kernel(bmodel, oracle, case, tile):
    cin, cout, k, stride, h = case
    g = np.random.Generator(np.random.Philox(key=[77, cin * 1000 + cout]))
    n = 3 if h <= 28 else 2
    x = _rb(g.standard_normal((n, cin, h, h)))
    w = _rb(g.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k)))
    b = (g.standard_normal((cout,)) * 0.1).astype(np.float32)
    ho = (h + 2 * (k // 2) - k) // stride + 1
    add = _rb(g.standard_normal((n, cout, ho, ho)))
    ref = torch.relu(oracle.conv2d(x, w, stride=stride, bias=b) + torch.from_numpy(add)).numpy()
    got = bmodel.op_conv2d(torch.from_numpy(x).cuda(), w, b, stride=stride, relu=True, add=torch.from_numpy(add).cuda(), tile_hint=tile).cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, _rb(got))                                 # the output IS bf16
    # products of bf16 operands are exact in fp32; what remains is fp32 summation order + ONE output rounding (half an ulp = 2^-9)
    assert np.all(np.abs(got - ref) <= np.abs(ref) * 2.0 ** -8 + 1e-5), float(np.abs(got - ref).max())


def test_bf16_forward_error_is_storage_rounding_noise(bmodel, pkg, oracle, synth_weights, synth_smpl):
    frames = pkg.synth.make_frames(4)
    keys = ("features", "part_attn", "smpl_feats", "point_local_feat")
    out = bmodel(torch.from_numpy(frames).cuda(), extras=keys)[-1]
    ref = oracle.grnet_forward(frames, synth_weights, synth_smpl, return_intermediates=True)
    with oracle.bf16_storage():
        emu = oracle.grnet_forward(frames, synth_weights, synth_smpl, return_intermediates=True)
    report = {}
    for k in keys + ("theta", "rotmat", "kp_3d", "kp_2d", "verts"):
        a = out[k].cpu().numpy()
        a = a[:, 1:] if k == "part_attn" else a
        r, e = np.asarray(ref[k]).reshape(a.shape), np.asarray(emu[k]).reshape(a.shape)
        report[k] = (rel_err(a, r), rel_err(e, r))
        # no further from fp32 than the emulation is -- as far as two different realisations of the same rounding noise can be compared through the MAX over four
        # frames: the GPU and the emulation round the same tensors at the same places, but sum in different orders, and one flipped bf16 rounding early in the
        # network moves the worst theta entry by tens of per cent (measured: 1.5-1.8 x the emulation's figure as unrelated kernels changed their last bits)
        assert report[k][0] < 2.0 * report[k][1] + 1e-3, (k, report[k])
    print(report)
    assert report["features"][0] < 4e-2 and report["point_local_feat"][0] < 1.5e-2
    d = out["kp_3d"].cpu().numpy().reshape(-1, 29, 3) - np.asarray(ref["kp_3d"]).reshape(-1, 29, 3)
    assert np.linalg.norm(d, axis=-1).mean() < 0.015                     # MPJPE vs the fp32 oracle: millimetres on a metre-sized body
    R = out["rotmat"].reshape(-1, 3, 3)
    assert (R @ R.transpose(1, 2) - torch.eye(3, device=R.device)).abs().max() < 1e-4   # the fp32 tail still returns rotations


@pytest.mark.parametrize("n", [64, 256])
def test_bf16_production_call_sizes_vs_oracle(pkg, oracle, synth_weights, synth_smpl, n):
    """The same bound as above at the call sizes the bf16 configs actually run (BASELINE configs[2] / [4]: 64-256 frames per call), where
    EVERY kernel group of GRNET_OPT_BF16_CHAIN is on (they start at 19-64 frames per call, include/grnet_hip.h): BasicBlock chains, wide
    bands, stride-2 bands, layer1's 1x1 pairs / Bottleneck launches, the fused stem -- and with them the plan logic that wires the launches
    together (fuse-layer merges, shifted addends, chain members).  8 distinct frames are tiled, so the oracle (grnet.py:129-175 outputs +
    the intermediates) and its bf16-storage emulation run on 8 frames only; every copy of a frame must give the same bits."""
    base = pkg.synth.make_frames(8)
    m = pkg.build_synthetic_model(max_frames=n, with_gru=False, dtype="bf16")
    try:
        frames = torch.from_numpy(np.tile(base, (n // 8, 1, 1, 1))).cuda()
        keys = ("features", "part_attn", "smpl_feats", "point_local_feat")
        out = m(frames, extras=keys)[-1]
        n_on = m.num_kernel_launches()
        m.set_option(pkg._lib.OPT_BF16_CHAIN, 0)
        m(frames[:n])
        n_off = m.num_kernel_launches()
        m.set_option(pkg._lib.OPT_BF16_CHAIN, -1)
        torch.cuda.synchronize()
        # the LDS-resident groups really ran: 18 chains of 8 convolutions are 18 launches, the 56x56 branch's 8 chains 4 launches each, layer1 and the stem fewer still
        assert n_off - n_on >= 7 * (8 + 7 + 3) + 4 * 8 + 3, (n_on, n_off)
        ref = oracle.grnet_forward(base, synth_weights, synth_smpl, return_intermediates=True)
        with oracle.bf16_storage():
            emu = oracle.grnet_forward(base, synth_weights, synth_smpl, return_intermediates=True)
        report = {}
        for k in keys + ("theta", "rotmat", "kp_3d", "kp_2d", "verts"):
            full = out[k].reshape(n // 8, 8, *out[k].shape[1:]) if out[k].shape[0] == n else out[k].reshape(n // 8, 8, *out[k].shape[2:])
            assert torch.equal(full[0], full[n // 8 - 1]) and torch.equal(full[0], full[(n // 8) // 2]), k      # a frame's result does not depend on its place in the call
            a = full[0].cpu().numpy()
            a = a[:, 1:] if k == "part_attn" else a
            r, e = np.asarray(ref[k]).reshape(a.shape), np.asarray(emu[k]).reshape(a.shape)
            report[k] = (rel_err(a, r), rel_err(e, r))
            assert report[k][0] < 2.0 * report[k][1] + 1e-3, (n, k, report[k])
        print(n, report)
        assert report["features"][0] < 4e-2 and report["point_local_feat"][0] < 1.5e-2
        d = out["kp_3d"].reshape(-1, 29, 3)[:8].cpu().numpy() - np.asarray(ref["kp_3d"]).reshape(-1, 29, 3)
        assert np.linalg.norm(d, axis=-1).mean() < 0.015
    finally:
        m.close()


def test_bf16_frames_are_independent_and_graph_equals_eager(bmodel, pkg):
    frames = torch.from_numpy(pkg.synth.make_frames(16)).cuda()
    full = bmodel(frames)[-1]
    one = bmodel(frames[5:6])[-1]
    part = bmodel(frames[8:11])[-1]
    torch.cuda.synchronize()
    for k in ("theta", "kp_3d", "verts", "rotmat"):
        assert torch.equal(full[k][0, 5], one[k][0, 0]), k                # same kernel per layer at every batch size: bit for bit
        assert torch.equal(full[k][0, 8:11], part[k][0]), k
    bmodel.set_option(pkg._lib.OPT_USE_GRAPH, 1)
    try:
        g1 = bmodel(frames)[-1]
        g2 = bmodel(frames)[-1]                                          # replay
        torch.cuda.synchronize()
        for k in ("theta", "verts"):
            assert torch.equal(g1[k], full[k]) and torch.equal(g2[k], full[k]), k
    finally:
        bmodel.set_option(pkg._lib.OPT_USE_GRAPH, 0)


def test_bf16_config3_shape(pkg):
    """BASELINE configs[2]: 8 clips x 32 frames in one call, bf16."""
    m = pkg.build_synthetic_model(max_frames=256, with_gru=False, dtype="bf16")
    base = pkg.synth.make_frames(4)
    frames = torch.from_numpy(np.tile(base, (64, 1, 1, 1))).cuda().reshape(8, 32, 3, 224, 224)
    out = m(frames)[-1]
    torch.cuda.synchronize()
    assert out["theta"].shape == (8, 32, 85) and out["verts"].shape == (8, 32, 6890, 3)
    th = out["theta"].reshape(64, 4, 85)
    assert torch.equal(th[0], th[37])                                    # the 4 distinct frames repeat exactly
    assert torch.isfinite(out["verts"]).all()
    m.close()


def test_bf16_is_a_separate_handle_dtype(pkg):
    with pytest.raises(ValueError):
        pkg.GRNet(max_frames=1, dtype="fp8")


@pytest.mark.parametrize("n", [1, 3])
def test_bf16_stem_kernel_reads_fp32_frames(bmodel, oracle, n):
    """conv_bf16_stem (round 4): the stem's 3 -> 64 stride-2 convolution straight from fp32 NCHW frames, K = (channel, tap) flattened to one
    32-wide MFMA k-step.  It rounds the frames to bf16 itself, so it must equal the fp32 oracle on bf16-rounded frames and weights up to
    the one rounding of its bf16 output -- and the generic kernel (which the conversion launch used to feed) on the same data likewise.
    The top row and the left column read the zero padding: looked at separately."""
    g = np.random.Generator(np.random.Philox(key=[79, n]))
    x32 = g.standard_normal((n, 3, 224, 224)).astype(np.float32)          # NOT pre-rounded: the kernel does the rounding
    w = _rb(g.standard_normal((64, 3, 3, 3)) * np.sqrt(2.0 / 27))
    b = (g.standard_normal((64,)) * 0.1).astype(np.float32)
    ref = torch.relu(oracle.conv2d(_rb(x32), w, stride=2, bias=b)).numpy()
    got = bmodel.op_conv2d(torch.from_numpy(x32).cuda(), w, b, stride=2, relu=True, tile_hint=3001).cpu().numpy()
    assert got.shape == ref.shape == (n, 64, 112, 112)
    assert np.array_equal(got, _rb(got))
    assert np.all(np.abs(got - ref) <= np.abs(ref) * 2.0 ** -8 + 1e-5), float(np.abs(got - ref).max())
    assert np.all(np.abs(got[:, :, 0] - ref[:, :, 0]) <= np.abs(ref[:, :, 0]) * 2.0 ** -8 + 1e-5) and np.all(np.abs(got[..., 0] - ref[..., 0]) <= np.abs(ref[..., 0]) * 2.0 ** -8 + 1e-5)
    lin = oracle.conv2d(_rb(x32), w, stride=2).numpy()
    got = bmodel.op_conv2d(torch.from_numpy(x32).cuda(), w, None, stride=2, relu=False, tile_hint=3001).cpu().numpy()
    assert np.all(np.abs(got - lin) <= np.abs(lin) * 2.0 ** -8 + 1e-5)
    gen = bmodel.op_conv2d(torch.from_numpy(_rb(x32)).cuda(), w, None, stride=2, relu=False, tile_hint=0).cpu().numpy()
    assert np.mean(got != gen) < 0.02                                      # the same sums in another order: they differ on output-rounding ties only


def test_bf16_forward_is_bit_stable_under_hbm_contention(pkg):
    """The band- and row-walking kernels overlap their own loads (LDS-DMA, register prefetch) with compute behind hand-counted or compiler-counted vmcnt
    waits (conv_bf16_block_frame: round-5 advice on a wait that assumed a store count; conv_bf16_bneck_dma: 3 x ND outstanding operations).  A wait that is one
    operation short shows only when memory is slow: here a copy kernel on a second stream keeps HBM busy while 64-frame forwards (every kernel group on) run, and
    every forward must reproduce the undisturbed one bit for bit."""
    n = 64
    m = pkg.build_synthetic_model(max_frames=n, with_gru=False, dtype="bf16")
    try:
        frames = torch.from_numpy(np.tile(pkg.synth.make_frames(8), (n // 8, 1, 1, 1))).cuda()
        ref = {k: v.clone() for k, v in m(frames)[-1].items()}
        torch.cuda.synchronize()
        big = torch.empty(1 << 27, dtype=torch.float32, device="cuda")
        big2 = torch.empty_like(big)
        side = torch.cuda.Stream()
        for it in range(24):
            with torch.cuda.stream(side):
                for _ in range(4):
                    big2.copy_(big)
            out = m(frames)[-1]
            torch.cuda.synchronize()
            for k in ("theta", "verts", "kp_3d", "rotmat"):
                assert torch.equal(out[k], ref[k]), (it, k)
    finally:
        m.close()
