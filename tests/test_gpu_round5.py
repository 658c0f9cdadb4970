"""Round 5: the bf16 BasicBlock chain kernel (csrc/conv_bf16_chain.hip) -- the four BasicBlocks of an HR branch (lib/models/hrnet.py:30-59,
141-187) as ONE launch with the frame resident in LDS.

Bar (bf16 has no reference mode; stated as in test_gpu_bf16.py): every intermediate is rounded to bf16 exactly where the launch-per-
convolution path stores it, so the chain must equal (i) the fp32 oracle evaluated block by block on bf16-rounded operands with the
intermediates rounded to bf16, and (ii) the launch-per-convolution kernels, both up to fp32 summation order: a different order flips an
output rounding on ties, i.e. single elements differ by ONE bf16 ulp (2^-8 .. 2^-7 relative)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rb(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.fixture(scope="module")
def bmodel(pkg):
    m = pkg.build_synthetic_model(max_frames=64, with_gru=False, dtype="bf16")
    yield m
    m.close()


def _chain_weights(g, c, nconv):
    ws = [_rb(g.standard_normal((c, c, 3, 3)) * np.sqrt(2.0 / (c * 9))) for _ in range(nconv)]
    bs = [(g.standard_normal((c,)) * 0.1).astype(np.float32) for _ in range(nconv)]
    return ws, bs


def _oracle_chain(oracle, x, ws, bs):
    """BasicBlock by BasicBlock on the CPU: t = rb(relu(conv1(x) + b1)); x = rb(relu(conv2(t) + b2 + x))."""
    x = torch.from_numpy(x)
    for k in range(0, len(ws), 2):
        t = torch.from_numpy(_rb(torch.relu(oracle.conv2d(x.numpy(), ws[k], bias=bs[k])).numpy()))
        x = torch.from_numpy(_rb(torch.relu(oracle.conv2d(t.numpy(), ws[k + 1], bias=bs[k + 1]) + x).numpy()))
    return x.numpy()


def _close_up_to_rounding_ties(got, ref, max_mismatch):
    err = np.abs(got - ref)
    floor = 2e-3 * float(np.sqrt(np.mean(ref * ref)))
    assert np.all(err <= np.abs(ref) * 2.0 ** -7 + floor), float((err / (np.abs(ref) * 2.0 ** -7 + floor)).max())      # never more than one ulp
    frac = float(np.mean(err > np.abs(ref) * 2.0 ** -12 + 1e-6))
    assert frac <= max_mismatch, frac                                                                                # and only on a few elements
    return frac


@pytest.mark.parametrize("shape", [(64, 28), (128, 14), (256, 7)], ids=lambda s: f"{s[0]}ch{s[1]}")
@pytest.mark.parametrize("nconv", [2, 8])
def test_bf16_chain_equals_oracle_blocks(bmodel, oracle, shape, nconv):
    c, w = shape
    g = np.random.Generator(np.random.Philox(key=[85, c * 10 + nconv]))
    n = 3
    x = _rb(g.standard_normal((n, c, w, w)))
    ws, bs = _chain_weights(g, c, nconv)
    got = bmodel.op_conv_chain(torch.from_numpy(x).cuda(), ws, bs).cpu().numpy()
    assert got.shape == x.shape and np.array_equal(got, _rb(got))
    ref = _oracle_chain(oracle, x, ws, bs)
    frac = _close_up_to_rounding_ties(got, ref, 0.02 * nconv)
    print(f"chain {c}ch @{w} x{nconv}: {frac:.4f} of the elements off by a rounding tie")
    # the borders are where the zero halo of the LDS image is read: looked at separately
    for sl in (np.s_[:, :, 0], np.s_[:, :, -1], np.s_[:, :, :, 0], np.s_[:, :, :, -1]):
        _close_up_to_rounding_ties(got[sl], ref[sl], 0.05 * nconv)


@pytest.mark.parametrize("shape", [(64, 28), (128, 14), (256, 7)], ids=lambda s: f"{s[0]}ch{s[1]}")
def test_bf16_chain_equals_launch_per_convolution(bmodel, shape):
    """One BasicBlock: the chain launch against two launches of the per-convolution kernels on the same handle."""
    c, w = shape
    g = np.random.Generator(np.random.Philox(key=[86, c]))
    x = _rb(g.standard_normal((2, c, w, w)))
    ws, bs = _chain_weights(g, c, 2)
    xd = torch.from_numpy(x).cuda()
    t = bmodel.op_conv2d(xd, ws[0], bs[0], relu=True)
    ref = bmodel.op_conv2d(t, ws[1], bs[1], relu=True, add=xd).cpu().numpy()
    got = bmodel.op_conv_chain(xd, ws, bs).cpu().numpy()
    _close_up_to_rounding_ties(got, ref, 0.04)
    again = bmodel.op_conv_chain(xd, ws, bs).cpu().numpy()
    assert np.array_equal(got, again)                                    # deterministic


def test_bf16_chain_frames_are_independent(bmodel):
    """A workgroup is a frame: the same frame gives the same bits wherever it sits in the call, and a 70-frame call (more workgroups than a
    test usually launches) equals its frames one by one."""
    c, w = 128, 14
    g = np.random.Generator(np.random.Philox(key=[87, 1]))
    base = _rb(g.standard_normal((5, c, w, w)))
    ws, bs = _chain_weights(g, c, 4)
    x = torch.from_numpy(np.tile(base, (14, 1, 1, 1))).cuda()            # 70 frames
    out = bmodel.op_conv_chain(x, ws, bs)
    one = bmodel.op_conv_chain(x[:5].contiguous(), ws, bs)
    for k in range(14):
        assert torch.equal(out[5 * k:5 * k + 5], one)


def test_bf16_forward_with_chains_equals_forward_without(bmodel, pkg):
    """The whole bf16 forward at 64 frames (the chain launches are taken from 64 frames per call on) against the same forward with one
    launch per convolution (GRNET_OPT_BF16_CHAIN = 0): same network, same roundings, different summation order in 120 of its launches."""
    frames = torch.from_numpy(np.tile(pkg.synth.make_frames(8), (8, 1, 1, 1))).cuda()
    keys = ("features", "point_local_feat")
    with_chain = bmodel(frames, extras=keys)[-1]
    n_with = bmodel.num_kernel_launches()
    bmodel.set_option(pkg._lib.OPT_BF16_CHAIN, 0)
    try:
        without = bmodel(frames, extras=keys)[-1]
        n_without = bmodel.num_kernel_launches()
    finally:
        bmodel.set_option(pkg._lib.OPT_BF16_CHAIN, 7)
    torch.cuda.synchronize()
    assert n_without - n_with == 7 * (8 + 7 + 3), (n_with, n_without)    # 18 chains of 8 convolutions became 18 launches
    for k in keys + ("theta", "kp_3d", "verts"):
        a, b = with_chain[k].float().cpu().numpy(), without[k].float().cpu().numpy()
        rel = float(np.abs(a - b).max() / np.abs(b).max())
        print(k, rel)
        assert rel < (2e-2 if k in keys else 5e-3), (k, rel)             # bf16 rounding-tie noise through ~300 layers; the bf16 path's distance from fp32 is 1.7e-2
    th = with_chain["theta"].reshape(8, 8, 85)
    assert torch.equal(th[0], th[5])                                     # the 8 distinct frames repeat exactly


@pytest.mark.parametrize("shape", [(64, 28), (128, 14), (256, 7)], ids=lambda s: f"{s[0]}ch{s[1]}")
def test_bf16_chain_time_at_256_frames(pkg, shape):
    """Not a parity test: prints the duration of the 8-convolution chain launch at 256 frames (one frame per CU) next to eight launches of the
    per-convolution kernel (tools/bf16_micro.py measures those: 31.3 / 30.5 / 31.2 us each in round 4)."""
    c, w = shape
    m = pkg.build_synthetic_model(max_frames=4, with_gru=False, dtype="bf16")
    g = np.random.Generator(np.random.Philox(key=[88, c]))
    x = torch.from_numpy(_rb(g.standard_normal((256, c, w, w)))).cuda()
    ws, bs = _chain_weights(g, c, 8)
    out, us = m.op_conv_chain(x, ws, bs, reps=20)
    flops = 2.0 * 256 * w * w * c * c * 9 * 8
    print(f"\nconv_bf16_chain<{c},{w}> x8 at 256 frames: {us:.1f} us per launch = {us / 8:.2f} us per convolution, {flops / us / 1e6:.0f} TFLOP/s")
    assert torch.isfinite(out).all() and us > 0
    m.close()
