"""Round 6: the stem pair (lib/models/hrnet.py:470-476) and each layer1 Bottleneck (hrnet.py:62-100) as ONE row-walking launch on the bf16 path
(csrc/conv_bf16_roll.hip): every intermediate (the 64 ch @112x112 stem tensor; a Bottleneck's two 64-channel tensors) stays in LDS.

Bar (as for every bf16 launch, tests/test_gpu_bf16.py): each launch, fed the GPU's OWN input tensor, equals the fp32 oracle evaluated on bf16-rounded
operands with every intermediate rounded to bf16 where the launch-per-convolution path stores it (oracle.bf16_storage) -- up to fp32 summation order,
i.e. single elements one bf16 ulp apart on rounding ties -- and the whole forward equals the forward with these launches switched off likewise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROLL_BITS = 256 + 512


def _rb(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def _close_up_to_ties(got, ref, what, depth=2, floor=2.0 ** -6):
    """An element is off by at most one bf16 ulp of itself or of the summands that nearly cancel in it (floor: 2^-6 of the tensor's rms behind `depth`
    rounded stages; 2^-5 where TWELVE rounded stages lie between the two evaluations, whose ties compound), only a few elements are off at all, and the mean error is noise."""
    err = np.abs(got - ref)
    rms = float(np.sqrt(np.mean(ref * ref)))
    bound = np.abs(ref) * 2.0 ** -7 + floor * rms
    assert np.all(err <= bound), (what, float((err / bound).max()))
    frac = float(np.mean(err > np.abs(ref) * 2.0 ** -12 + 1e-6))
    assert frac <= 0.05 * depth, (what, frac)
    assert float(err.mean()) <= 2e-3 * rms * depth ** 0.5, (what, float(err.mean() / rms))
    return frac


@pytest.mark.parametrize("n", [64, 128], ids=["4_segments", "2_segments"])
def test_roll_launches_equal_oracle_on_their_own_inputs(pkg, oracle, synth_weights, n):
    """64 frames: a workgroup is a quarter of a frame (14 rows + one ring row beyond each inner boundary); 128 frames: half a frame; 256 frames (one workgroup per
    frame) is the size of tests/test_gpu_bf16.py::test_bf16_production_call_sizes_vs_oracle.  The borders (the 3x3's zero rows -1 and 56, the zero columns, the stem's
    padding row and column) are looked at separately."""
    m = pkg.build_synthetic_model(max_frames=n, with_gru=False, dtype="bf16")
    try:
        base = pkg.synth.make_frames(8)
        frames = torch.from_numpy(np.tile(base, (n // 8, 1, 1, 1))).cuda()
        m(frames)
        n_on = m.num_kernel_launches()
        torch.cuda.synchronize()
        names = ["stem_conv2"] + [f"layer1.{k}" for k in range(4)]
        got = {}
        for name in names:
            t = m.debug_tensor(name, n).cpu().numpy()
            assert np.array_equal(t[:8], t[n - 8:]) and np.array_equal(t[:8], t[n // 2:n // 2 + 8]), name      # a frame's result does not depend on its place in the call
            got[name] = t[:8]
            assert np.array_equal(got[name], _rb(got[name]))
        sd = synth_weights
        with oracle.bf16_storage():
            x = torch.from_numpy(_rb(base))
            s1 = oracle.conv_bn(x, sd, "backbone.conv1.weight", "backbone.bn1", stride=2, relu=True)
            ref = {"stem_conv2": oracle.conv_bn(s1, sd, "backbone.conv2.weight", "backbone.bn2", stride=2, relu=True).numpy()}
            prev = "stem_conv2"
            for k in range(4):
                ref[f"layer1.{k}"] = oracle.bottleneck(torch.from_numpy(got[prev]), sd, f"backbone.layer1.{k}.", k == 0).numpy()
                prev = f"layer1.{k}"
        for name in names:
            g, r = got[name], ref[name]
            assert g.shape == r.shape, (name, g.shape, r.shape)
            frac = _close_up_to_ties(g, r, name, depth=3)
            for sl in (np.s_[:, :, 0], np.s_[:, :, -1], np.s_[:, :, :, 0], np.s_[:, :, :, -1], np.s_[:, :, 13:15], np.s_[:, :, 27:29]):      # image borders and the segment boundaries
                _close_up_to_ties(g[sl], r[sl], name + " border", depth=6)
            print(f"{name}: {frac:.4f} of the elements off by a rounding tie")
        # the same forward with these launches off (layer1 then runs its 1x1 pairs / stream launches, the stem its two launches)
        m.set_option(pkg._lib.OPT_BF16_CHAIN, 1023 - ROLL_BITS)
        m(frames)
        n_off = m.num_kernel_launches()
        torch.cuda.synchronize()
        assert n_off - n_on == 5 + 1, (n_on, n_off)                          # layer1: 9 launches -> 4; stem: 2 -> 1
        for name in ("stem_conv2", "layer1"):
            other = m.debug_tensor(name, n).cpu().numpy()[:8]
            mine = got[name if name != "layer1" else "layer1.3"]
            _close_up_to_ties(mine, other, name + " vs launch-per-convolution", depth=2 if name == "stem_conv2" else 12, floor=2.0 ** -6 if name == "stem_conv2" else 2.0 ** -5)
    finally:
        m.close()


def test_roll_launches_are_refused_where_they_do_not_apply(pkg):
    """Small calls and forced tiles keep the launch-per-convolution kernels (a workgroup of the row-walking launches is at least a quarter of a frame: below 64 frames
    the chip would be mostly idle); GRNET_OPT_BF16_MIN_FRAMES = 1 takes them at any size, and the result then equals the small-call path up to rounding ties."""
    m = pkg.build_synthetic_model(max_frames=16, with_gru=False, dtype="bf16")
    try:
        frames = torch.from_numpy(pkg.synth.make_frames(16)).cuda()
        m(frames)
        n_small = m.num_kernel_launches()
        small = m.debug_tensor("layer1", 16).cpu().numpy()
        m.set_option(pkg._lib.OPT_BF16_CHAIN, 1023 - ROLL_BITS)
        m(frames)
        assert m.num_kernel_launches() == n_small                             # the bits change nothing at 16 frames
        m.set_option(pkg._lib.OPT_BF16_CHAIN, ROLL_BITS)                      # ONLY the row-walking launches ...
        m.set_option(pkg._lib.OPT_BF16_MIN_FRAMES, 1)                         # ... at any call size
        m(frames)
        assert n_small - m.num_kernel_launches() == 12 - 4 + 1                # (no pairs at 16 frames: layer1 was 12 launches)
        forced = m.debug_tensor("layer1", 16).cpu().numpy()
        _close_up_to_ties(forced, small, "layer1 forced at 16 frames", depth=12, floor=2.0 ** -5)
        m.set_option(pkg._lib.OPT_CONV_TILE, 7)                               # a forced tile switches every special kernel off
        m(frames)
        assert m.num_kernel_launches() >= n_small
    finally:
        m.close()
