"""Round-4 GPU parity tests: the fuse layer of every HR module (hrnet.py:189-244, 258-265) as the fp32 path launches it since round 4 --
the stride-2 chains run early as plain convolutions on the streams of the branches they start from, ONE grouped launch
(csrc/hr_fuse.hip) finishes outputs 0 .. nb-2 (all 1x1 "up" terms, the identity, the finished chains, ReLU) and one stride-2
convolution finishes the last output -- against the oracle's fuse layer fed the SAME branch outputs the GPU produced."""
import numpy as np
import pytest
import torch

from .conftest import rel_err

pytestmark = pytest.mark.gpu

MODULES = [("stage2", 0, 2)] + [("stage3", m, 3) for m in range(4)] + [("stage4", m, 4) for m in range(3)]


@pytest.fixture(scope="module")
def model(pkg):
    m = pkg.build_synthetic_model(max_frames=16, with_gru=False)
    yield m
    m.close()


@pytest.mark.parametrize("n", [1, 3, 16])
def test_fuse_layer_of_every_hr_module_matches_oracle(model, pkg, oracle, synth_weights, n):
    """For each of the 8 HR modules: the module's branch outputs x_b and outputs y_i are read back from the HIP forward
    (grnet_debug_tensor), the oracle's hr_fuse runs on those x_b with the same weights, and every y_i must agree to 2e-5 of its scale
    (fp32 sums re-associated: the grouped launch adds identity, chains, bias and up terms in its own order).  Covers stage 2 (one 1x1 term, one stride-2 convolution), stage 3 (3 + 3 incl. a two-convolution chain)
    and stage 4 (6 terms, chains of one / two / three stride-2 convolutions, the merged first convolution of chains (2,0) and (3,0));
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
    stage-4 chains (2,0) / (3,0) three more launches: 282 convolutions + 8 grouped launches, and the MACs still add up to SURVEY
    8(d)'s 15 441 563 648 per frame (the 1x1 terms are computed, not dropped)."""
    m = pkg.build_synthetic_model(max_frames=2, with_gru=False)
    convs = m.describe_convs()
    grouped = [c for c in convs if c["cin"] == 0]
    assert len(grouped) == 8 and all(c["name"].endswith("fuse_layers(up)") for c in grouped)
    assert len(convs) == m.num_conv_launches() == 290
    assert sum(c["macs"] for c in convs) == 15441563648
    assert not any(c["ks"] == 1 and "fuse_layers" in c["name"] and c["cin"] for c in convs)      # no separate 1x1 fuse launch is left
    m.close()
