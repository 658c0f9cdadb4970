#!/usr/bin/env python3
"""Golden vectors for row f2 (the pose-feature corrector and the use_gait_feat branch of GRNet.forward), produced by RUNNING the
reference's own code (lib/models/layers/feature_correction.py:18-157, lib/models/grnet.py:154-173) in this container.

The reference class does not construct as shipped: its __init__ / forward read module-level names that are defined nowhere
(SURVEY 0.3).  This script binds exactly those names in the imported module's namespace -- the "repair by specification"
DESIGN.md records -- and changes nothing else:

    temporal_encode = "none", gf_mode = "", cparam_mode = ""   stored lower-cased, never read again
    use_pe = initialize_h = spatial_smask = leff_fc_in = False   stored, never read again
    use_leff = leff_smpl_feats = False                            forward(): skips a branch that only unpacks x_smplf's shape
    N = n                                                         forward() :144 reshapes y to (b, N, 128, -1); y has n frames

Everything else (stubs for absent third-party modules, the synthetic weights) is tools/make_goldens.py's.  Only input/output
tensors are written: tests/golden/featcorr.npz.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as mg  # noqa: E402

ROOT, synth, netspec = mg.ROOT, mg.synth, mg.netspec


def main():
    import torch
    torch.manual_seed(0)
    tmp, stubs, sd, smpl = mg.setup_workdir()
    os.chdir(tmp)
    sys.path[:0] = [stubs, mg.REF]
    pare_sd = {"model." + k: torch.from_numpy(np.asarray(v)) for k, v in sd.items() if k.startswith("head.")}
    torch.save({"state_dict": pare_sd}, "data/grnet_data/pare_w_3dpw_checkpoint.ckpt")

    import lib.models.layers.feature_correction as fc
    for name, val in (("temporal_encode", "none"), ("gf_mode", ""), ("cparam_mode", ""), ("use_pe", False), ("initialize_h", False),
                      ("spatial_smask", False), ("use_leff", False), ("leff_smpl_feats", False), ("leff_fc_in", False)):
        assert not hasattr(fc, name), f"{name} is defined after all"
        setattr(fc, name, val)

    # the weights are drawn by reference key name: use the names of a MAX-GRNet checkpoint (pfeat_corrector.*) for the standalone module too
    fsd = {k[len("pfeat_corrector."):]: v for k, v in synth.make_featcorr_state_dict(prefix="pfeat_corrector.").items()}
    corr = fc.FeatCorrector(x_size=128, num_avg_gfeat=3, seqlen=100, num_layers=1, estim_phase=True, num_joints=24, h_size=1024,
                            num_transformer_head=4, use_jwff=True).eval()
    spec = netspec.featcorr_spec("")
    assert list(corr.state_dict().keys()) == list(spec.keys()), set(corr.state_dict()) ^ set(spec)
    for k, (shape, _) in spec.items():
        assert tuple(corr.state_dict()[k].shape) == tuple(shape), (k, corr.state_dict()[k].shape, shape)
    corr.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in fsd.items()}, strict=True)
    print(f"FeatCorrector state_dict matches netspec.featcorr_spec: {len(spec)} tensors, h_size {corr.h_size}")
    out = {}
    for (b, n) in ((2, 8), (1, 16)):
        x, cp = synth.make_featcorr_inputs(b, n)
        fc.N = n
        with torch.no_grad():
            y, avg, ph = corr(torch.from_numpy(x), cparams=torch.from_numpy(cp))
        out[f"y_{b}_{n}"], out[f"avg_{b}_{n}"], out[f"phase_{b}_{n}"] = y.numpy(), avg.numpy(), ph.numpy()
        print(f"featcorr b{b} n{n}: y {tuple(y.shape)} absmax {float(y.abs().max()):.3f}  x absmax {np.abs(x).max():.3f}")

    # --- the whole use_gait_feat branch of GRNet.forward on 4 frames -------------------------------------------
    from lib.models.grnet import GRNet
    GRNet.is_demo = True
    cfg = types.SimpleNamespace(AVG_DIM=3, ESTIM_PHASE=True, NUM_LAYERS=1, H_SIZE=1024, NUM_HEADS=4, USE_JWFF=True)
    model = GRNet(writer=None, seqlen=100, use_gait_feat=True, featcorr=cfg).eval()
    full = dict(sd)
    full.update(synth.make_featcorr_state_dict(prefix="pfeat_corrector."))
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in full.items()}, strict=False)
    assert not unexpected and all(k.startswith("regressor.") for k in missing), (missing[:5], unexpected[:5])
    b, n = 1, 4
    frames = torch.from_numpy(synth.make_frames(n)).reshape(b, n, 3, 224, 224)
    bbox, cimg = synth.make_gait_boxes(b, n)
    fc.N = n
    with torch.no_grad():
        res = model(frames, bbox=torch.from_numpy(bbox), cimg=torch.from_numpy(cimg))[-1]
    for k in ("theta", "kp_3d", "kp_2d", "rotmat", "pred_cparam", "pred_avg", "pred_phase"):
        out["gait_" + k] = res[k].numpy()
    out["gait_verts_s5"] = res["verts"].numpy()[:, :, ::5]
    print("gait branch keys:", sorted(res.keys()))
    p = os.path.join(ROOT, "tests/golden/featcorr.npz")
    np.savez_compressed(p, **out)
    print(f"wrote {p}")


if __name__ == "__main__":
    main()
