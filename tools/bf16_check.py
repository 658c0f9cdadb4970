"""bf16 path bring-up: conv kernel vs the fp32 oracle on bf16-rounded operands, then the whole forward vs the fp32 oracle."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
import oracle.grnet_oracle as oracle

def rb(a):  # round to bf16 (nearest even), back to f32
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()

m = pkg.GRNet(max_frames=1, dtype="bf16")
if "--forward" in sys.argv:
    cases_skip = True
else:
    cases_skip = False
g = np.random.default_rng(0)
cases = [(32, 32, 3, 1, 56), (64, 64, 3, 1, 28), (3, 64, 3, 2, 224), (64, 64, 3, 2, 112), (64, 256, 1, 1, 56), (256, 64, 1, 1, 56), (128, 128, 3, 1, 14),
         (256, 256, 3, 1, 7), (128, 25, 1, 1, 56), (480, 256, 3, 1, 56), (128, 256, 3, 2, 14), (256, 32, 1, 1, 7), (32, 64, 3, 2, 56)]
for (cin, cout, k, s, h) in ([] if cases_skip else cases):
    n = 3 if h <= 28 else 2
    x = rb(g.standard_normal((n, cin, h, h)))
    w = rb(g.standard_normal((cout, cin, k, k)) * np.sqrt(2.0 / (cin * k * k)))
    b = (g.standard_normal((cout,)) * 0.1).astype(np.float32)
    ho = (h + 2 * (k // 2) - k) // s + 1
    add = rb(g.standard_normal((n, cout, ho, ho)))
    ref = torch.relu(oracle.conv2d(x, w, stride=s, bias=b) + torch.from_numpy(add)).numpy()
    for hint in (0, 7, 14):
        try:
            got = m.op_conv2d(torch.from_numpy(x).cuda(), w, b, stride=s, relu=True, add=torch.from_numpy(add).cuda(), tile_hint=hint).cpu().numpy()
        except Exception as e:
            print((cin, cout, k, s, h), hint, "ERR", str(e)[:80]); continue
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        print((cin, cout, k, s, h), "hint", hint, "rel err %.2e" % err, "OK" if err < 6e-3 else "BAD")
m.close()
if "--forward" in sys.argv:
    mb = pkg.build_synthetic_model(max_frames=4, with_gru=False, dtype="bf16")
    frames = pkg.synth.make_frames(4)
    out = mb(torch.from_numpy(frames).cuda(), extras=("features", "part_attn", "smpl_feats", "point_local_feat"))[-1]
    ref = oracle.grnet_forward(frames, pkg.synth.make_state_dict(), pkg.synth.make_smpl_tables(), return_intermediates=True)
    for k in ("features", "part_attn", "smpl_feats", "point_local_feat", "theta", "rotmat", "kp_3d", "kp_2d", "verts"):
        a, r = out[k].cpu().numpy(), np.asarray(ref[k])
        if k == "part_attn":
            a = a[:, 1:]
        print(k, a.shape, "rel err %.3e" % (np.abs(a - r.reshape(a.shape)).max() / np.abs(r).max()))
    d = out["kp_3d"].cpu().numpy().reshape(-1, 29, 3) - np.asarray(ref["kp_3d"]).reshape(-1, 29, 3)
    print("MPJPE (m)", float(np.linalg.norm(d, axis=-1).mean()))
    with oracle.bf16_storage():
        emu = oracle.grnet_forward(frames, pkg.synth.make_state_dict(), pkg.synth.make_smpl_tables(), return_intermediates=True)
    print("--- vs the oracle with bf16 storage emulated")
    for k in ("features", "part_attn", "smpl_feats", "point_local_feat", "theta", "rotmat", "kp_3d", "kp_2d", "verts"):
        a, r = out[k].cpu().numpy(), np.asarray(emu[k])
        if k == "part_attn":
            a = a[:, 1:]
        print(k, "rel err %.3e" % (np.abs(a - r.reshape(a.shape)).max() / np.abs(r).max()), " emu-vs-fp32 %.3e" % (np.abs(np.asarray(ref[k]) - r).max() / np.abs(r).max()))
    d = out["kp_3d"].cpu().numpy().reshape(-1, 29, 3) - np.asarray(emu["kp_3d"]).reshape(-1, 29, 3)
    print("MPJPE vs emulation (m)", float(np.linalg.norm(d, axis=-1).mean()))
