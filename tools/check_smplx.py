"""Pin the SMPL linear-blend skinning of this repo to the REAL smplx package, the first time its assets exist (rows a7 / f3 are
"parity unpinned" until then: smplx==0.1.26 and SMPL_NEUTRAL.pkl are neither installed nor vendored offline, requirements.txt:13).

    python tools/check_smplx.py [--smpl-dir data/smpl_data] [--n 64]

If `import smplx` works and <smpl-dir>/SMPL_NEUTRAL.pkl (+ J_regressor_extra.npy, lib/models/smpl.py:8-10) is found, it
  1. builds smplx.SMPL(model_path=smpl_dir, create_transl=False) as lib/models/smpl.py:111-113 does and runs it with pose2rot=False on
     n random (betas, rotation matrices) -- the call of SMPLHead.forward (smpl.py:157-165);
  2. feeds the SAME tables (v_template, shapedirs, posedirs, J_regressor, lbs weights, parents, J_regressor_extra) and inputs to
     oracle.smpl_lbs / oracle.smpl_joints29 (the CPU restatement) and, on a GPU box, to grnet_load_smpl + grnet_smpl_forward;
  3. reports max |difference| of vertices and of the 29 "spin2" joints against smplx; exit code 1 above 1e-4 m (fp32 re-association is ~1e-6).
Skips cleanly (exit code 0, says what is missing) otherwise.  Test infrastructure: imports oracle/."""
import argparse
import importlib
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "video-based-gait-analysis-for-dementia_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--smpl-dir", default="data/smpl_data")
    ap.add_argument("--n", type=int, default=64)
    ap.add_argument("--tol", type=float, default=1e-4)
    a = ap.parse_args()
    try:
        import smplx
    except ImportError:
        print("check_smplx: SKIPPED -- the smplx package is not installed (pip install smplx==0.1.26 where a network exists)")
        return 0
    pkl, jx = os.path.join(a.smpl_dir, "SMPL_NEUTRAL.pkl"), os.path.join(a.smpl_dir, "J_regressor_extra.npy")
    missing = [p for p in (pkl, jx) if not os.path.isfile(p)]
    if missing:
        print("check_smplx: SKIPPED -- missing " + ", ".join(missing) + " (the SMPL model files are licensed assets, smpl.py:8-10)")
        return 0
    import torch
    oracle = importlib.import_module("oracle.grnet_oracle")
    pkg = importlib.import_module(PKG)
    ref_model = smplx.SMPL(model_path=a.smpl_dir, create_transl=False)
    with open(pkl, "rb") as f:
        raw = pickle.load(f, encoding="latin1")
    dense = lambda m: np.asarray(m.todense() if hasattr(m, "todense") else m)
    tables = {"v_template": np.asarray(raw["v_template"], np.float32), "shapedirs": np.asarray(raw["shapedirs"], np.float32)[:, :, :10],
              "posedirs": np.asarray(raw["posedirs"], np.float32).reshape(6890 * 3, 207).T.copy(),
              "J_regressor": dense(raw["J_regressor"]).astype(np.float32), "lbs_weights": np.asarray(raw["weights"], np.float32),
              "parents": np.asarray(raw["kintree_table"][0], np.int64).copy(), "J_regressor_extra": np.load(jx).astype(np.float32)}
    tables["parents"][0] = -1
    g = np.random.Generator(np.random.Philox(key=[77, a.n]))
    betas = (g.standard_normal((a.n, 10)) * 1.5).astype(np.float32)
    rot6d = g.standard_normal((a.n * 24, 6)).astype(np.float32)
    rotmat = oracle.rot6d_to_rotmat(rot6d).reshape(a.n, 24, 3, 3).astype(np.float32)
    with torch.no_grad():
        out = ref_model(betas=torch.from_numpy(betas), body_pose=torch.from_numpy(rotmat[:, 1:]), global_orient=torch.from_numpy(rotmat[:, :1]), pose2rot=False)
    v_ref, j24_ref = out.vertices.numpy(), out.joints.numpy()[:, :24]
    worst = 0.0
    v_or, j24_or = oracle.smpl_lbs(betas, rotmat, tables)
    for name, got, ref in (("oracle verts", v_or, v_ref), ("oracle joints", j24_or, j24_ref)):
        d = float(np.abs(np.asarray(got) - ref).max())
        worst = max(worst, d)
        print(f"check_smplx: {name:14s} max |diff| vs smplx {d:.3e} m")
    if torch.cuda.is_available():
        m = pkg.GRNet(max_frames=a.n)
        m.load_state_dict(pkg.synth.make_state_dict(), strict=True)        # the conv weights do not matter to this leg; the handle needs them to finalize
        m.load_smpl(tables)                                                # grnet_load_smpl with the REAL tables
        m.finalize()
        v_gpu, kp3d, _ = m.smpl_forward(torch.from_numpy(betas), torch.from_numpy(rotmat))
        d = float(np.abs(v_gpu.cpu().numpy() - v_ref).max())
        worst = max(worst, d)
        print(f"check_smplx: HIP verts      max |diff| vs smplx {d:.3e} m")
        j29 = oracle.smpl_joints29(v_ref, j24_ref, tables)
        d = float(np.abs(kp3d.cpu().numpy() - np.asarray(j29)).max())
        worst = max(worst, d)
        print(f"check_smplx: HIP 29 joints  max |diff| vs the 29 joints regressed from smplx's vertices {d:.3e} m")
        m.close()
    else:
        print("check_smplx: no GPU here -- the HIP leg (grnet_smpl_forward) runs on a GPU box")
    ok = worst <= a.tol
    print(f"check_smplx: {'PINNED' if ok else 'MISMATCH'} (worst {worst:.3e} m, tolerance {a.tol:g})")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
