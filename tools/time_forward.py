"""Wall time of whole forwards (no parity check, no CPU leg): python tools/time_forward.py [--frames 16] [--steps 200] [--dtype f32] [--no-tune]
Used for timing-only ablations (GRNET_ABL_SKIP) whose results are garbage, and for quick A/B runs of environment switches."""
import argparse, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--no-tune", action="store_true")
ap.add_argument("--tag", default="")
a = ap.parse_args()
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.build_synthetic_model(max_frames=a.frames, with_gru=False, dtype=a.dtype)
x = torch.from_numpy(pkg.synth.make_frames(a.frames)).cuda().unsqueeze(0)
m(x); torch.cuda.synchronize()
if not a.no_tune:
    m.tune(a.frames)
for _ in range(20):
    m(x)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(a.steps):
        m(x)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / a.steps)
print(f"{a.tag} {a.frames} frames: {best * 1e3:.4f} ms/step  {a.frames / best:.0f} frames/s  {m.num_kernel_launches()} launches")
m.close()
