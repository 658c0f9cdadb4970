"""Host time of ONE eager forward's enqueue (292 launches on 4 streams + events) against the GPU time of the step: is the CPU ever the bound?
After a device sync the queues are empty, so the first calls return as fast as the host can enqueue: python tools/host_enqueue_time.py [--frames 16]"""
import argparse, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16)
a = ap.parse_args()
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
h = pkg.harness
m = pkg.build_synthetic_model(max_frames=a.frames, with_gru=False)
frames = torch.from_numpy(pkg.synth.make_frames(a.frames)).cuda()
r = h.ClipRunner(m, frames, use_graph=False, tune_level=1)
for _ in range(20):
    r.step()
torch.cuda.synchronize()
host = []
for rep in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print("per step: host enqueue ms / enqueue + drain ms:", ["%.2f / %.2f" % x for x in host])
t0 = time.perf_counter()
for _ in range(200):
    r.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"200 steps back to back: host returned after {(t1 - t0) * 5:.3f} ms per step, all done after {(t2 - t0) * 5:.3f} ms per step")
m.close()
