"""Times of the temporal modules (GRU gait encoder, attention block, whole feature corrector) on the GPU at hand.
GRNET_GRU_SPLIT=0 switches the recurrence back to one workgroup per (sequence, direction) for A/B runs."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.build_synthetic_model(max_frames=2, use_gait_feat=True)
sizes = [(1, 16), (8, 32), (1, 450), (4, 64), (1, 1250), (1, 10000)]
for (b, t) in sizes:
    x = torch.randn(b, t, 3072, device="cuda"); cp = torch.randn(b, t, 3, device="cuda")
    xx = torch.randn(b, t, 128, 24, device="cuda"); xs = torch.randn(b, t, 128, 25, device="cuda")
    fns = [("gru", lambda: m.gru_forward(x, cp))]
    if t <= 4096:
        fns.append(("tsattn", lambda: m.tsattn_forward(xx, xs)))
    for name, fn in fns:
        fn(); torch.cuda.synchronize()
        reps = 10 if t < 5000 else 3
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / reps
        print(f"{name:7s} (b,T)=({b},{t}): {ms:9.3f} ms" + (f"   {ms * 1e3 / (2 * t):.2f} us per step and layer" if name == "gru" else ""), flush=True)
