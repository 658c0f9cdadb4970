import importlib, sys, time, torch
sys.path.insert(0, "/root/repo")
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.build_synthetic_model(max_frames=2, with_gru=True, with_tsattn=True)
for (b, t) in ((1, 16), (8, 32), (1, 450), (4, 64)):
    x = torch.randn(b, t, 3072, device="cuda"); cp = torch.randn(b, t, 3, device="cuda")
    xx = torch.randn(b, t, 128, 24, device="cuda"); xs = torch.randn(b, t, 128, 25, device="cuda")
    for name, fn in (("gru", lambda: m.gru_forward(x, cp)), ("tsattn", lambda: m.tsattn_forward(xx, xs))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        print(name, (b, t), "%.3f ms" % ((time.perf_counter() - t0) * 100))
