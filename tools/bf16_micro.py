"""Single bf16 convolutions at BASELINE configs[2]'s size (256 frames): us per launch and algorithmic TB/s (input + output (+ residual) once).
usage: python tools/bf16_micro.py [cin,cout,k,s,h,add ...]     (MICRO_N frames, default 256)"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.GRNet(max_frames=1, dtype="bf16")
N = int(os.environ.get("MICRO_N", "256"))
cases = [(32, 32, 3, 1, 56, 0), (32, 32, 3, 1, 56, 1), (64, 64, 3, 1, 28, 0), (64, 64, 3, 1, 28, 1), (128, 128, 3, 1, 14, 0), (128, 128, 3, 1, 14, 1),
         (256, 256, 3, 1, 7, 0), (256, 256, 3, 1, 7, 1), (64, 256, 1, 1, 56, 1), (256, 64, 1, 1, 56, 0), (64, 64, 3, 1, 56, 0), (128, 128, 3, 1, 56, 0),
         (256, 256, 3, 1, 56, 0), (480, 256, 3, 1, 56, 0), (32, 64, 3, 2, 56, 0), (64, 32, 1, 1, 28, 0)]
if len(sys.argv) > 1:
    cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
os.environ["GRNET_CONV_REPS"] = "30"
for (cin, cout, k, s, h, add) in cases:
    x = torch.randn(N, cin, h, h, device="cuda")
    w = (np.random.randn(cout, cin, k, k) * 0.05).astype(np.float32)
    r = torch.randn(N, cout, h // s, h // s, device="cuda") if add else None
    for dbg in [int(v) for v in os.environ.get("MICRO_DBG", "0").split(",")]:      # 2: no patch reads, 4: no stores (diagnostic build only)
        os.environ["GRNET_CONV_DBG"] = str(dbg)
        if dbg: print(f"   dbg {dbg}:", file=sys.stderr, end="")
        m.op_conv2d(x, w, None, stride=s, relu=True, add=r, tile_hint=int(os.environ.get("MICRO_HINT", "0")))
