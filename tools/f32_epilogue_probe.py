"""Ablation of the whole-K epilogue (diagnostic build): dbg 8 = half of the tile stores, dbg 16 = none."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.GRNet(max_frames=1)
x = torch.randn(16, 32, 56, 56, device="cuda")
w = (np.random.randn(32, 32, 3, 3) * 0.05).astype(np.float32)
add = torch.randn(16, 32, 56, 56, device="cuda")
for dbg in (0, 8, 16):
    os.environ["GRNET_CONV_DBG"] = str(dbg)
    print("dbg", dbg, file=sys.stderr)
    m.op_conv2d(x, w, None, stride=1, relu=True, add=add, tile_hint=14)
