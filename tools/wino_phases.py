"""Per-phase ticks of the F(4x4,3x3) workgroups of conv_wino4_f32 (diagnostic build: make ABLATION=1 BUILD=build_abl LIB=../libgrnet_hip_abl.so;
GRNET_LIB_PATH=.../libgrnet_hip_abl.so GRNET_W4_PHASES=1).  The first launch of a shape is cold: read the LAST line per shape."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.GRNet(max_frames=1)
N = int(os.environ.get("MICRO_N", "16"))
for (c, h, add) in [(32, 56, 0), (32, 56, 1), (64, 28, 0), (64, 28, 1), (64, 56, 0), (256, 56, 0)]:
    x = torch.randn(N, c, h, h, device="cuda")
    w = (np.random.randn(c, c, 3, 3) * 0.05).astype(np.float32)
    r = torch.randn(N, c, h, h, device="cuda") if add else None
    for _ in range(3):
        m.op_conv2d(x, w, None, stride=1, relu=True, add=r, tile_hint=2001)
