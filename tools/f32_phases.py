"""Per-phase ticks of the fp32 split-K and whole-K workgroups (diagnostic build: make -C .../csrc clean all ABLATION=1; GRNET_F32_PHASES=1)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.GRNet(max_frames=1)
N = int(os.environ.get("MICRO_N", "16"))
for (cin, cout, k, s, h, hint) in [(64, 64, 3, 1, 28, 1171), (64, 64, 3, 1, 28, 1071), (128, 128, 3, 1, 14, 1171), (256, 256, 3, 1, 7, 1141), (32, 32, 3, 1, 56, 1071),
                                   (128, 32, 1, 1, 14, 1041), (64, 128, 3, 2, 28, 1071),
                                   (32, 32, 3, 1, 56, 14), (64, 64, 3, 1, 56, 7), (64, 256, 1, 1, 56, 7), (256, 64, 1, 1, 56, 7), (128, 128, 3, 1, 56, 14),
                                   (256, 256, 3, 1, 56, 14), (3, 64, 3, 2, 224, 7)]:
    x = torch.randn(N, cin, h, h, device="cuda")
    w = (np.random.randn(cout, cin, k, k) * 0.05).astype(np.float32)
    add = torch.randn(N, cout, h // s, h // s, device="cuda")
    m.op_conv2d(x, w, None, stride=s, relu=True, add=add, tile_hint=hint)
