"""Winograd kernel vs the direct kernel on the eligible layer shapes: us per launch (HIP events inside grnet_op_conv2d, GRNET_CONV_REPS)."""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GRNET_CONV_REPS"] = "30"
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.build_synthetic_model(max_frames=2, with_gru=False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = ((480, 256), (256, 256), (128, 128), (64, 64), (256, 32), (32, 32)) if "GRNET_CONV_DBG" not in os.environ else ((480, 256),)
for cin, cout in shapes:
    x = torch.randn(n, cin, 56, 56, device="cuda")
    w = (np.random.randn(cout, cin, 3, 3) * 0.02).astype(np.float32)
    b = np.zeros(cout, np.float32)
    for hint in (0, 2000):
        m.op_conv2d(x, w, b, relu=True, tile_hint=hint)
# the BasicBlock form of the 56x56 HR branch: residual addend, same shape as the output
for cin, cout in ((32, 32), (64, 64)):
    x = torch.randn(n, cin, 56, 56, device="cuda")
    r = torch.randn(n, cout, 56, 56, device="cuda")
    w = (np.random.randn(cout, cin, 3, 3) * 0.02).astype(np.float32)
    print(f"[with residual] {cin}->{cout}", file=sys.stderr)
    for hint in (0, 2000):
        m.op_conv2d(x, w, np.zeros(cout, np.float32), relu=True, add=r, tile_hint=hint)
# 28x28 maps (upsample heads)
for cin, cout in ((256, 256), (128, 128), (64, 64)):
    x = torch.randn(n, cin, 28, 28, device="cuda")
    w = (np.random.randn(cout, cin, 3, 3) * 0.02).astype(np.float32)
    for hint in (0, 2000):
        m.op_conv2d(x, w, np.zeros(cout, np.float32), relu=True, tile_hint=hint)
# F(4x4,3x3) on the widest layers (hint 2001)
for cin, cout in ((480, 256), (256, 256), (128, 128), (64, 64), (32, 32), (256, 32)):
    x = torch.randn(n, cin, 56, 56, device="cuda")
    w = (np.random.randn(cout, cin, 3, 3) * 0.02).astype(np.float32)
    for hint in (2000, 2001):
        m.op_conv2d(x, w, np.zeros(cout, np.float32), relu=True, tile_hint=hint)
# F(4x4,3x3) on 28x28 maps
for cin, cout in ((256, 256), (128, 128), (64, 64)):
    x = torch.randn(n, cin, 28, 28, device="cuda")
    w = (np.random.randn(cout, cin, 3, 3) * 0.02).astype(np.float32)
    for hint in (2000, 2001):
        m.op_conv2d(x, w, np.zeros(cout, np.float32), relu=True, tile_hint=hint)
