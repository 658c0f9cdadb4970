"""The Winograd F(4x4,3x3) kernels vs the direct kernel on the eligible layer shapes: us per launch (HIP events inside grnet_op_conv2d,
GRNET_CONV_REPS).  Hint 2001 = conv_wino4_f32 / conv_wino4w_f32 (56x56 / 28x28 maps; the 8-wave kernel where the output channels come in 128s on
56x56 maps), 2003 = the 4-wave kernel everywhere, 2020 = conv_wino4s_f32 (14x14 / 7x7 maps)."""
import importlib, os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GRNET_CONV_REPS", "30")
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.build_synthetic_model(max_frames=2, with_gru=False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for hw, hint, shapes in ((56, 2001, ((480, 256), (256, 256), (128, 128), (64, 64), (256, 32), (32, 32))), (28, 2001, ((256, 256), (128, 128), (64, 64))),
                         (14, 2020, ((128, 128), (256, 256))), (7, 2020, ((256, 256),))):
    for cin, cout in shapes:
        x = torch.randn(n, cin, hw, hw, device="cuda")
        w = (np.random.randn(cout, cin, 3, 3) * 0.02).astype(np.float32)
        r = torch.randn(n, cout, hw, hw, device="cuda") if cin == cout else None
        for h in ((0, hint, 2003) if hint == 2001 and (cout % 128 == 0 or cout == 64) else (0, hint)):
            m.op_conv2d(x, w, np.zeros(cout, np.float32), relu=True, add=r, tile_hint=h)
