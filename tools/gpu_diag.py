"""Ad-hoc GPU diagnostics (not a test): where does batch-vs-single differ?"""
import importlib, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
m = pkg.build_synthetic_model(max_frames=16, with_gru=False)
frames = torch.from_numpy(pkg.synth.make_frames(16)).cuda()
names = ["stem_conv1", "stem_conv2", "layer1", "stage2.0", "stage2.1", "stage3.0", "stage3.1", "stage3.2",
         "stage4.0", "stage4.1", "stage4.2", "stage4.3", "up2.0.bilinear", "up2.0.conv", "up3.0.bilinear", "up3.0.conv",
         "up3.1.bilinear", "up3.1.conv", "up4.0.bilinear", "up4.0.conv", "up4.1.bilinear", "up4.1.conv", "up4.2.bilinear", "up4.2.conv"]
m(frames)
full = {k: m.debug_tensor(k, 16).clone() for k in names}
m(frames[5:6])
one = {k: m.debug_tensor(k, 1).clone() for k in names}
perm = torch.cat([frames[5:6], frames[:5], frames[6:]])
m(perm)
first = {k: m.debug_tensor(k, 16).clone() for k in names}
torch.cuda.synchronize()
for k in names:
    d = (full[k][5] - one[k][0]).abs(); d2 = (first[k][0] - one[k][0]).abs()
    print(f"{k:16s} {tuple(full[k].shape[1:])!s:16s} pos5-vs-single max {float(d.max()):.3e} ndiff {int((d>0).sum()):8d} | pos0-in-batch-vs-single max {float(d2.max()):.3e}")
