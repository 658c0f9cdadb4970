"""Per-kernel times of the conv-class launches of one forward, each layer shape launched ALONE (GRNet.kernel_table): python tools/kernel_alone.py [frames] [dtype] [filter]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dtype = sys.argv[2] if len(sys.argv) > 2 else "f32"
flt = sys.argv[3] if len(sys.argv) > 3 else ""
m = pkg.build_synthetic_model(max_frames=n, device_id=0, with_gru=False, dtype=dtype)
m.finalize()
m.tune(n, level=1)
for r in m.kernel_table(n, reps=50):
    if flt in r["name"]:
        print(f"{r['name']:40s} {r['launches']:4d} x {r['avg_us']:8.2f} us = {r['total_us']:9.1f} us")
