"""Calibration of rocprofv3's FETCH_SIZE on the conv kernels' own access pattern (MI355X_MICROARCH.md: "calibrate on a
known byte count in your own access pattern before trusting an absolute").  Each case is ONE convolution launch whose
input was evicted from L2 / Infinity Cache by a 1 GiB fill first; cases with a single output-channel block and no halo
(1x1) read exactly input + weights.
    on the GPU box:  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib -o fetch -- python3 tools/fetch_calib.py
    here:            python3 tools/fetch_calib.py --join
"""
import csv, importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 16
CASES = [  # cin, cout, k, stride, h, tile hint
    (64, 64, 1, 1, 56, 7), (64, 64, 1, 1, 56, 14), (64, 64, 1, 1, 56, 1071), (64, 64, 1, 1, 56, 1041),
    (256, 64, 1, 1, 56, 7), (256, 64, 1, 1, 56, 14), (256, 64, 1, 1, 56, 1071),
    (64, 64, 3, 1, 56, 14), (64, 64, 3, 1, 56, 7), (64, 64, 3, 1, 56, 1071),
    (480, 64, 3, 1, 56, 14), (480, 128, 3, 1, 56, 14), (480, 128, 3, 1, 56, 7),
    (64, 64, 3, 1, 28, 1071), (128, 128, 3, 1, 14, 1071), (256, 256, 3, 1, 7, 1041),
]


def run():
    import numpy as np, torch
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module("video-based-gait-analysis-for-dementia_amd")
    m = pkg.GRNet(max_frames=1)
    flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    for (cin, cout, k, s, h, hint) in CASES:
        x = torch.randn(N, cin, h, h, device="cuda")
        w = (np.random.randn(cout, cin, k, k) * 0.05).astype(np.float32)
        m.op_conv2d(x, w, None, stride=s, relu=True, tile_hint=hint)      # first call uploads the packed weights
        flush.fill_(1.0)
        torch.cuda.synchronize()
        m.op_conv2d(x, w, None, stride=s, relu=True, tile_hint=hint)      # the measured launch: the LAST conv dispatch of the case
        torch.cuda.synchronize()
        flush.fill_(2.0)
        torch.cuda.synchronize()


def join():
    rows = [r for r in csv.DictReader(open(os.path.join(ROOT, "gpurun_out", "calib", "fetch_counter_collection.csv")))]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    conv = [r for r in rows if "conv_" in r["Kernel_Name"]]
    assert len(conv) == 2 * len(CASES), len(conv)
    out = []
    for i, (cin, cout, k, s, h, hint) in enumerate(CASES):
        r = conv[2 * i + 1]
        raw = float(r["Counter_Value"]) * 1024
        alg = N * cin * h * h * 4 + k * k * cin * cout * 4
        out.append(dict(case=f"{cin}->{cout} k{k} s{s} {h}x{h}", hint=hint, kernel=r["Kernel_Name"].split("(")[0].replace("void grk::", ""),
                        input_plus_weights_MB=round(alg / 1e6, 3), fetch_raw_MB=round(raw / 1e6, 3), raw_over_alg=round(raw / alg, 3)))
        print(out[-1])
    json.dump(out, open(os.path.join(ROOT, "profiles", "r01_fetch_calibration.json"), "w"), indent=1)


if __name__ == "__main__":
    join() if "--join" in sys.argv else run()
