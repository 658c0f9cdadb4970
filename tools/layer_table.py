#!/usr/bin/env python3
"""Per-layer table of the convolution launches of one 16-frame step: duration, TFLOP/s, and measured L2<->fabric
bytes next to the algorithmic bytes (input + fused addends + weights read once, output written once).

    on the GPU box (tools/gpu_layers.sh):  python3 tools/layer_table.py --dump gpurun_out/layers/convs.json
    here, after the run was merged back:   python3 tools/layer_table.py r01   -> profiles/r01_layer_table.{md,csv}

The join key is the dispatch order: with GRNET_MULTI_LANE=0 and no tuning the k-th convolution dispatch of every forward
is launch k of grnet_describe_conv().  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 and the factor is
cross-checked on the launches whose traffic is known exactly (single output-channel block, no halo: 1x1 convolutions).
"""
import collections
import csv
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "video-based-gait-analysis-for-dementia_amd"
N = 16


def dump(path):
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module(PKG)
    m = pkg.build_synthetic_model(max_frames=N, with_gru=False)
    json.dump(m.describe_convs(), open(path, "w"))
    m.close()


def conv_rows(path, value_col):
    rows = [r for r in csv.DictReader(open(path)) if "conv_" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(r["Kernel_Name"].split("(")[0].replace("void grk::", ""), value_col(r)) for r in rows]


def per_position(rows, n_conv):
    """rows of all forwards in dispatch order -> per launch position: (kernel name, mean over forwards, skipping the first)."""
    n_fw = len(rows) // n_conv
    assert n_fw >= 2 and len(rows) == n_fw * n_conv, (len(rows), n_conv)
    out = []
    for k in range(n_conv):
        vals = [rows[f * n_conv + k][1] for f in range(1, n_fw)]
        names = {rows[f * n_conv + k][0] for f in range(n_fw)}
        assert len(names) == 1, names
        out.append((names.pop(), sum(vals) / len(vals)))
    return out


def fetch_factor(kernel, c):
    """True bytes per counted byte of FETCH_SIZE for one launch.  Measured on launches with exactly known reads
    (tools/fetch_calib.py, profiles/*_fetch_calibration.json): the counter reads  bytes x (1/2 + 128 B / segment)  where
    `segment` is the contiguous run one channel of a tile is staged from with 16-byte LDS-DMA (448 B -> 0.79, 896 B -> 0.64,
    1344 B -> 0.60 measured; long streams -> the guide's 1/2), and reads the bytes exactly for the dword-per-lane staging
    of the gather / planes modes (like the scalar loads of bilinear2x_kernel)."""
    args = [a.strip() for a in kernel[kernel.index("<") + 1:kernel.rindex(">")].split(",")]
    rows = args[0] in ("true", "1")
    if not rows:
        return 1.0
    tps, s, ks = int(args[3]), c["stride"], c["ks"]
    r = min(c["hout"], max(1, tps * 16 // c["wout"]))
    seg = ((r - 1) * s + ks) * c["win"] * 4
    return 1.0 / (0.5 + 128.0 / seg)


def stage_of(name):
    for pre, tag in (("backbone.conv", "stem"), ("backbone.layer1", "layer1"), ("backbone.transition", "transition"),
                     ("backbone.stage2", "stage2"), ("backbone.stage3", "stage3"), ("backbone.stage4", "stage4"),
                     ("backbone.upsample", "upsample heads"), ("head.", "PARE head")):
        if name.startswith(pre):
            return tag
    return "other"


def main(rnd):
    src = os.path.join(ROOT, "gpurun_out", "layers")
    convs = json.load(open(os.path.join(src, "convs.json")))
    nc = len(convs)
    dur = per_position(conv_rows(os.path.join(src, "trace_kernel_trace.csv"),
                                 lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3), nc)
    fetch = per_position(conv_rows(os.path.join(src, "FETCH_SIZE_counter_collection.csv"), lambda r: float(r["Counter_Value"]) * 1024), nc)
    write = per_position(conv_rows(os.path.join(src, "WRITE_SIZE_counter_collection.csv"), lambda r: float(r["Counter_Value"]) * 1024), nc)
    table = []
    for k, c in enumerate(convs):
        in_b = N * c["cin"] * c["hin"] * c["win"] * 4
        out_b = N * c["cout"] * c["hout"] * c["wout"] * 4
        add_b = N * c["add_elems"] * 4
        w_b = c["ks"] * c["ks"] * c["cin"] * c["cout"] * 4
        flop = 2.0 * N * c["cout"] * c["hout"] * c["wout"] * c["cin"] * c["ks"] ** 2
        table.append(dict(pos=k, name=c["name"], stage=stage_of(c["name"]), shape=f'{c["cin"]}->{c["cout"]} k{c["ks"]} s{c["stride"]} {c["hin"]}x{c["win"]}',
                          n_add=c["n_add"], kernel=dur[k][0], us=dur[k][1], tflops=flop / dur[k][1] / 1e6, gflop=flop / 1e9,
                          alg_read=in_b + add_b + w_b, alg_write=out_b, fetch_raw=fetch[k][1], write=write[k][1],
                          fetch_est=fetch[k][1] * fetch_factor(dur[k][0], c)))
    # calibration of the FETCH_SIZE factor on launches with exactly known reads: 1x1, stride 1, one output-channel block
    cal = [t for t, c in zip(table, convs) if c["ks"] == 1 and c["cout"] <= 64 and c["hin"] >= 28]
    factor = sum(t["alg_read"] for t in cal) / max(1.0, sum(t["fetch_raw"] for t in cal))
    dst = os.path.join(ROOT, "profiles")
    with open(os.path.join(dst, f"{rnd}_layer_table.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(table[0].keys()))
        w.writeheader()
        for t in table:
            w.writerow({k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()})
    tot = lambda key, rows=table: sum(t[key] for t in rows)
    lines = [f"# {rnd}: convolution launches of one 16-frame fp32 step, one after another (GRNET_MULTI_LANE=0, cost-model configurations)", "",
             f"{nc} launches, {tot('us') / 1e3:.3f} ms serial, {tot('gflop') / tot('us') * 1e3:.1f} TFLOP/s average; "
             f"algorithmic bytes {tot('alg_read') / 1e9:.3f} GB read + {tot('alg_write') / 1e9:.3f} GB written; measured FETCH_SIZE x2 "
             f"{2 * tot('fetch_raw') / 1e9:.3f} GB (upper bound), with the per-launch calibration of fetch_factor() {tot('fetch_est') / 1e9:.3f} GB, "
             f"WRITE_SIZE {tot('write') / 1e9:.3f} GB.", "",
             f"FETCH_SIZE calibration on {len(cal)} launches with exactly known reads (1x1, <= 64 output channels, one channel block, no halo): "
             f"algorithmic read bytes / raw FETCH_SIZE = {factor:.2f} (the guide prescribes x2 for 16 B/lane streaming reads).", "",
             "## by stage", "", "| stage | launches | ms | TFLOP/s | alg. read MB | FETCHx2 MB | ratio | calibrated MB | ratio | alg. write MB | WRITE MB |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    by = collections.OrderedDict()
    for t in table:
        by.setdefault(t["stage"], []).append(t)
    for st, rows in by.items():
        lines.append(f"| {st} | {len(rows)} | {tot('us', rows) / 1e3:.3f} | {tot('gflop', rows) / tot('us', rows) * 1e3:.1f} | {tot('alg_read', rows) / 1e6:.1f} | "
                     f"{2 * tot('fetch_raw', rows) / 1e6:.1f} | {2 * tot('fetch_raw', rows) / tot('alg_read', rows):.2f} | {tot('fetch_est', rows) / 1e6:.1f} | "
                     f"{tot('fetch_est', rows) / tot('alg_read', rows):.2f} | {tot('alg_write', rows) / 1e6:.1f} | {tot('write', rows) / 1e6:.1f} |")
    lines += ["", "## by kernel instantiation", "", "| kernel | launches | ms | TFLOP/s | FETCHx2 / alg. read |", "|---|---|---|---|---|"]
    byk = collections.OrderedDict()
    for t in sorted(table, key=lambda t: t["kernel"]):
        byk.setdefault(t["kernel"], []).append(t)
    for kn, rows in sorted(byk.items(), key=lambda kv: -tot("us", kv[1])):
        lines.append(f"| `{kn}` | {len(rows)} | {tot('us', rows) / 1e3:.3f} | {tot('gflop', rows) / tot('us', rows) * 1e3:.1f} | {2 * tot('fetch_raw', rows) / tot('alg_read', rows):.2f} |")
    lines += ["", "## the 25 longest launches", "", "| pos | weight key | shape | addends | kernel | us | TFLOP/s | alg. read MB | FETCHx2 MB | alg. write MB | WRITE MB |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for t in sorted(table, key=lambda t: -t["us"])[:25]:
        lines.append(f"| {t['pos']} | {t['name']} | {t['shape']} | {t['n_add']} | `{t['kernel']}` | {t['us']:.1f} | {t['tflops']:.1f} | {t['alg_read'] / 1e6:.1f} | "
                     f"{2 * t['fetch_raw'] / 1e6:.1f} | {t['alg_write'] / 1e6:.1f} | {t['write'] / 1e6:.1f} |")
    open(os.path.join(dst, f"{rnd}_layer_table.md"), "w").write("\n".join(lines) + "\n")
    json.dump({"launches": nc, "serial_ms": tot("us") / 1e3, "algorithmic_read_bytes": tot("alg_read"), "algorithmic_write_bytes": tot("alg_write"),
               "fetch_size_x2_bytes": 2 * tot("fetch_raw"), "fetch_calibrated_bytes": tot("fetch_est"), "write_size_bytes": tot("write"),
               "calibration": "FETCH_SIZE = bytes x (1/2 + 128 B / contiguous staged segment) for the 16-byte LDS-DMA row staging, x1 for dword "
                              "staging; measured on launches with exactly known reads, tools/fetch_calib.py"},
              open(os.path.join(dst, f"{rnd}_layer_traffic.json"), "w"), indent=1)
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--dump":
        dump(sys.argv[2])
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else "r01")
