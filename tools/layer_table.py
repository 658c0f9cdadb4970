#!/usr/bin/env python3
"""Per-launch table of the convolution-class launches of one 16-frame fp32 step: duration one after another, TFLOP/s by the algorithmic
(direct-convolution) count and by the multiplies the kernel executes, counter bytes (FETCH_SIZE x 2 + WRITE_SIZE, the guide's
correction) next to the algorithmic bytes (input + fused addends + weights read once, output written once).

    on the GPU box (tools/gpu_profile.sh):   python3 tools/layer_table.py --dump gpurun_out/layers/convs.json
    here, after the run was merged back:     python3 tools/layer_table.py r04   -> profiles/r04_layer_table.{md,csv}, r04_layer_traffic.json
                                             (bench.py reads counter_over_algorithmic_by_kernel from the latter for roofline.dominant_kernel)

Join key: dispatch order.  With GRNET_MULTI_LANE=0, --no-graph and --tune-level 0 the conv_* / hr_fuse_up_* dispatches of every forward
come in the order of grnet_describe_conv(); a layer whose last round runs as half-size workgroups is two dispatches (the second one is
conv_wino4_f32<2, W, 0, true>) and is merged into one row."""
import collections
import csv
import importlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = "video-based-gait-analysis-for-dementia_amd"
sys.path.insert(0, ROOT)
accounting = importlib.import_module(PKG + ".accounting")
N = 16


def dump(path, dtype="f32"):
    sys.path.insert(0, ROOT)
    pkg = importlib.import_module(PKG)
    import ctypes as C
    n = 256 if dtype == "bf16" else N                    # the lane scheduler orders the launches for max_frames: the SAME value as the profiled run
    m = pkg.build_synthetic_model(max_frames=n, with_gru=False, dtype=dtype)
    convs = m.describe_convs()
    for pos, c in enumerate(convs):                      # the kernel family<shape> name bench.py's kernel table uses
        name = C.create_string_buffer(96)
        m._lib.grnet_conv_kernel_info(m._h, pos, n, name, 96, None)
        c["kernel_family"] = name.value.decode()
        c["dispatches"] = 0 if c["kernel_family"].endswith("+") else 1      # (rounds 5: 4 for the 56x56 branch's chain, one launch per BasicBlock; round 6: conv_bf16_chain_pipe, one launch)
    json.dump(convs, open(path, "w"))
    m.close()


def short(name):
    name = re.sub(r"void grk::\(anonymous namespace\)::|void grk::|grk::\(anonymous namespace\)::", "", name)
    return name.split("(")[0]


def conv_dispatches(path, value):
    rows = [r for r in csv.DictReader(open(path)) if "conv_" in r["Kernel_Name"] or "hr_fuse_up" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = []
    for r in rows:
        k = short(r["Kernel_Name"])
        if re.match(r"conv_wino4_f32<2, \d+, 0, true>", k) and out:           # the half-size last round of the previous layer
            out[-1] = (out[-1][0], out[-1][1] + value(r), out[-1][2] + 1)
        else:
            out.append((k, value(r), 1))
    return out


def per_position(rows, convs):
    """Dispatch list (all forwards of the run, in order) -> per convolution of the plan: (kernel, value averaged over the forwards but the first,
    dispatches).  A plan entry owns c["dispatches"] consecutive dispatches: 1 normally, 0 for a convolution that runs inside a bf16 chain launch
    (its value is 0: the chain's first member carries the launch), 4 for the 56x56 branch's chain (one launch per BasicBlock)."""
    per_fw = sum(c.get("dispatches", 1) for c in convs)
    n_fw = len(rows) // per_fw
    assert n_fw >= 2 and len(rows) == n_fw * per_fw, (len(rows), per_fw, "dispatches do not tile into forwards")
    out, base = [], 0
    for c in convs:
        nd = c.get("dispatches", 1)
        if nd == 0:
            out.append((c.get("kernel_family", "?"), 0.0, 0))
            continue
        names = {rows[f * per_fw + base][0] for f in range(n_fw)}
        assert len(names) == 1, (c["name"], names)
        vals = [sum(rows[f * per_fw + base + j][1] for j in range(nd)) for f in range(1, n_fw)]
        out.append((names.pop(), sum(vals) / len(vals), sum(rows[base + j][2] for j in range(nd))))
        base += nd
    return out


def executed_ratio(kernel, c):
    if kernel.startswith("conv_wino4s"):
        return 0.25 * (256.0 / 196.0 if c["hin"] == 14 else 64.0 / 49.0)
    if kernel.startswith("conv_wino4"):
        return 0.25
    return 1.0


def stage_of(name):
    for key, st in (("backbone.conv", "stem"), ("backbone.layer1", "layer1"), ("transition1", "transition1"), ("stage2", "stage2"), ("transition2", "stage3"),
                    ("stage3", "stage3"), ("transition3", "stage4"), ("stage4", "stage4"), ("upsample", "upsample heads"), ("head.", "PARE head")):
        if key in name:
            return st
    return "other"


def main(rnd, dtype="f32"):
    global N
    bf = dtype == "bf16"
    if bf:
        N = 256                                          # BASELINE configs[2]: 8 clips x 32 frames per call
    esz, peak = (2.0, 2500.0) if bf else (4.0, 157.3)
    src = os.path.join(ROOT, "gpurun_out", "bf16") if bf else os.path.join(ROOT, "gpurun_out")
    convs = json.load(open(os.path.join(src, "layers", "convs.json")))
    for c in convs:
        if c.get("dispatches", 1) > 1: c["dispatches"] = 1
    n_conv = len(convs)
    dur = per_position(conv_dispatches(os.path.join(src, "prof_serial", "bench_kernel_trace.csv"), lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3), convs)
    fetch = per_position(conv_dispatches(os.path.join(src, "pmc", "FETCH_SIZE_counter_collection.csv"), lambda r: float(r["Counter_Value"]) * 1024.0), convs)
    write = per_position(conv_dispatches(os.path.join(src, "pmc", "WRITE_SIZE_counter_collection.csv"), lambda r: float(r["Counter_Value"]) * 1024.0), convs)
    rows, by_stage, by_kernel, by_family = [], collections.OrderedDict(), collections.OrderedDict(), collections.OrderedDict()
    open_launch = {}                                         # kernel family -> (row index, table keys) of its latest launch that carries fused layers
    for c, (k, us, nd), (_, fb, _), (_, wb, _) in zip(convs, dur, fetch, write):
        flop = 2.0 * N * c["macs"]
        alg_r, alg_w = accounting.conv_algorithmic_bytes(c, N, esz)      # the ONE definition (bench.py uses the same)
        ex = executed_ratio(k, c)
        if nd == 0:                                          # runs inside a chain / pair launch (family name + "+"): its FLOPs and per-layer bytes count THERE.  The
            ri, keys = open_launch[c["kernel_family"].rstrip("+")]          # plan interleaves the branches of a module, so the launch is found by family, not by position
            r = rows[ri]
            r["gflop"] += flop / 1e9; r["alg_mb"] += (alg_r + alg_w) / 1e6; r["fused_layers"] += 1
            r["tflops"] = r["exec_tflops"] = r["gflop"] / r["us"] * 1e3
            for key, table in keys:
                t = table[key]
                t["gflop"] += flop / 1e9; t["ex"] += flop * ex / 1e9; t["alg"] += (alg_r + alg_w) / 1e6
            continue
        rows.append(dict(name=c["name"], kernel=k, dispatches=nd, fused_layers=1, shape=f'{c["cin"]}->{c["cout"]} k{c["ks"]} s{c["stride"]} @{c["hin"]}', us=us, gflop=flop / 1e9,
                         tflops=flop / us / 1e6, exec_tflops=flop * ex / us / 1e6, alg_mb=(alg_r + alg_w) / 1e6, counter_mb=(2 * fb + wb) / 1e6))
        chain_keys = ((stage_of(c["name"]), by_stage), (re.sub(r"<.*", "", k) + " " + rows[-1]["shape"], by_kernel), (c.get("kernel_family", k), by_family))
        open_launch[c.get("kernel_family", k)] = (len(rows) - 1, chain_keys)
        for key, table in chain_keys:
            t = table.setdefault(key, dict(n=0, us=0.0, gflop=0.0, ex=0.0, alg=0.0, cnt=0.0))
            t["n"] += 1; t["us"] += us; t["gflop"] += flop / 1e9; t["ex"] += flop * ex / 1e9; t["alg"] += (alg_r + alg_w) / 1e6; t["cnt"] += (2 * fb + wb) / 1e6
    dst = os.path.join(ROOT, "profiles")
    with open(os.path.join(dst, f"{rnd}_layer_table.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0]))
        w.writeheader()
        w.writerows(rows)
    tot = dict(n=len(rows), us=sum(r["us"] for r in rows), gflop=sum(r["gflop"] for r in rows), alg=sum(r["alg_mb"] for r in rows), cnt=sum(r["counter_mb"] for r in rows))
    lines = [f"# Per-launch table of the {len(rows)} convolution-class launches ({n_conv} layers; a bf16 BasicBlock chain is one row) of a {N}-frame {dtype} step ({rnd}), launched one after another (GRNET_MULTI_LANE=0)", "",
             f"Total: {tot['us'] / 1e3:.3f} ms, {tot['gflop']:.1f} algorithmic GFLOP = {tot['gflop'] / tot['us'] * 1e3:.1f} TFLOP/s; counter bytes (FETCH_SIZE x 2 + WRITE_SIZE) "
             f"{tot['cnt'] / 1e3:.2f} GB vs {tot['alg'] / 1e3:.2f} GB algorithmic ({tot['cnt'] / tot['alg']:.2f} x).",
             f"Peak of the {dtype} matrix cores: {peak} TFLOP/s.  `exec` = the multiplies the kernel issues (fp32: F(4x4,3x3) at 1/4 of the direct count, x 1.31 on the padded 14x14 / 7x7 maps; bf16: the direct count).", "",
             "## By stage", "", "| stage | launches | ms | algorithmic TFLOP/s | executed TFLOP/s | counter MB | algorithmic MB |", "|---|---|---|---|---|---|---|"]
    for k, t in by_stage.items():
        lines.append(f"| {k} | {t['n']} | {t['us'] / 1e3:.3f} | {t['gflop'] / t['us'] * 1e3:.1f} | {t['ex'] / t['us'] * 1e3:.1f} | {t['cnt']:.0f} | {t['alg']:.0f} |")
    lines += ["", "## By kernel and shape", "", f"| kernel, shape | launches | us each | ms | algorithmic TFLOP/s | executed TFLOP/s (of {peak}) | counter / algorithmic bytes |", "|---|---|---|---|---|---|---|"]
    for k, t in sorted(by_kernel.items(), key=lambda kv: -kv[1]["us"]):
        lines.append(f"| {k} | {t['n']} | {t['us'] / t['n']:.1f} | {t['us'] / 1e3:.3f} | {t['gflop'] / t['us'] * 1e3:.1f} | {t['ex'] / t['us'] * 1e3:.1f} ({t['ex'] / t['us'] * 1e3 / peak:.2f}) | {t['cnt'] / t['alg']:.2f} |")
    open(os.path.join(dst, f"{rnd}_layer_table.md"), "w").write("\n".join(lines) + "\n")
    json.dump({"launches": n_conv, "serial_ms": tot["us"] / 1e3, "algorithmic_bytes": tot["alg"] * 1e6, "counter_bytes_fetch_x2_plus_write": tot["cnt"] * 1e6,
               "by_stage": by_stage,
               "counter_over_algorithmic_by_kernel": {k: round(t["cnt"] / t["alg"], 3) for k, t in by_family.items()},
               "serial_us_by_kernel": {k: round(t["us"], 1) for k, t in by_family.items()}}, open(os.path.join(dst, f"{rnd}_layer_traffic.json"), "w"), indent=1)
    print("\n".join(lines[:4]))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--dump":
        dump(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "f32")
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else "r04", sys.argv[2] if len(sys.argv) > 2 else "f32")
