#!/usr/bin/env python3
"""Copy the rocprofv3 summaries produced on the GPU box (gpurun_out/, scratch) into profiles/ (tracked).

  profiles/rNN_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline`
  profiles/rNN_pmc_traffic.json   FETCH_SIZE / WRITE_SIZE (separate --pmc passes) of `bench.py --steps 3 --warmup 1 --no-graph`,
                                  summed over the conv launches of ONE forward, with the gfx950 correction of
                                  MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests at 64 B: x2 for wide reads)
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
# optional second argument: a sub-directory of gpurun_out/ holding the same layout (e.g. "bf16" for tools/gpu_profile_bf16.sh;
# the round tag then carries the variant: `summarize_profiles.py r02_bf16_n256 bf16`)
src, dst = os.path.join(ROOT, "gpurun_out", *(sys.argv[2:3])), os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = os.path.join(src, "prof", "bench_kernel_stats.csv")
if os.path.isfile(stats):
    shutil.copy(stats, os.path.join(dst, f"{rnd}_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    conv = [r for r in rows if "conv_" in r["Name"] or "hr_fuse_up" in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in conv)
    calls = sum(int(r["Calls"]) for r in conv)
    print(f"conv kernels: {calls} launches, {tot / 1e6:.2f} ms total, avg {tot / calls / 1e3:.2f} us/launch")
    for line in open(os.path.join(src, "prof", "bench_stdout.log")):
        if line.startswith("{"):
            open(os.path.join(dst, f"{rnd}_bench_under_rocprof.json"), "w").write(line)

serial = os.path.join(src, "prof_serial", "bench_kernel_stats.csv")
if os.path.isfile(serial):                      # GRNET_MULTI_LANE=0: launches strictly one after another
    shutil.copy(serial, os.path.join(dst, f"{rnd}_kernel_stats_serial.csv"))
    rows = list(csv.DictReader(open(serial)))
    conv = [r for r in rows if "conv_" in r["Name"] or "hr_fuse_up" in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in conv)
    calls = sum(int(r["Calls"]) for r in conv)
    print(f"serial run: conv kernels {calls} launches, {tot / 1e6:.2f} ms total, avg {tot / calls / 1e3:.2f} us/launch")
    for line in open(os.path.join(src, "prof_serial", "bench_stdout.log")):
        if line.startswith("{"):
            open(os.path.join(dst, f"{rnd}_bench_under_rocprof_serial.json"), "w").write(line)

out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = os.path.join(src, "pmc", f"{c}_counter_collection.csv")
    if not os.path.isfile(f):
        continue
    rows = list(csv.DictReader(open(f)))
    per_kernel = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        per_kernel[n][0] += float(r["Counter_Value"])
        per_kernel[n][1] += 1
    is_conv = lambda k: "conv_" in k or "hr_fuse_up" in k
    conv_launches_total = sum(v[1] for k, v in per_kernel.items() if is_conv(k))
    per_fw = int(os.environ.get("CONV_DISPATCHES_PER_FORWARD", "292"))   # fp32: 290 conv-class launches, two of them with a half-size last round (2 dispatches)
    lj = os.path.join(src, "layers", "convs.json")                       # tools/layer_table.py --dump: the plan with the dispatches each layer owns (bf16 chains: 0 / 1 / 4)
    if os.path.isfile(lj) and "CONV_DISPATCHES_PER_FORWARD" not in os.environ:
        plan = json.load(open(lj))
        if any("dispatches" in c for c in plan) and any(c.get("kernel_family", "").startswith("conv_bf16") for c in plan):
            per_fw = sum(c.get("dispatches", 1) for c in plan)
    n_forwards = max(1, round(conv_launches_total / per_fw))
    conv_kb = sum(v[0] for k, v in per_kernel.items() if is_conv(k))
    conv_launches = sum(v[1] for k, v in per_kernel.items() if is_conv(k))
    all_kb = sum(v[0] for v in per_kernel.values())
    out[c] = {"unit": "KB (rocprofv3 derived counter)", "forwards_in_run": n_forwards,
              "conv_kernels_kb_per_forward": conv_kb / n_forwards, "all_kernels_kb_per_forward": all_kb / n_forwards,
              "conv_launches_per_forward": conv_launches / n_forwards,
              "per_kernel_kb_per_dispatch": {k: v[0] / v[1] for k, v in sorted(per_kernel.items(), key=lambda x: -x[1][0])[:12]}}
if out:
    f, w = out.get("FETCH_SIZE"), out.get("WRITE_SIZE")
    if f and w:
        out["hbm_bytes_per_step_conv_kernels"] = (2.0 * f["conv_kernels_kb_per_forward"] + w["conv_kernels_kb_per_forward"]) * 1024.0
        out["correction"] = "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for 16 B/lane streaming reads), WRITE_SIZE x1"
        out["note"] = ("memory-side (fabric) requests of the L2: Infinity-Cache hits are counted, so this is an upper bound on HBM "
                       "bytes; the 16-frame working set (~1.7 GB of activations) does not fit the 256 MiB cache")
    lt = os.path.join(dst, f"{rnd}_layer_traffic.json")          # tools/layer_table.py: the same counters joined per launch (THIS round's only)
    if os.path.isfile(lt):
        t = json.load(open(lt))
        out["algorithmic_bytes_per_step_conv_kernels"] = t["algorithmic_bytes"]
        out["per_launch_table"] = f"profiles/{rnd}_layer_table.md"
    json.dump(out, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if not isinstance(v, dict)}, indent=1))


# SQ counters per kernel (tools/gpu_pmc_sq.sh / gpu_profile_r02.sh: GRNET_MULTI_LANE=0, one pass): MFMA pipe busy and wait shares
sq = os.path.join(src, "pmc_sq", "sq_counter_collection.csv")
if os.path.isfile(sq):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(sq)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("grk::", "")
        per[n][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[n].add(r["Dispatch_Id"])
    lines = ["kernel,dispatches,mfma_busy_frac,wait_any_frac,wait_inst_frac,active_inst_frac,valu_inst_frac,lds_inst_frac,busy_cu_cycles_per_dispatch"]
    tot = collections.defaultdict(float)
    for n, c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
        w, b = max(c.get("SQ_WAVE_CYCLES", 0), 1.0), max(c.get("SQ_BUSY_CU_CYCLES", 0), 1.0)
        lines.append(f'"{n}",{len(disp[n])},{c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * b):.4f},{c.get("SQ_WAIT_ANY", 0) / w:.4f},'
                     f'{c.get("SQ_WAIT_INST_ANY", 0) / w:.4f},{c.get("SQ_ACTIVE_INST_ANY", 0) / w:.4f},{c.get("SQ_ACTIVE_INST_VALU", 0) / w:.4f},'
                     f'{c.get("SQ_ACTIVE_INST_LDS", 0) / w:.4f},{b / len(disp[n]):.0f}')
        for k, v in c.items():
            tot[k] += v
    lines.append(f'"ALL KERNELS",{sum(len(d) for d in disp.values())},{tot["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * max(tot["SQ_BUSY_CU_CYCLES"], 1)):.4f},'
                 f'{tot["SQ_WAIT_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1):.4f},{tot["SQ_WAIT_INST_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1):.4f},'
                 f'{tot["SQ_ACTIVE_INST_ANY"] / max(tot["SQ_WAVE_CYCLES"], 1):.4f},{tot["SQ_ACTIVE_INST_VALU"] / max(tot["SQ_WAVE_CYCLES"], 1):.4f},'
                 f'{tot["SQ_ACTIVE_INST_LDS"] / max(tot["SQ_WAVE_CYCLES"], 1):.4f},')
    open(os.path.join(dst, f"{rnd}_sq_summary.csv"), "w").write("\n".join(lines) + "\n")
    print("SQ summary:", lines[-1])

g2 = os.path.join(src, "bench_gpus2_gloo.log")
if os.path.isfile(g2):
    for line in open(g2):
        if line.startswith("{"):
            open(os.path.join(dst, f"{rnd}_bench_gpus2_selflaunch_gloo.json"), "w").write(line)
