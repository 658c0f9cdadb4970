#!/bin/bash
# Clip-length sweep of the fp32 path and two bf16 points (same library, same rules as the headline run) -> gpurun_out/sweep/*.json
mkdir -p gpurun_out/sweep; export TMPDIR=/tmp
for n in 4 8 16 32 64 128 256; do
  steps=$(( 3200 / n )); [ $steps -gt 300 ] && steps=300; [ $steps -lt 30 ] && steps=30
  timeout 600 python bench.py --frames $n --steps $steps --warmup 10 --no-cpu-baseline --no-secondary 2>gpurun_out/sweep/f32_$n.err | grep '^{' > gpurun_out/sweep/f32_$n.json
done
for n in 16 256; do
  timeout 600 python bench.py --dtype bf16 --frames $n --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/sweep/bf16_$n.err | grep '^{' > gpurun_out/sweep/bf16_$n.json
done
python - <<'PY'
import json, glob
out = {}
for f in sorted(glob.glob("gpurun_out/sweep/*.json")):
    try:
        d = json.loads(open(f).read()); r = d["roofline"]
        out[f.split("/")[-1][:-5]] = {"frames_per_s": d["value"], "ms_per_step": d["ms_per_step"], "frac": r["frac"], "effective_frac": r["effective_frac"]}
        print(f, d["value"], d["ms_per_step"], r["frac"], r["effective_frac"])
    except Exception as e:
        print(f, "unreadable", e)
json.dump(out, open("gpurun_out/sweep/summary.json", "w"), indent=1)
PY
