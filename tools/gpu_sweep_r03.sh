#!/bin/bash
# Clip-length sweep of the fp32 path and the two bf16 points of DESIGN.md section 5 (same library, same rules as the headline run).
mkdir -p gpurun_out/sweep
export TMPDIR=/tmp
for n in 4 8 16 32 64 128 256; do
  steps=$(( 3200 / n )); [ $steps -gt 300 ] && steps=300; [ $steps -lt 30 ] && steps=30
  timeout 600 python bench.py --frames $n --steps $steps --warmup 10 --no-cpu-baseline 2>gpurun_out/sweep/f32_$n.err | grep '^{' > gpurun_out/sweep/f32_$n.json
done
for n in 16 256; do
  timeout 600 python bench.py --dtype bf16 --frames $n --steps 60 --warmup 10 --no-cpu-baseline 2>gpurun_out/sweep/bf16_$n.err | grep '^{' > gpurun_out/sweep/bf16_$n.json
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/sweep/*.json")):
    try:
        d = json.loads(open(f).read()); r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], r["frac"], r.get("executed_frac"), d.get("parity", {}).get("ok") if isinstance(d.get("parity"), dict) else d.get("parity"))
    except Exception as e:
        print(f, "unreadable", e)
PY
