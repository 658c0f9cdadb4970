#!/bin/bash
# Quick visit: all GPU tests (stop at the first failure), smoke, the default bench without the CPU baseline, a kernel trace summary of the tail.
mkdir -p gpurun_out/q
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -8 > gpurun_out/q/pytest.log
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/q/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/q/prof -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/q/prof_stdout.log 2>&1
cat gpurun_out/q/pytest.log
python - <<'PY'
import json, csv, glob
d = json.loads(open("gpurun_out/q/bench.json").read()); print("bench", d["value"], d["ms_per_step"])
f = glob.glob("gpurun_out/q/prof/**/bench_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    n = r["Name"]
    if any(k in n for k in ("attn_pool", "head_tail", "softmax_stats", "smpl_", "conv_mfma_f32<true, 3, 2, 7, 4, 1, 4, 8>")):
        print(f'{float(r["AverageNs"]) / 1e3:8.1f} us  x{r["Calls"]:>5}  {n[:90]}')
PY
