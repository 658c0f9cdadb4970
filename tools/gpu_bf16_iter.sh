#!/bin/bash
# One iteration on the bf16 kernels: parity tests of the bf16 path, single-convolution timings, per-phase ticks, the 256-frame and 16-frame bench lines.
mkdir -p gpurun_out/bf
timeout 900 python -m pytest tests -m gpu -q -x -k "bf16" 2>&1 | tail -4 > gpurun_out/bf/pytest.log
python tools/bf16_micro.py 2> gpurun_out/bf/micro.log
GRNET_LIB_PATH=$PWD/video-based-gait-analysis-for-dementia_amd/libgrnet_hip_abl.so GRNET_BF16_PHASES=1 python tools/bf16_phases.py 2> gpurun_out/bf/phases.log
for n in 256 16; do timeout 600 python bench.py --dtype bf16 --frames $n --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/bf/bench_$n.json; done
cat gpurun_out/bf/pytest.log; grep -v amdgpu.ids gpurun_out/bf/micro.log gpurun_out/bf/phases.log | cut -d: -f2- ; python - <<'PY'
import json
for n in (256, 16):
    d = json.loads(open(f"gpurun_out/bf/bench_{n}.json").read()); print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], json.dumps(d.get("parity"))[:300])
PY
