#!/bin/bash
# One iteration visit to the GPU box: the round's new tests + the end-to-end parity file (stop at the first failure), then the default bench
# under each value of an A/B environment switch, then a kernel trace of the default configuration.
#   tools/gpu_iter.sh [ENV_NAME "v0 v1 ..."] [pytest files...]
mkdir -p gpurun_out/it
export TMPDIR=/tmp
AB=${1:-GRNET_FUSE_UP}; VALS=${2:-"0 1"}; shift 2
TESTS=${@:-tests/test_gpu_round4.py tests/test_gpu_parity.py}
timeout 1800 python -m pytest $TESTS -m gpu -q -x --timeout 900 2>&1 | tail -15 > gpurun_out/it/pytest.log
cat gpurun_out/it/pytest.log
for v in $VALS; do
  env $AB=$v timeout 600 python bench.py --no-cpu-baseline 2>gpurun_out/it/bench_${AB}_$v.err | grep '^{' > gpurun_out/it/bench_${AB}_$v.json
  python - "$AB=$v" gpurun_out/it/bench_${AB}_$v.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read())
    print(sys.argv[1], "frames/s", d["value"], "ms/step", d["ms_per_step"], "launches", d.get("config", {}).get("kernel_launches_per_step"), "parity", d.get("parity", {}).get("max_rel_err"))
except Exception as e:
    print(sys.argv[1], "bench failed:", e)
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/it/prof -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/it/prof_stdout.log 2>&1
f=$(find gpurun_out/it/prof -name 'bench_kernel_trace.csv' | head -1)
[ -n "$f" ] && python tools/trace_timeline.py "$f" 60 > gpurun_out/it/timeline.txt 2>&1 && head -12 gpurun_out/it/timeline.txt
