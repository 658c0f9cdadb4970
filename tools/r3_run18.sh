mkdir -p gpurun_out/r3e
GRNET_WINO4_REPS=2 timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -q -x --timeout 600 -k "winograd_f43" 2>&1 | tail -2
GRNET_WINO4_REPS=2 timeout 300 python tools/block_micro.py 16 2>&1 | grep -E "hint 2001"
timeout 300 python tools/block_micro.py 16 2>&1 | grep -E "hint 2001"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 > gpurun_out/r3e/bench_$tag.json
  python -c "
import json;d=json.loads(open('gpurun_out/r3e/bench_$tag.json').read());print('$tag:',d['value'],d['ms_per_step'],d.get('parity'))"; }
run reps1 A=1
run reps2 GRNET_WINO4_REPS=2
run reps4 GRNET_WINO4_REPS=4
