mkdir -p gpurun_out/r3e
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 > gpurun_out/r3e/bench_$tag.json
  python -c "
import json;d=json.loads(open('gpurun_out/r3e/bench_$tag.json').read());print('$tag:',d['value'],d['ms_per_step'])"; }
run reps1 A=1
run reps2 GRNET_WINO4_REPS=2
