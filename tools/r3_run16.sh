mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x --timeout 600 -k "small_map" 2>&1 | tail -5
timeout 300 python tools/small_micro.py 16 2>&1 | grep -E "conv_micro|---" | grep -A5 "256 ch @ 7x7"
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 > gpurun_out/r3e/bench_$tag.json
  python -c "
import json;d=json.loads(open('gpurun_out/r3e/bench_$tag.json').read());print('$tag:',d['value'],d['ms_per_step'])"; }
run ksg4 A=1
run ksg2 GRNET_WINO4S_KSG=2
run ksg1 GRNET_WINO4S_KSG=1
