mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x --timeout 600 -k "chain or small_map" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -q -x --timeout 600 -k "golden or stage_taps or batch_invariance or graph_replay or winograd_layers_match" 2>&1 | tail -5
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 > gpurun_out/r3e/bench_$tag.json
  python -c "
import json;d=json.loads(open('gpurun_out/r3e/bench_$tag.json').read());print('$tag:',d['value'],d['ms_per_step'])"; }
run chain A=1
run nochain GRNET_WINO4S_CHAIN=0
