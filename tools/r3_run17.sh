mkdir -p gpurun_out/r3e
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -q -x --timeout 600 -k "register_resident" 2>&1 | tail -2
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 > gpurun_out/r3e/bench_$tag.json
  python -c "
import json;d=json.loads(open('gpurun_out/r3e/bench_$tag.json').read());print('$tag:',d['value'],d['ms_per_step'])"; }
run base A=1
run r3_k11 GRNET_WINO4R=3 GRNET_WINO4R_KS56=1 GRNET_WINO4R_KS28=1
run r1_k1 GRNET_WINO4R=1 GRNET_WINO4R_KS56=1
run r2_k1 GRNET_WINO4R=2 GRNET_WINO4R_KS28=1
