mkdir -p gpurun_out/r3e
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 > gpurun_out/r3e/bench_$tag.json
  python -c "
import json;d=json.loads(open('gpurun_out/r3e/bench_$tag.json').read());print('$tag:',d['value'],d['ms_per_step'])"; }
run base A=1
run ks2 GRNET_WINO4S_KS=2
run w4r1 GRNET_WINO4R=1
run w4r2 GRNET_WINO4R=2
run w4r3 GRNET_WINO4R=3
run w4r3_ks1 GRNET_WINO4R=3 GRNET_WINO4R_KS56=1
run lanes3 GRNET_LANES=3
run lanes6 GRNET_LANES=6
