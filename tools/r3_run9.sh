mkdir -p gpurun_out/r3e
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 > gpurun_out/r3e/bench_$tag.json
  python -c "
import json;d=json.loads(open('gpurun_out/r3e/bench_$tag.json').read());print('$tag:',d['value'],d['ms_per_step'])"; }
run base A=1
run prio2 GRNET_WINO4S_PRIO=2
run prio3 GRNET_WINO4S_PRIO=3
run prio2_noch GRNET_WINO4S_PRIO=2 GRNET_WINO_PRIO=0
run noch GRNET_WINO_PRIO=0
run lanes5 GRNET_LANES=5
run w4r2 GRNET_WINO4R=2
run w4r2ks4 GRNET_WINO4R=2 GRNET_WINO4R_KS28=4
