mkdir -p gpurun_out/r3c
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x --timeout 600 -k register_resident 2>&1 | tail -5 > gpurun_out/r3c/pytest.log
timeout 300 python tools/block_micro.py 16 2>&1 | grep -v "^$" > gpurun_out/r3c/micro.log
tail -3 gpurun_out/r3c/pytest.log; grep "201[0-9]" gpurun_out/r3c/micro.log
for v in "3 2 2" "3 1 2" "3 2 4" "1 2 2" "2 2 2"; do set -- $v
  GRNET_WINO4R=$1 GRNET_WINO4R_KS56=$2 GRNET_WINO4R_KS28=$3 timeout 600 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 > gpurun_out/r3c/bench_$1_$2_$3.json
  python -c "
import json;d=json.loads(open('gpurun_out/r3c/bench_$1_$2_$3.json').read());print('w4r $v:',d['value'],d['ms_per_step'])"
done
