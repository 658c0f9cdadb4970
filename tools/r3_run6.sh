mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x --timeout 600 -k small_map 2>&1 | tail -15 > gpurun_out/r3d/pytest.log
tail -12 gpurun_out/r3d/pytest.log
timeout 300 python tools/small_micro.py 16 2>&1 | grep -v "^$" > gpurun_out/r3d/micro.log; grep -E "conv_micro|---" gpurun_out/r3d/micro.log
for v in 0 1 2 3 7; do
  GRNET_WINO4S=$v timeout 600 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 > gpurun_out/r3d/bench_$v.json
  python -c "
import json;d=json.loads(open('gpurun_out/r3d/bench_$v.json').read());print('w4s $v:',d['value'],d['ms_per_step'],d.get('parity'))"
done
