D=gpurun_out/v9; mkdir -p $D; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_bf16_roll.py tests/test_gpu_bf16.py tests/test_gpu_round5.py tests/test_gpu_options.py -m gpu -q -x --timeout 900 2>&1 | tail -4
timeout 120 python tools/roll_micro.py 256 2>&1 | grep "^n="
timeout 600 python bench.py --dtype bf16 --frames 256 --steps 20 --warmup 5 --no-cpu-baseline > $D/bench_bf16.json 2> $D/bench_bf16.err; tail -1 $D/bench_bf16.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/v9/bench_bf16.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config'].get('kernel_launches_per_step'), d.get('parity',{}).get('ok'))
PY
