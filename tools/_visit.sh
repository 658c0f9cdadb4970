export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_conv_bf16.py tests/test_gpu_bf16.py tests/test_gpu_bf16_roll.py tests/test_gpu_options.py -m gpu -q -x --timeout 1200 > gpurun_out/pytest_bf16.log 2>&1
tail -3 gpurun_out/pytest_bf16.log
python bench.py --dtype bf16 --frames 256 --steps 40 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_bf16_256.json
python - <<'PY'
import json; j=json.loads(open("gpurun_out/bench_bf16_256.json").read()); print("bf16 256:", j["value"], j["ms_per_step"], j["roofline"]["frac"], j.get("parity",{}).get("ok"))
PY
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default.json
python - <<'PY'
import json; j=json.loads(open("gpurun_out/bench_default.json").read()); print("default:", j["value"], j["ms_per_step"], j["roofline"]["frac"], j["parity"]["ok"], "secondary", j["secondary"].get("value"), j["secondary"].get("ms_per_step"), j["secondary"].get("roofline",{}).get("frac"))
PY
