export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_conv_bf16.py tests/test_gpu_bf16.py tests/test_gpu_bf16_roll.py tests/test_gpu_options.py tests/test_gpu_harness.py -m gpu -q --timeout 1200 > gpurun_out/pytest_bf16.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_bf16.log | tail -2
