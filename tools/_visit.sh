export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_temporal.py tests/test_gpu_harness.py -m gpu -q -x --timeout 900 > gpurun_out/pytest_temporal.log 2>&1
tail -3 gpurun_out/pytest_temporal.log
