mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -q -x --timeout 600 2>&1 | tail -30 > gpurun_out/r3a/pytest.log
timeout 300 python tools/block_micro.py 16 > gpurun_out/r3a/micro.log 2>&1
tail -30 gpurun_out/r3a/pytest.log; cat gpurun_out/r3a/micro.log | grep -v "^$" | tail -20
