mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
bash tools/gpu_sweep_r03.sh > gpurun_out/sweep.log 2>&1
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/sweep.log
