#!/bin/bash
# One GPU-box visit at the end of round 3: parity tests, smoke, the default bench (with cpu_baseline + parity), the two-rank gloo rehearsal of
# `bench.py --gpus 2`, then the profiles of tools/gpu_profile_r03.sh.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -60 > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1
echo "bench exit: $?" >> gpurun_out/bench.log
GRNET_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 30 --warmup 5 > gpurun_out/bench_gpus2_gloo.log 2>&1; echo "exit $?" >> gpurun_out/bench_gpus2_gloo.log
bash tools/gpu_profile_r03.sh > gpurun_out/profile_r03.log 2>&1
tail -4 gpurun_out/pytest_gpu.log; tail -7 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log | cut -c1-600; tail -2 gpurun_out/bench_gpus2_gloo.log | cut -c1-300
