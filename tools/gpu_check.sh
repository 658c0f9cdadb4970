#!/bin/bash
# One GPU-box visit: parity tests, smoke, the default bench.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8 > gpurun_out/rocminfo.txt
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -150 > gpurun_out/pytest_gpu.log
echo "pytest exit: ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
echo "smoke exit: $?" >> gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.log 2>&1
echo "bench exit: $?" >> gpurun_out/bench.log
tail -8 gpurun_out/pytest_gpu.log; tail -8 gpurun_out/smoke.log; tail -3 gpurun_out/bench.log
