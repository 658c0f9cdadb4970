#!/bin/bash
# SQ-side counters per kernel (one pass, 8 SQ slots), launches strictly serial so per-kernel numbers are clean.
mkdir -p gpurun_out/pmc_sq
export TMPDIR=/tmp
GRNET_MULTI_LANE=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_sq -o sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --tune-level 0 > gpurun_out/pmc_sq/log.txt 2>&1
echo "exit $?" >> gpurun_out/pmc_sq/log.txt
ls gpurun_out/pmc_sq
