#!/bin/bash
# Round-3 evidence in one GPU-box visit (everything lands in gpurun_out/; tools/summarize_profiles.py r03 and tools/layer_table_r03.py
# copy the summaries to profiles/):
#   prof/         rocprofv3 --kernel-trace --stats of the bench command, lanes overlapped (as timed)
#   prof_serial/  the same with GRNET_MULTI_LANE=0: launches strictly one after another (per-kernel averages without overlap inflation)
#   pmc_sq/       SQ counters per dispatch, serial launches (MFMA busy, waits)
#   pmc/          FETCH_SIZE / WRITE_SIZE in separate passes (kernel-trace / stats only in their own runs)
#   layers/       the launch list (grnet_describe_conv) the per-launch table is joined on
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/prof_serial gpurun_out/pmc_sq gpurun_out/pmc gpurun_out/layers
ARGS="bench.py --steps 10 --warmup 3 --no-cpu-baseline"
SER="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --tune-level 0"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 $ARGS > gpurun_out/prof/bench_stdout.log 2>&1
GRNET_MULTI_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_serial -o bench -- python3 $SER > gpurun_out/prof_serial/bench_stdout.log 2>&1
GRNET_MULTI_LANE=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_sq -o sq -- python3 $SER > gpurun_out/pmc_sq/log.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  GRNET_MULTI_LANE=0 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc -o $c -- python3 $SER > gpurun_out/pmc/$c.log 2>&1
done
GRNET_MULTI_LANE=0 python3 tools/layer_table_r03.py --dump gpurun_out/layers/convs.json > gpurun_out/layers/dump.log 2>&1
find gpurun_out/prof gpurun_out/prof_serial gpurun_out/pmc_sq gpurun_out/pmc gpurun_out/layers -name "*.csv" -o -name "*.json" | head -20
