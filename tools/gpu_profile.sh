#!/bin/bash
# The round's profile evidence in one GPU-box visit; `python tools/summarize_profiles.py rNN [subdir]` and `python tools/layer_table.py rNN`
# then copy the summaries into profiles/ (tracked).     tools/gpu_profile.sh [f32|bf16]
#   prof/         rocprofv3 --kernel-trace --stats of the bench command, lanes overlapped (as timed)
#   prof_serial/  the same with GRNET_MULTI_LANE=0: launches strictly one after another (per-kernel averages without overlap inflation)
#   pmc_sq/       SQ counters per dispatch, serial launches (MFMA busy, waits)
#   pmc/          FETCH_SIZE / WRITE_SIZE in separate passes (counters never share a run with a trace)
#   layers/       the launch list (grnet_describe_conv + kernel names) the per-launch table is joined on
# The program itself follows `--` (python3 bench.py ...): the profiler's preloaded library has initialised the GPU by then.
export TMPDIR=/tmp
DT=${1:-f32}
if [ "$DT" = bf16 ]; then D=gpurun_out/bf16; EXTRA="--dtype bf16 --frames 256"; else D=gpurun_out; EXTRA=""; fi
mkdir -p $D/prof $D/prof_serial $D/pmc_sq $D/pmc $D/layers
ARGS="bench.py $EXTRA --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-kernel-table"
SER="bench.py $EXTRA --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-table --no-graph --tune-level 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -o bench -- python3 $ARGS > $D/prof/bench_stdout.log 2>&1
GRNET_MULTI_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_serial -o bench -- python3 $SER > $D/prof_serial/bench_stdout.log 2>&1
GRNET_MULTI_LANE=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $D/pmc_sq -o sq -- python3 $SER > $D/pmc_sq/log.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  GRNET_MULTI_LANE=0 rocprofv3 --pmc $c --output-format csv -d $D/pmc -o $c -- python3 $SER > $D/pmc/$c.log 2>&1
done
GRNET_MULTI_LANE=0 python3 tools/layer_table.py --dump $D/layers/convs.json $DT > $D/layers/dump.log 2>&1
f=$(find $D/prof -name 'bench_kernel_trace.csv' | head -1); [ -n "$f" ] && python tools/trace_timeline.py "$f" 60 > $D/prof/timeline.txt 2>&1
find $D/prof $D/prof_serial $D/pmc_sq $D/pmc $D/layers -name "*.csv" -o -name "*.json" | head -20
