#!/bin/bash
# bf16 path at BASELINE configs[2]'s shape (256 frames per step): kernel stats with the lanes overlapped and one after another, SQ
# counters (serial).  Everything lands in gpurun_out/bf16/; `python tools/summarize_profiles.py r02_bf16_n256 bf16` copies the summaries.
export TMPDIR=/tmp
D=gpurun_out/bf16
mkdir -p $D/prof $D/prof_serial $D/pmc_sq
ARGS="bench.py --dtype bf16 --frames 256 --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -o bench -- python3 $ARGS > $D/prof/bench_stdout.log 2>&1
GRNET_MULTI_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_serial -o bench -- python3 $ARGS --no-graph --tune-level 0 > $D/prof_serial/bench_stdout.log 2>&1
GRNET_MULTI_LANE=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $D/pmc_sq -o sq -- python3 bench.py --dtype bf16 --frames 256 --steps 2 --warmup 1 --no-cpu-baseline --no-graph --tune-level 0 > $D/pmc_sq/log.txt 2>&1
tail -1 $D/prof/bench_stdout.log | cut -c1-300
