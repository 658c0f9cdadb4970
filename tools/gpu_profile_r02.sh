#!/bin/bash
# Round-2 evidence in one GPU-box visit (everything lands in gpurun_out/, tools/summarize_profiles.py r02 copies the summaries to profiles/):
#   prof/         rocprofv3 --kernel-trace --stats of the bench command, lanes overlapped (as timed)
#   prof_serial/  the same with GRNET_MULTI_LANE=0: launches strictly one after another (per-kernel averages without overlap inflation)
#   pmc_sq/       SQ counters per dispatch, serial launches (MFMA busy, waits)
#   pmc/          FETCH_SIZE / WRITE_SIZE in separate passes
#   bench_gpus2_gloo.log   `python bench.py --gpus 2` self-launching two ranks that share GPU 0 (gloo): the N > 1 code path end to end
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/prof_serial gpurun_out/pmc_sq gpurun_out/pmc
ARGS="bench.py --steps 10 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 $ARGS > gpurun_out/prof/bench_stdout.log 2>&1
GRNET_MULTI_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_serial -o bench -- python3 $ARGS --no-graph --tune-level 0 > gpurun_out/prof_serial/bench_stdout.log 2>&1
GRNET_MULTI_LANE=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_sq -o sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --tune-level 0 > gpurun_out/pmc_sq/log.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc -o $c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --tune-level 0 > gpurun_out/pmc/$c.log 2>&1
done
GRNET_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 30 --warmup 5 > gpurun_out/bench_gpus2_gloo.log 2>&1; echo "exit $?" >> gpurun_out/bench_gpus2_gloo.log
find gpurun_out/prof gpurun_out/prof_serial gpurun_out/pmc_sq gpurun_out/pmc -name "*.csv" | head -20
tail -2 gpurun_out/bench_gpus2_gloo.log | cut -c1-400
