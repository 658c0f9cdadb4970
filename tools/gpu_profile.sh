#!/bin/bash
# rocprofv3 kernel trace of the default bench command; summaries are copied to profiles/ by hand.
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --tune-level 0 --no-graph > gpurun_out/prof/bench_stdout.log 2>&1
echo "rocprof exit: $?" >> gpurun_out/prof/bench_stdout.log
ls -R gpurun_out/prof | head -30
