#!/bin/bash
# One GPU-box visit: every GPU test, smoke(), the default bench (headline + secondary bf16 leg + cpu_baseline + parity), `--workload batchgen` and `--workload tracks` at one
# GPU, and the two-rank gloo rehearsals of `--gpus 2` (both workloads).  Everything lands in gpurun_out/check/.
#   /usr/local/graft/bin/gpurun --timeout 2700 -- 'bash tools/gpu_check.sh'
D=gpurun_out/check; mkdir -p $D; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -40 > $D/pytest_gpu.log; echo "pytest exit: ${PIPESTATUS[0]}" >> $D/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > $D/smoke.log 2>&1; echo "smoke exit: $?" >> $D/smoke.log
timeout 900 python bench.py > $D/bench.json 2> $D/bench.err; echo "bench exit: $?" >> $D/bench.err
timeout 600 python bench.py --workload batchgen > $D/bench_batchgen_n1.json 2> $D/bench_batchgen_n1.err
timeout 600 python bench.py --workload tracks > $D/bench_tracks_n1.json 2> $D/bench_tracks_n1.err
GRNET_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 30 --warmup 5 > $D/bench_gpus2_gloo.json 2> $D/bench_gpus2_gloo.err
GRNET_BENCH_BACKEND=gloo timeout 600 python bench.py --workload batchgen --gpus 2 --total-frames 2000 > $D/bench_batchgen_gpus2_gloo.json 2> $D/bench_batchgen_gpus2_gloo.err
tail -4 $D/pytest_gpu.log; tail -7 $D/smoke.log; for f in bench bench_batchgen_n1 bench_tracks_n1 bench_gpus2_gloo bench_batchgen_gpus2_gloo; do tail -1 $D/$f.json | cut -c1-260; done
python tools/temporal_phases.py 10000 > $D/temporal_phases.txt 2>&1; python tools/temporal_phases.py 450 >> $D/temporal_phases.txt 2>&1; cat $D/temporal_phases.txt
