export TMPDIR=/tmp
mkdir -p gpurun_out/prof_serial
GRNET_MULTI_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_serial -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --tune-level 0 > gpurun_out/prof_serial/bench_stdout.log 2>&1
head -25 gpurun_out/prof_serial/bench_kernel_stats.csv | cut -d, -f1-5 | cut -c1-150
