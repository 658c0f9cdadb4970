# usage: r3_trace.sh <tag> [ENV=VAL ...]: kernel trace + stats of a short bench run with the given environment
export TMPDIR=/tmp
tag=$1; shift
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out/r3t_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3t_$tag -o bench -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3t_$tag/stdout.log 2>&1
tail -1 gpurun_out/r3t_$tag/stdout.log | cut -c1-200
