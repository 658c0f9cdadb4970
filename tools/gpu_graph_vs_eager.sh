export TMPDIR=/tmp
mkdir -p gpurun_out/graph
GRNET_TRACE=1 python bench.py --no-cpu-baseline --no-secondary 2>&1 | grep -E "tuned n=|^\{" | cut -c1-400 > gpurun_out/graph/tune_trace.txt
for mode in graph eager; do
  extra=""; [ $mode = eager ] && extra="--no-graph"
  python bench.py --steps 100 --warmup 10 --tune-level 0 --no-cpu-baseline --no-secondary $extra 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode untraced', d['value'], d['ms_per_step'])" >> gpurun_out/graph/tune_trace.txt
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/graph/$mode -o b -- python3 bench.py --steps 10 --warmup 3 --tune-level 0 --no-cpu-baseline --no-secondary $extra > gpurun_out/graph/$mode.log 2>&1
  f=$(find gpurun_out/graph/$mode -name 'b_kernel_trace.csv' | head -1)
  python tools/trace_timeline.py "$f" 60 > gpurun_out/graph/timeline_$mode.txt 2>&1
done
cat gpurun_out/graph/tune_trace.txt; head -14 gpurun_out/graph/timeline_graph.txt; head -14 gpurun_out/graph/timeline_eager.txt
