export TMPDIR=/tmp
for e in 0 1 2; do
  GRNET_GRAPH_EDGES=$e rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gq/$e -o b -- python3 bench.py --steps 10 --warmup 3 --tune-level 0 --no-cpu-baseline --no-secondary --no-kernel-table > gpurun_out/gq/$e.log 2>&1
  f=$(find gpurun_out/gq/$e -name 'b_kernel_trace.csv' | head -1)
  echo "edges $e"; python tools/trace_timeline.py "$f" 60 2>&1 | sed -n 1,12p
done
