#!/bin/bash
# Per-layer evidence: every convolution launch of a step, one after another (GRNET_MULTI_LANE=0, cost-model launch
# configurations), with its duration (kernel trace) and its L2<->fabric bytes (FETCH_SIZE / WRITE_SIZE, separate --pmc
# passes).  tools/layer_table.py joins the three CSVs with the launch list into profiles/rNN_layer_table.{md,csv}.
mkdir -p gpurun_out/layers
export TMPDIR=/tmp
export GRNET_MULTI_LANE=0
python3 tools/layer_table.py --dump gpurun_out/layers/convs.json > gpurun_out/layers/dump.log 2>&1
ARGS="bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --tune-level 0"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/layers -o trace -- python3 $ARGS > gpurun_out/layers/trace.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/layers -o $c -- python3 $ARGS > gpurun_out/layers/$c.log 2>&1
done
ls -la gpurun_out/layers
