#!/bin/bash
# HBM traffic of the bench step from the PMC counters, separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc -o $c -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --tune-level 0 > gpurun_out/pmc/$c.log 2>&1
  echo "$c exit $?" >> gpurun_out/pmc/$c.log
done
ls gpurun_out/pmc
