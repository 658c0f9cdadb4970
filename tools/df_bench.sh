#!/bin/bash
# A/B of the dataflow HR section against the lane schedule on the headline workload.
export TMPDIR=/tmp
for df in 0 1; do
  echo "== GRNET_DATAFLOW=$df"
  GRNET_DATAFLOW=$df GRNET_TRACE=1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2> gpurun_out/df_$df.err | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], j['config']['kernel_launches_per_step'], j['config']['launch'], j['roofline']['conv_only_ms_per_step'], j['roofline']['conv_ms_per_step_serial'])"
  grep -E "tuned|dataflow plan|XCC" gpurun_out/df_$df.err | tail -4
done
