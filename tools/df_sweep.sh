#!/bin/bash
# dataflow HR section vs lane streams over frames per call and task granularity
export TMPDIR=/tmp
run() { timeout 150 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --frames $1 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('   ', j['value'], 'fps', j['ms_per_step'], 'ms', j['config']['kernel_launches_per_step'], 'launches', j['config']['launch'][:22])"; }
for n in 16 256; do
  echo "== frames $n: lanes"; GRNET_DATAFLOW=0 run $n
  for mt in 16; do echo "== frames $n: dataflow min_tasks $mt"; GRNET_DATAFLOW=1 GRNET_DF_MINTASKS=$mt run $n; done
done
