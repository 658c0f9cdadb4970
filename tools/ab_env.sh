#!/bin/bash
# alternating A/B of the default bench over one environment switch: tools/ab_env.sh VAR valueA valueB [reps]
export TMPDIR=/tmp
V=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do for x in $A $B; do
  env $V=$x python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('$V=$x', j['value'], j['ms_per_step'], j['config']['kernel_launches_per_step'])"
done; done
