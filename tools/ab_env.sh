#!/bin/bash
# A/B whole-forward time under values of one environment variable:  tools/ab_env.sh NAME "v0|v1|..." [time_forward.py args]
# (values separated by |; "-" = unset).  tools/time_forward.py: no parity check, so timing-only ablations (GRNET_ABL_SKIP) run too.
NAME=$1; IFS='|' read -ra VALS <<< "$2"; shift 2
for v in "${VALS[@]}"; do
  if [ "$v" = "-" ]; then unset $NAME; else export $NAME="$v"; fi
  python tools/time_forward.py --tag "$NAME=$v" "$@" 2>/dev/null
done
