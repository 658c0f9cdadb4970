#!/bin/bash
# A/B of an environment switch on the default bench: tools/gpu_ab.sh VAR v1 v2 ...   (two alternating rounds)
var=$1; shift
for round in 1 2; do
  for v in "$@"; do
    export $var=$v
    python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ab.json
    python3 - "$var=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
r = d["roofline"]
print(sys.argv[1], "fps", d["value"], "ms/step", d["ms_per_step"], "conv_ms", r["conv_ms_per_step"], "serial", r["conv_ms_per_step_serial"], flush=True)
PY
  done
done
