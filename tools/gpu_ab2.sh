#!/bin/bash
# A/B over two environment switches: tools/gpu_ab2.sh "A=1 B=2" "A=0 B=2" ...   (two alternating rounds)
for round in 1 2; do
  for combo in "$@"; do
    env $combo python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/ab.json
    python3 - "$combo" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
r = d["roofline"]
print(sys.argv[1], "fps", d["value"], "ms/step", d["ms_per_step"], "conv_ms", r["conv_ms_per_step"], "serial", r["conv_ms_per_step_serial"], flush=True)
PY
  done
done
