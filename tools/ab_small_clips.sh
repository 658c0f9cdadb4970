mkdir -p gpurun_out/sweep2
for n in 1 2 4 8 12; do for m in 0 1 2 7; do
  GRNET_WINO4S=$m timeout 300 python bench.py --frames $n --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/sweep2/n${n}_m$m.json
done; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/sweep2/*.json")):
    try:
        d = json.loads(open(f).read()); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "unreadable")
PY
