mkdir -p gpurun_out/r3b
for f in 0 1 2 3; do
  GRNET_FUSE_BLOCKS=$f timeout 600 python bench.py --no-cpu-baseline --steps 200 > gpurun_out/r3b/bench_fuse$f.json 2> gpurun_out/r3b/bench_fuse$f.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/r3b/bench_fuse$f.json") if x.startswith("{")]
d=json.loads(l[-1]); print("fuse $f:", d["value"], d["ms_per_step"], d.get("parity"))
PY
done
