export TMPDIR=/tmp
mkdir -p gpurun_out/bf/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bf/prof -o b -- python3 bench.py --dtype bf16 --frames 256 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bf/prof.log 2>&1
python - <<"PY"
import csv, glob
f = glob.glob("gpurun_out/bf/prof/**/b_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:40]:
    n = r["Name"]
    if any(k in n for k in ("attn_pool", "head_tail", "softmax", "bilinear", "fuse_sum", "nchw_f32")): print(round(float(r["AverageNs"]) / 1e3, 1), r["Calls"], n[:70])
PY
