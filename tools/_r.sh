timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_round5.py -x -q -m gpu 2>&1 | tail -2
export TMPDIR=/tmp; mkdir -p gpurun_out/ps
GRNET_MULTI_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps -o b -- python3 bench.py --dtype bf16 --frames 256 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-table --no-graph --tune-level 0 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/ps/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('attn_pool','pw_stream','stem','fuse_sum')): print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
for i in 1 2; do python bench.py --dtype bf16 --frames 256 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"; done
