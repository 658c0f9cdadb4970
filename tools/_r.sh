GRNET_BF16_POOL_WAVES=12 timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_round5.py -x -q -m gpu -k "forward or pool or storage or stream_kernel" 2>&1 | tail -2
export TMPDIR=/tmp
for w in 6 12; do mkdir -p gpurun_out/ps$w
GRNET_BF16_POOL_WAVES=$w GRNET_MULTI_LANE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps$w -o b -- python3 bench.py --dtype bf16 --frames 256 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-kernel-table --no-graph --tune-level 0 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/ps$w/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'attn_pool' in r['Name'] or 'head_tail' in r['Name']: print($w, r['Name'][:60], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
for w in 6 12 6 12; do GRNET_BF16_POOL_WAVES=$w python bench.py --dtype bf16 --frames 256 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('waves=$w', d['value'], d['ms_per_step'])"; done
