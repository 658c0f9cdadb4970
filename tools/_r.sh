timeout 1800 python -m pytest tests/test_gpu_round5.py tests/test_gpu_bf16.py -x -q -m gpu 2>&1 | tail -3
for v in 0 1; do echo "ct32=$v"; GRNET_BF16_WIDE_CT32=$v python tools/shape_table.py bf16 256 2>/dev/null | grep "256->32"; done
for v in 0 1 0 1; do GRNET_BF16_WIDE_CT32=$v python bench.py --dtype bf16 --frames 256 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ct32=$v', d['value'], d['ms_per_step'])"; done
