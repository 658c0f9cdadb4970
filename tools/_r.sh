for v in 14 6; do echo "R=$v"; GRNET_BF16_WIDE64_R=$v python tools/shape_table.py bf16 256 2>/dev/null | grep "64->64   k3 s1 @56"; done
GRNET_BF16_WIDE64_R=6 timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "wide_band or stream_kernel" 2>&1 | tail -2
for v in 14 6 14 6; do GRNET_BF16_WIDE64_R=$v python bench.py --dtype bf16 --frames 256 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('R=$v', d['value'], d['ms_per_step'])"; done
