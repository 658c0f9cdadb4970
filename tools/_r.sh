timeout 1000 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q -m gpu -k "bilinear or stream or layer1 or s2 or stride" 2>&1 | tail -3
for v in 0 1; do echo "wreg=$v"; GRNET_BF16_S2_WREG=$v python tools/shape_table.py bf16 256 2>/dev/null | grep "conv_bf16_s2"; done
for v in 0 1 0 1; do GRNET_BF16_S2_WREG=$v python bench.py --dtype bf16 --frames 256 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('wreg=$v', d['value'], d['ms_per_step'])"; done
