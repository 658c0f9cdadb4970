export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv_bf16.py tests/test_gpu_bf16.py -m gpu -q -x --timeout 900 2>&1 | tail -3
timeout 120 python tools/chain_micro.py 20 2>&1 | grep "chain<32"
