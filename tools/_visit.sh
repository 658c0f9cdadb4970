export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bf16_roll.py -m gpu -q -x --timeout 900 2>&1 | tail -5
export GRNET_LIB_PATH=$PWD/video-based-gait-analysis-for-dementia_amd/libgrnet_hip_abl.so
for d in 0 2 8 16 31; do GRNET_ROLL_DBG=$d timeout 120 python tools/roll_micro.py 256 2>&1 | grep "^n="; done
