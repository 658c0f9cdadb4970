timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -x -q -m gpu -k "attention or tsattn or gait" 2>&1 | tail -3
python tools/temporal_phases.py 10000 2>&1 | tail -3
python tools/temporal_phases.py 2000 2>&1 | tail -3
