mkdir -p gpurun_out/r3a
GRNET_LIB_PATH=$PWD/video-based-gait-analysis-for-dementia_amd/libgrnet_hip_abl.so GRNET_BB_PHASES=1 GRNET_CONV_REPS=3 timeout 300 python tools/block_micro.py 16 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r3a/phases.log
cat gpurun_out/r3a/phases.log
