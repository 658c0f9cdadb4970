export TMPDIR=/tmp
export GRNET_LIB_PATH=$PWD/video-based-gait-analysis-for-dementia_amd/libgrnet_hip_abl.so
for d in 0 16 17 18 20 22 23; do echo "dbg bits $d (1 no k-loop, 2 no epilogue, 4 no seeds, 16 no barrier)"; GRNET_BF16_FRAME_DBG=$d python3 tools/chain_micro.py 20 2>&1 | grep "chain<32"; done
