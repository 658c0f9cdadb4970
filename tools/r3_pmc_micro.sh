export TMPDIR=/tmp
mkdir -p gpurun_out/r3pmc
GRNET_CONV_REPS=5 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d gpurun_out/r3pmc -o sq -- python3 tools/block_micro.py 16 > gpurun_out/r3pmc/log.txt 2>&1
ls gpurun_out/r3pmc
