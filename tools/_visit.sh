D=gpurun_out/v10; mkdir -p $D; export TMPDIR=/tmp
export GRNET_LIB_PATH=$PWD/video-based-gait-analysis-for-dementia_amd/libgrnet_hip_abl.so
{
for rep in 1 2; do
python tools/time_forward.py --tag "f32 default" 
GRNET_ABL_SKIP=keypoint_final_layer,smpl_final_layer python tools/time_forward.py --tag "f32 without the two 1x1 heads (128->25, 128->64)"
GRNET_ABL_SKIP=fuse_up python tools/time_forward.py --tag "f32 without the 8 grouped fuse launches"
GRNET_ABL_SKIP=layer1 python tools/time_forward.py --tag "f32 without layer1 (13 launches)"
GRNET_ABL_SKIP=backbone.conv1,backbone.conv2 python tools/time_forward.py --tag "f32 without the stem (2 launches)"
done
} 2>&1 | grep "frames:" | tee $D/f32_ablation.txt
unset GRNET_LIB_PATH
python tools/temporal_phases.py 10000 2>&1 | grep "T=" | tee $D/temporal_phases.txt
