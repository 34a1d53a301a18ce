set -e
python -m pytest tests/test_gpu_roi_compact.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -2
mkdir -p gpurun_out/r4_prof
timeout -k 10 300 python tools/probes/alter_leg_split.py profiles/roofline_rois_resnet50_alter_weak_r4000.npy > gpurun_out/r4_prof/alter_leg_split_saved.log 2>&1
bash tools/profile_round.sh r04 leg > gpurun_out/round_leg.log 2>&1 || { tail -5 gpurun_out/round_leg.log; exit 1; }
tail -3 gpurun_out/round_leg.log | cut -c1-300
bash tools/profile_round.sh r04 bench resnet50_alter > gpurun_out/round_bench_alter.log 2>&1 || { tail -5 gpurun_out/round_bench_alter.log; exit 1; }
tail -2 gpurun_out/round_bench_alter.log
timeout -k 10 300 python tools/probes/alter_leg_split.py profiles/roofline_rois_vgg16_joint_r4128.npy 37,62,512 0:-1,8:-1,4:-1,16:-1,8:13,4:13,8:2,4:2,8:4,8:6 > gpurun_out/r4_prof/vgg_leg_split.log 2>&1 || true
cut -c1-140 gpurun_out/r4_prof/vgg_leg_split.log | tail -12
