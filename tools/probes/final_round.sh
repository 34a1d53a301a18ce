python -m pytest tests/test_gpu_roi_compact.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -2
bash tools/profile_round.sh r04 leg > gpurun_out/round_leg.log 2>&1 || { tail -5 gpurun_out/round_leg.log; exit 1; }
tail -2 gpurun_out/round_leg.log | cut -c1-200
bash tools/profile_round.sh r04 bench > gpurun_out/round_bench.log 2>&1 || { tail -5 gpurun_out/round_bench.log; exit 1; }
tail -6 gpurun_out/round_bench.log
