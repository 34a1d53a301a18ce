#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/round/: gpu tests, smoke, the bench
# line, rocprofv3 kernel-trace stats of the same bench command, and -- on the roofline leg alone
# (tools/roofline_leg.py: the fixed R = 8512 RoI set bench.py reports its roofline on) -- kernel
# trace + one PMC pass per counter group.  Usage: bash tools/profile_round.sh [tag]
TAG=${1:-r02}
OUT=gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --timeout 900 > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1
tail -1 $OUT/smoke.log
python bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench_resnet50_joint_b8.json.log 2>&1
tail -1 $OUT/${TAG}_bench_resnet50_joint_b8.json.log | cut -c1-900
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_profiled_run.json.log 2>&1
cp $(ls $OUT/prof_bench/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_resnet50_joint_b8_kernel_stats.csv
# the roofline leg alone: kernel trace, then PMC passes (separate runs, kernel-trace only)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_leg -- python3 tools/roofline_leg.py --iters 20 > $OUT/${TAG}_roofline_leg.json.log 2>&1
cp $(ls $OUT/prof_leg/*/*kernel_stats.csv | head -1) $OUT/${TAG}_roofline_leg_kernel_stats.csv
tail -1 $OUT/${TAG}_roofline_leg.json.log | cut -c1-600
rocprofv3 -L > $OUT/counters_available.txt 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  name=$(echo $pass | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 tools/roofline_leg.py --iters 5 --warmup 1 > $OUT/pmc_$name.log 2>&1
done
python3 tools/pmc_summary.py $OUT roi_pool | sort > $OUT/${TAG}_roofline_leg_pmc.txt
python3 tools/pmc_summary.py $OUT walk | sort >> $OUT/${TAG}_roofline_leg_pmc.txt
python3 tools/traffic_json.py $OUT $OUT/hotpath_traffic.json $OUT/${TAG}_roofline_leg.json.log
cat $OUT/${TAG}_roofline_leg_pmc.txt | cut -c1-150
