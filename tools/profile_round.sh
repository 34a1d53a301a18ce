#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/round/: the bench line of every
# workload, rocprofv3 kernel-trace stats of the SAME commands (find-db warm: the script fails if a
# MIOpen naive_conv_* kernel -- the find search -- shows up in a trace, or if the profiled run's
# images/s is more than 5 % below the unprofiled one), the idle-gap analysis of the default
# workload's timed steps (tools/trace_gaps.py), and -- on the roofline leg alone (tools/roofline_leg.py:
# the fixed R = 8512 RoI set bench.py reports its roofline on) -- kernel trace + one PMC pass per
# counter group.  Usage: bash tools/profile_round.sh <tag> bench [workloads...]   (bench lines + traces)
#                             bash tools/profile_round.sh <tag> leg                   (roofline leg: trace + PMC)
TAG=${1:-r03}; MODE=${2:-bench}; shift; shift
WORKLOADS=${@:-resnet50_joint_b8 resnet18_sup_b2 resnet50_alter resnet101_1600_test vgg16_joint}
OUT=gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
value() { grep '^{"metric"' $1 | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['value'])"; }
if [ "$MODE" = "bench" ]; then
for w in $WORKLOADS; do
  # (1) the bench line as the driver runs it (MIOpen find mode on the shipped find-db)
  timeout -k 10 400 python3 bench.py --workload $w --steps 10 --warmup 3 > $OUT/${TAG}_bench_$w.json.log 2>&1 || { echo "bench $w failed"; tail -5 $OUT/${TAG}_bench_$w.json.log; exit 1; }
  # (2) + (3) the traced step.  Under rocprofv3 the find-db path does not reproduce its untraced speed
  # (VERDICT r2), so the traced command uses MIOpen's heuristic solver choice (--no-miopen-benchmark)
  # and the SAME command runs untraced beside it: the two must agree within 5 %.  The per-kernel summary
  # that is kept covers the TIMED STEPS only (tools/trace_gaps.py): MIOpen runs naive_conv_* reference
  # kernels for every new convolution in the warm-up steps, which dominate a whole-process summary.
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -- python3 bench.py --workload $w --steps 5 --warmup 3 --no-cpu-baseline --no-miopen-benchmark > $OUT/${TAG}_bench_${w}_profiled_run.json.log 2>&1 || { echo "profiled bench $w failed"; exit 1; }
  timeout -k 10 400 python3 bench.py --workload $w --steps 5 --warmup 3 --no-cpu-baseline --no-miopen-benchmark > $OUT/${TAG}_bench_${w}_unprofiled_same_command.json.log 2>&1 || { echo "bench $w (heuristic) failed"; exit 1; }
  cp $(ls $OUT/prof_$w/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_${w}_whole_process_kernel_stats.csv
  mps=1; if [ "$w" = "resnet50_alter" ]; then mps=2; fi
  python3 tools/trace_gaps.py $OUT/prof_$w --steps 5 --warmup 3 --markers-per-step $mps --stats-csv $OUT/${TAG}_bench_${w}_kernel_stats.csv > $OUT/${TAG}_bench_${w}_step_gaps.json || { echo "timed steps of $w hold naive_conv kernels or the trace is short"; exit 1; }
  a=$(value $OUT/${TAG}_bench_${w}_unprofiled_same_command.json.log); b=$(value $OUT/${TAG}_bench_${w}_profiled_run.json.log)
  # (the 5 % bound is for the default workload; the small ones are bound by the host's launch rate, which
  # the tracer taxes: their ratio is recorded, not asserted)
  python3 -c "a,b=$a,$b; print('$w: untraced %.2f images/s, traced %.2f (%.1f %%)' % (a,b,100*b/a)); assert '$w' != 'resnet50_joint_b8' or b >= 0.95*a, 'traced run more than 5 % slower'" | tee -a $OUT/${TAG}_traced_vs_untraced.txt || exit 1
done
exit 0
fi
# the roofline leg alone: parity check on the timed set, kernel trace, then PMC passes (separate runs, kernel-trace only)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_leg -- python3 tools/roofline_leg.py --iters 20 --check > $OUT/${TAG}_roofline_leg.json.log 2>&1 || { echo "leg failed"; exit 1; }
cp $(ls $OUT/prof_leg/*/*kernel_stats.csv | head -1) $OUT/${TAG}_roofline_leg_kernel_stats.csv
tail -1 $OUT/${TAG}_roofline_leg.json.log | cut -c1-600
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  name=$(echo $pass | tr ' ' '+')
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 tools/roofline_leg.py --iters 5 --warmup 1 > $OUT/pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; exit 1; }
done
python3 tools/pmc_summary.py $OUT roi_pool | sort > $OUT/${TAG}_roofline_leg_pmc.txt
python3 tools/pmc_summary.py $OUT walk | sort >> $OUT/${TAG}_roofline_leg_pmc.txt
python3 tools/traffic_json.py $OUT $OUT/hotpath_traffic.json $OUT/${TAG}_roofline_leg.json.log
cat $OUT/${TAG}_roofline_leg_pmc.txt | cut -c1-150
