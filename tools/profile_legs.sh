#!/bin/bash
# Round 6: the roofline leg of EVERY workload under rocprofv3 -- kernel trace + stats, then one PMC pass per counter
# group (separate runs, kernel-trace only) -- on the workload's committed RoI set (profiles/roofline_rois_*.npy).
#   bash tools/profile_legs.sh <tag>            -> gpurun_out/legs/<tag>_roofline_leg_<workload>_{json.log,kernel_stats.csv,pmc.txt}
#                                                  and gpurun_out/legs/hotpath_traffic.json (default at the top level, others under "legs")
TAG=${1:-r06}
OUT=gpurun_out/legs
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp profiles/hotpath_traffic.json $OUT/hotpath_traffic.json 2>/dev/null
leg() {   # name rois map
  local name=$1 rois=$2 map=$3
  local extra=()
  [ -n "$rois" ] && extra=(--rois "$rois" --map "$map")
  rm -rf $OUT/prof_$name $OUT/pmc_*
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- python3 tools/roofline_leg.py --iters 20 "${extra[@]}" > $OUT/${TAG}_roofline_leg_$name.json.log 2>&1 || { echo "leg $name failed"; tail -3 $OUT/${TAG}_roofline_leg_$name.json.log; return 1; }
  cp $(ls $OUT/prof_$name/*/*kernel_stats.csv | head -1) $OUT/${TAG}_roofline_leg_${name}_kernel_stats.csv
  for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    pn=$(echo $pass | tr ' ' '+')
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$pn -- python3 tools/roofline_leg.py --iters 5 --warmup 1 "${extra[@]}" > $OUT/pmc_$pn.log 2>&1 || { echo "pmc pass $pn of $name failed"; return 1; }
  done
  python3 tools/pmc_summary.py $OUT roi_pool | sort > $OUT/${TAG}_roofline_leg_${name}_pmc.txt
  python3 tools/pmc_summary.py $OUT walk | sort >> $OUT/${TAG}_roofline_leg_${name}_pmc.txt
  python3 tools/pmc_summary.py $OUT blocks_build | sort >> $OUT/${TAG}_roofline_leg_${name}_pmc.txt
  python3 tools/pmc_summary.py $OUT rows_scatter | sort >> $OUT/${TAG}_roofline_leg_${name}_pmc.txt
  if [ "$name" = "resnet50_joint_b8" ]; then
    python3 tools/traffic_json.py $OUT $OUT/hotpath_traffic.json $OUT/${TAG}_roofline_leg_$name.json.log
  else
    python3 tools/traffic_json.py $OUT $OUT/hotpath_traffic.json $OUT/${TAG}_roofline_leg_$name.json.log $name
  fi
  rm -rf $OUT/pmc_* $OUT/prof_$name
  echo "leg $name done"
}
leg resnet50_joint_b8 "" "" || exit 1
leg resnet50_alter profiles/roofline_rois_resnet50_alter_weak_r4000_large.npy 38,63,1024 || exit 1
leg vgg16_joint profiles/roofline_rois_vgg16_joint_r4128.npy 37,62,512 || exit 1
[ -f profiles/roofline_rois_resnet101_1600_test_r300.npy ] && { leg resnet101_1600_test profiles/roofline_rois_resnet101_1600_test_r300.npy 63,100,1024 || exit 1; }
[ -f profiles/roofline_rois_resnet18_sup_b2_r256.npy ] && { leg resnet18_sup_b2 profiles/roofline_rois_resnet18_sup_b2_r256.npy 38,63,256 || exit 1; }
ls $OUT
