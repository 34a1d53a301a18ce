#!/bin/bash
# Time (tools/bwd_fixed_sweep.py) and HBM-side reads (FETCH_SIZE, one rocprofv3 pass per plan) of a few plans of the
# RoI-pool backward walk on the fixed roofline set.  usage: bash tools/pmc_bwd_plans.sh   -> gpurun_out/r3j/
OUT=gpurun_out/r3j; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 python3 tools/bwd_fixed_sweep.py --plans 11,13,6,7,8,25,9,17 > $OUT/sweep.log 2>&1; cat $OUT/sweep.log
for v in 11 7 25 9; do
  timeout -k 5 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_${v}_fetch -- python3 tools/bwd_fixed_sweep.py --one $v > $OUT/pmc_${v}_fetch.log 2>&1 || { echo "pass failed: $v"; exit 1; }
  echo "== plan $v"; python3 tools/pmc_summary.py $OUT/pmc_${v}_fetch bwd_walk
done
