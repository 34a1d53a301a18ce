#!/bin/bash
# Kernel times (rocprofv3 --kernel-trace --stats) and HBM-side bytes (FETCH_SIZE / WRITE_SIZE, one pass each) of the
# bin-owner backward next to the exact walk on the fixed roofline set.
#   usage: bash tools/pmc_owner.sh <outdir> "<owner plans>" "<exact plans>"
OUT=${1:-gpurun_out/r5_owner}; OWN=${2:-"0 9 4"}; EX=${3:-"11"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in $EX; do
  timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_exact_$v -- python3 tools/bwd_fixed_sweep.py --one $v > $OUT/stats_exact_$v.log 2>&1 || { echo "pass failed"; exit 1; }
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_exact_${v}_$c -- python3 tools/bwd_fixed_sweep.py --one $v > $OUT/pmc_exact_${v}_$c.log 2>&1 || { echo "pass failed"; exit 1; }
    echo "== exact $v"; python3 tools/pmc_summary.py $OUT/pmc_exact_${v}_$c bwd_walk
  done
done
for v in $OWN; do
  timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_owner_$v -- python3 tools/bwd_fixed_sweep.py --one-owner $v > $OUT/stats_owner_$v.log 2>&1 || { echo "pass failed"; exit 1; }
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_owner_${v}_$c -- python3 tools/bwd_fixed_sweep.py --one-owner $v > $OUT/pmc_owner_${v}_$c.log 2>&1 || { echo "pass failed"; exit 1; }
    echo "== owner $v"; python3 tools/pmc_summary.py $OUT/pmc_owner_${v}_$c walk_
  done
done
for f in $OUT/stats_*/*/*kernel_stats.csv $OUT/stats_*/*kernel_stats.csv; do [ -f "$f" ] && { echo "== $f"; grep -E "walk|Name" "$f" | cut -c1-220; }; done
