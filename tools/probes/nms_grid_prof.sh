#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_grid
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_grid -- python3 tools/nms_grid_ab.py --iters 10 --reps 1 > gpurun_out/nms_grid_prof.log 2>&1 || { tail -5 gpurun_out/nms_grid_prof.log; exit 1; }
f=$(ls gpurun_out/prof_grid/*/*kernel_stats.csv | head -1)
test -n "$f" && grep -i "nms_grid\|Name\|fused" "$f" > gpurun_out/nms_grid_kernel_stats.csv
cat gpurun_out/nms_grid_kernel_stats.csv
