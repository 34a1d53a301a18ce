#!/bin/bash
# SQ counter passes on the roofline leg (tuning): one rocprofv3 run per group, kernel-trace only.
# (a pass with TA_* / TCP_* counters hung rocprofv3 on this pool and was dropped)
# usage: [LEG_ARGS="--rois profiles/x.npy --map 38,63,1024"] bash tools/pmc_sq.sh [kernel-name substring]    -> gpurun_out/pmc_sq/summary.txt
OUT=gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pass in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_LDS SQ_LEVEL_WAVES SQ_IFETCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_g$i -- python3 tools/roofline_leg.py --iters 3 --warmup 1 $LEG_ARGS > $OUT/pmc_g$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pmc_g$i.log; }
done
python3 tools/pmc_summary.py $OUT ${1:-roi_pool_fwd} | sort > $OUT/summary.txt
cat $OUT/summary.txt | cut -c1-200
