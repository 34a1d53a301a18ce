#!/bin/bash
# SQ counters of the NMS kernels (the mask as its own kernel against the mask role inside the fused launch), full-walk inputs:
# one rocprofv3 run per group, kernel-trace only.   bash tools/pmc_sq_nms.sh -> gpurun_out/pmc_sq_nms/summary.txt
OUT=gpurun_out/pmc_sq_nms
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pass in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_g$i -- python3 tools/nms_fused_ab.py --iters 5 --pred-scale 0.3 --nms-thresh 0.3 > $OUT/pmc_g$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pmc_g$i.log; }
done
python3 tools/pmc_summary.py $OUT nms_mask | sort > $OUT/summary.txt
cat $OUT/summary.txt | cut -c1-220
