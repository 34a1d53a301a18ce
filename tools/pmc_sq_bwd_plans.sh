#!/bin/bash
# SQ / TCC counters of a few plans of the RoI-pool backward walk on the fixed roofline set: what the 64-channel forms
# wait for (round 4).  One rocprofv3 pass per counter group and plan (kernel-trace only).
# usage: bash tools/pmc_sq_bwd_plans.sh "13 7 25"   -> gpurun_out/pmc_bwd_sq/summary.txt
PLANS=${1:-"13 7 25"}
OUT=gpurun_out/pmc_bwd_sq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $OUT/summary.txt
for v in $PLANS; do
  i=0
  for pass in \
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
    "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
    "FETCH_SIZE" \
    "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout -k 5 150 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/p${v}_g$i -- python3 tools/bwd_fixed_sweep.py --one $v > $OUT/p${v}_g$i.log 2>&1 || { echo "plan $v pass $i failed" | tee -a $OUT/summary.txt; tail -3 $OUT/p${v}_g$i.log; continue; }
    echo "== plan $v group $i" >> $OUT/summary.txt
    python3 tools/pmc_summary.py $OUT/p${v}_g$i bwd_walk | sort >> $OUT/summary.txt
  done
done
cut -c1-30,92-140 $OUT/summary.txt
