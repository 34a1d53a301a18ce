#!/bin/bash
# SQ counters of owner plans / exact plans of the RoI-pool backward walk on the fixed roofline set.
# usage: bash tools/pmc_sq_owner.sh <outdir> "<owner plans>" "<exact plans>"
OUT=${1:-gpurun_out/pmc_owner_sq}; OWN=${2:-"0 4"}; EX=${3:-"11"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
: > $OUT/summary.txt
for kind in exact owner; do
  if [ $kind = exact ]; then L="$EX"; FLAG=--one; else L="$OWN"; FLAG=--one-owner; fi
  for v in $L; do
    i=0
    for pass in \
      "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
      "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
      "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
      "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
      i=$((i+1))
      timeout -k 5 150 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/${kind}${v}_g$i -- python3 tools/bwd_fixed_sweep.py $FLAG $v > $OUT/${kind}${v}_g$i.log 2>&1 || { echo "$kind $v pass $i failed" | tee -a $OUT/summary.txt; tail -3 $OUT/${kind}${v}_g$i.log; continue; }
      echo "== $kind $v group $i" >> $OUT/summary.txt
      python3 tools/pmc_summary.py $OUT/${kind}${v}_g$i bwd_walk | sort >> $OUT/summary.txt
    done
  done
done
cut -c1-30,92-140 $OUT/summary.txt
