#!/bin/bash
# SQ / TA / TCP counter passes on the roofline leg (tuning): one rocprofv3 run per group, kernel-trace only.
# usage: bash tools/pmc_sq.sh [kernel-name substring]    -> gpurun_out/pmc_sq/summary.txt
OUT=gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for pass in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU" \
  "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_g$i -- python3 tools/roofline_leg.py --iters 3 --warmup 1 > $OUT/pmc_g$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pmc_g$i.log; }
done
python3 tools/pmc_summary.py $OUT ${1:-roi_pool_fwd} | sort > $OUT/summary.txt
cat $OUT/summary.txt | cut -c1-200
