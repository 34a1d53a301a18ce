#!/bin/bash
# Round 6 (review item 7c): the forward of the reference's op contract (f32 top + i32 arg-max: 8 bytes per pooled element)
# on the default set -- kernel trace, HBM-side bytes and the SQ counters that say what it waits for.
#   bash tools/pmc_fwd_i32.sh   -> gpurun_out/fwd_i32/summary.txt
OUT=gpurun_out/fwd_i32
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/roofline_leg.py --i32 --iters 20 > $OUT/leg.json.log 2>&1 || { echo "trace failed"; tail -3 $OUT/leg.json.log; exit 1; }
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
i=0
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_THREAD_CYCLES_VALU" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_g$i -- python3 tools/roofline_leg.py --i32 --iters 3 --warmup 1 > $OUT/pmc_g$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/pmc_g$i.log; }
done
{ echo "# forward of the reference contract (i32 arg-max), default set R = 8512, C = 1024; mean per launch"; grep -h "roi_pool_fwd" $OUT/kernel_stats.csv | cut -c1-200;
  python3 tools/pmc_summary.py $OUT roi_pool_fwd | sort; tail -1 $OUT/leg.json.log | cut -c1-700; } > $OUT/summary.txt
rm -rf $OUT/pmc_g* $OUT/trace
cat $OUT/summary.txt | cut -c1-220
