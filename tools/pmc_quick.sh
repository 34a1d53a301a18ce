#!/bin/bash
# usage: tools/pmc_quick.sh <outdir> ; env WSSDL_BWD_VARIANT honoured
OUT=${1:-gpurun_out/pmcq}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/kernel_bench.py --config 3 --iters 3 > $OUT/$name.log 2>&1; }
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run sq2 SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
python3 tools/pmc_summary.py $OUT roi_pool_bwd
