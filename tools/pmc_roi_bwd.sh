#!/bin/bash
# PMC passes for the RoI-pool kernels (separate passes, kernel-trace only: see the HBM section
# of MI355X_MICROARCH.md).  Usage: tools/pmc_roi_bwd.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 tools/kernel_bench.py --config 3 --iters 3 > $OUT/$name.log 2>&1
}
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
run utcl TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum
python3 - <<'PY'
import csv, glob, collections, sys, os
out = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("OUT", "gpurun_out/pmc")
PY
