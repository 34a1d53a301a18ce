#!/bin/bash
# Forward ablation matrix (tuning): build_abl/libablN.so = the library built with
# WSSDL_HIPCC_EXTRA=-DWSSDL_FWDC_ABLATE=N WSSDL_BUS_HIP_LIB=build_abl/libablN.so python -m wssdl_bus_amd.build --force
# usage: bash tools/fwd_ablation.sh "0 1 2 4 5" [variants]      (0 = the regular library)
set -e
out=gpurun_out/fwd_abl
mkdir -p $out
for a in ${1:-0 1}; do
  lib=$PWD/build_abl/libabl$a.so
  [ "$a" = 0 ] && lib=$PWD/wssdl_bus_amd/libwssdl_bus_hip.so
  KB_NO_CHECK=1 KB_FWD_VARIANTS=${2:-1,0} WSSDL_BUS_HIP_LIB=$lib python tools/kernel_bench.py --config 3 --joint > $out/kb_abl$a.log 2>&1
done
python - "$out" ${1:-0 1} <<PY
import json, sys
out = sys.argv[1]
for a in sys.argv[2:]:
    for l in open("%s/kb_abl%s.log" % (out, a)):
        l = l.strip()
        if l.startswith("{"):
            d = json.loads(l)
            if "op" in d and "roi_pool_forward_compact" in d["op"]:
                print("ablate", a, d["op"], round(d["ms"], 4))
PY
