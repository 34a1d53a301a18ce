#!/bin/bash
out=gpurun_out/one_bin_ab.log
: > $out
timeout -k 10 300 python3 tools/probes/one_bin_ab.py >> $out 2>&1 || { tail -20 $out; exit 1; }
run() { # name rois map tunes...
  local name=$1 rois=$2 map=$3; shift 3
  local args=(); for t in "$@"; do args+=(--tune "$t"); done
  timeout -k 10 200 python3 tools/roofline_leg.py --iters 30 --warmup 5 --rois $rois --map $map "${args[@]}" > gpurun_out/small_ab.json 2> gpurun_out/small_ab.err || { tail -5 gpurun_out/small_ab.err; return 1; }
  python3 - "$name" "$*" >> $out <<'PY'
import json,sys
d=json.loads(open('gpurun_out/small_ab.json').readlines()[-1]); o=d['ops']
print(sys.argv[1], sys.argv[2], 'fwd', round(o['roi_pool_forward']['avg_ms'],4))
PY
}
for rep in 1 2; do
for b in 7; do
run r101 profiles/roofline_rois_resnet101_1600_test_r300.npy 63,100,1024 roi_fwd_one_bin=$b || exit 1
run r18 profiles/roofline_rois_resnet18_sup_b2_r256.npy 38,63,256 roi_fwd_one_bin=$b || exit 1
done
done
cat $out
