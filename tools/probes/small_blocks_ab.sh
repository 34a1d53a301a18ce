#!/bin/bash
# round 6: block-table forward forced on the two small launch shapes (R-101 test R = 300, R-18 R = 256) against the rule's choice
out=gpurun_out/small_blocks_ab.log
: > $out
run() { # name rois map tunes...
  local name=$1 rois=$2 map=$3; shift 3
  local args=(); for t in "$@"; do args+=(--tune "$t"); done
  timeout -k 10 200 python3 tools/roofline_leg.py --iters 30 --warmup 5 --rois $rois --map $map $WT "${args[@]}" > gpurun_out/small_ab.json 2> gpurun_out/small_ab.err || { tail -5 gpurun_out/small_ab.err; return 1; }
  WT="$WT" python3 - "$name" "$*" >> $out <<'PY'
import json,sys
d=json.loads(open('gpurun_out/small_ab.json').readlines()[-1]); o=d['ops']
f=o['roi_pool_forward']
import os
print(sys.argv[1], sys.argv[2], os.environ.get('WT',''), 'fwd', round(f['avg_ms'],4), 'parts', f.get('parts_ms'), 'windows', round(o.get('roi_pool_forward_windows',{}).get('avg_ms',0),4), 'variant', d.get('launch',{}).get('forward_variant'))
PY
}
for rep in 1 2; do
for b in 0 1 2; do
WT=""; [ $b = 2 ] && { b=0; WT="--window-table-min-rois 1"; }
[ $b = 1 ] && WT="--window-table-min-rois 1"
run r101 profiles/roofline_rois_resnet101_1600_test_r300.npy 63,100,1024 roi_fwd_blocks=$b || exit 1
run r18 profiles/roofline_rois_resnet18_sup_b2_r256.npy 38,63,256 roi_fwd_blocks=$b || exit 1
done
done
cat $out
