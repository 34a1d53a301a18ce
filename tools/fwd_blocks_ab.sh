#!/bin/bash
# A/B of the RoI-pool forward on the saved proposal sets: rows kernel (roi_fwd_blocks=0) against the block-table
# forward (=1), bin rows sorted / in RoI order.  One process per point, alternating.  bash tools/fwd_blocks_ab.sh [out.log]
out=${1:-gpurun_out/fwd_blocks_ab.log}
: > "$out"
run() {   # name rois map tune...
    local name=$1 rois=$2 map=$3; shift 3
    local args=()
    for t in "$@"; do args+=(--tune "$t"); done
    local extra=()
    [ -n "$rois" ] && extra=(--rois "$rois" --map "$map")
    python3 tools/roofline_leg.py --iters 30 --warmup 5 "${extra[@]}" "${args[@]}" | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); o=d['ops']
def g(k): return o[k]['avg_ms'] if k in o else 0.0
pm=o['roi_pool_forward'].get('parts_ms'); w=g('roi_pool_forward_windows')
f=pm['pooling'] if pm else g('roi_pool_forward'); p=pm['tables_and_order'] if pm else 0.0
mv=o['roi_pool_forward'].get('min_moved_bytes',0)
print('%-14s %-40s fwd %.4f  prepare %.4f  windows %.4f  sum %.4f ms  frac_moved(fwd+prepare) %.3f' % ('$name', '$*', f, p, w, f+p+w, mv/((f+p)*1e-3)/8e12))
" >> "$out" || return 1
}
for rep in ${REPS:-1 2}; do
  for blk in 0 1; do
    run default "" "" roi_fwd_blocks=$blk || exit 1
    run alter_large profiles/roofline_rois_resnet50_alter_weak_r4000_large.npy 38,63,1024 roi_fwd_blocks=$blk || exit 1
    run alter profiles/roofline_rois_resnet50_alter_weak_r4000.npy 38,63,1024 roi_fwd_blocks=$blk || exit 1
    run vgg profiles/roofline_rois_vgg16_joint_r4128.npy 37,62,512 roi_fwd_blocks=$blk || exit 1
  done
done
for parts in ${PARTS:-1 2}; do
run alter_large profiles/roofline_rois_resnet50_alter_weak_r4000_large.npy 38,63,1024 roi_fwd_blocks=1 roi_fwd_blocks_parts=$parts
run alter profiles/roofline_rois_resnet50_alter_weak_r4000.npy 38,63,1024 roi_fwd_blocks=1 roi_fwd_blocks_parts=$parts
run vgg profiles/roofline_rois_vgg16_joint_r4128.npy 37,62,512 roi_fwd_blocks=1 roi_fwd_blocks_parts=$parts
done
run alter_large profiles/roofline_rois_resnet50_alter_weak_r4000_large.npy 38,63,1024 roi_fwd_blocks=1 roi_fwd_blocks_sort=0
run vgg profiles/roofline_rois_vgg16_joint_r4128.npy 37,62,512 roi_fwd_blocks=1 roi_fwd_blocks_sort=0
run default "" "" roi_fwd_blocks=1 roi_fwd_blocks_sort=0
cat "$out"
