#!/bin/bash
# Round 6: the bin-owner backward with 1 / 2 / 4 waves per tile stream against the split form (plan 7 x 8 segments) over
# the few-pair launch shapes.  bash tools/owner_split_ab.sh [out.log]
out=${1:-gpurun_out/owner_split_ab.log}
: > "$out"
one() {  # label, args...
  local label=$1; shift
  echo "== $label" >> "$out"
  timeout -k 10 250 python3 tools/bwd_fixed_sweep.py "$@" --iters 20 2>&1 | grep "^{" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('   ', {k: d[k] for k in ('plan', 'segments', 'owner', 'owner_segments', 'walk_ms', 'walk_plus_merge_ms', 'prepare_ms', 'frac_moved') if k in d})
" >> "$out"
}
one "VGG-16 1 + 2, its own proposals (3 x 512)" --plans 7 --segments 8 --owner 8,9 --owner-segments 1,2,4 --rois profiles/roofline_rois_vgg16_joint_r4128.npy --map 37,62 --channels 512
one "alternating weak step, large proposals (2 x 1024)" --plans 7 --segments 4 --owner 8 --owner-segments 1,2 --rois profiles/roofline_rois_resnet50_alter_weak_r4000_large.npy --map 38,63 --channels 1024
one "1 + 2 x 1024, VGG-16's proposals" --plans 7 --segments 4 --owner 8 --owner-segments 1,2 --rois profiles/roofline_rois_vgg16_joint_r4128.npy --map 38,63 --channels 1024
for spec in "0,4,5 512" "4 1024" "4,5 512" "4,5,6,7 256" "0,4,5 768" "4,5 256" "4 512" "4,5,6 256"; do
  set -- $spec
  one "default set, images $1 x $2 channels" --plans 7 --segments 8 --owner 8,9 --owner-segments 1,2,4 --images $1 --channels $2
done
cat "$out"
