#!/bin/bash
# A/B of the bin-owner backward against the exact walk (and the split form) over launch shapes, on subsets / copies of the
# fixed roofline set.  usage: bash tools/owner_ab.sh "<owner plans>" > log
OWN=${1:-"0,1,2,3"}
for args in "" "--channels 512" "--dup 2" "--images 4,5,6,7,0" "--images 4,5,6,7" "--images 0,1,2,3,4,5" "--images 4,5" "--images 0,4,5" "--images 4" "--images 0,1" "--images 4,5 --channels 256" "--images 0,4,5 --channels 512"; do
  echo "=== $args"
  timeout -k 10 200 python tools/bwd_fixed_sweep.py --plans 11,13,23,7 --segments 1,4,8 --owner $OWN --iters 30 $args 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l.rstrip()); continue
    print({k:d[k] for k in d if k in ('plan','segments','owner','walk_ms','walk_plus_merge_ms','max_diff_over_max_abs')})
"
done
