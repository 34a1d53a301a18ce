#!/bin/bash
# All randomized tools on the current tree (one GPU process at a time), summary lines into one log.
#   bash tools/fuzz_campaign.sh [out.log]
out=${1:-gpurun_out/fuzz_campaign.log}
: > "$out"
run() {   # name command...
  local name=$1; shift
  timeout -k 10 600 "$@" > gpurun_out/fuzz_$name.log 2>&1
  local rc=$?
  echo "$name: $(grep -E '^(cases|rounds) ' gpurun_out/fuzz_$name.log | tail -1) (rc $rc)" >> "$out"
  [ $rc -eq 0 ] || { echo "$name FAILED"; tail -5 gpurun_out/fuzz_$name.log; return 1; }
}
run nms_fuzz python3 tools/nms_fuzz.py --cases 2500 --seed 66 || exit 1
run roi_pool_fuzz python3 tools/roi_pool_fuzz.py --cases 300 --seed 66 || exit 1
run roi_blocks_fuzz python3 tools/roi_blocks_fuzz.py --cases 600 --seed 66 || exit 1
run layers_fuzz python3 tools/layers_fuzz.py --cases 400 --seed 66 || exit 1
run proposal_fuzz python3 tools/proposal_fuzz.py --cases 200 --seed 66 || exit 1
run image_fuzz python3 tools/image_fuzz.py --cases 250 --seed 66 || exit 1
run loss_fuzz python3 tools/loss_fuzz.py --cases 200 --seed 66 || exit 1
run mil_fuzz python3 tools/mil_fuzz.py --cases 400 --seed 66 || exit 1
run post_detect_fuzz python3 tools/post_detect_fuzz.py --cases 200 --seed 66 || exit 1
run sampler_fuzz python3 tools/sampler_fuzz.py --cases 150 --seed 66 || exit 1
run nms_fused_stress python3 tools/nms_fused_stress.py --rounds 600 || exit 1
run nms_fused_stress_busy python3 tools/nms_fused_stress.py --rounds 300 --busy || exit 1
cat "$out"
