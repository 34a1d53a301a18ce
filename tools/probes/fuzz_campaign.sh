# long runs of the randomized parity tools (SEED, default 2026), one after the other; logs under gpurun_out/fuzz/
out=gpurun_out/fuzz; mkdir -p $out
run() { name=$1; shift; timeout -k 10 1000 python3 tools/$name.py "$@" --seed ${SEED:-2026} > $out/$name.log 2>&1; echo "$name: $(tail -1 $out/$name.log)"; }
run nms_fuzz --cases 2500
run roi_pool_fuzz --cases 300
run layers_fuzz --cases 400
run proposal_fuzz --cases 200
run image_fuzz --cases 250
run loss_fuzz --cases 200
run mil_fuzz --cases 400
run post_detect_fuzz --cases 200
run sampler_fuzz --cases 150
timeout -k 10 600 python3 tools/nms_fused_stress.py --rounds 1500 --busy > $out/nms_fused_stress.log 2>&1; echo "nms_fused_stress: $(tail -1 $out/nms_fused_stress.log)"
