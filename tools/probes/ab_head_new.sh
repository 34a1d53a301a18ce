# same box, alternating: the proposal layer with the current library and with tools/probes/libwssdl_head.so
# (the committed HEAD's sources built by hand); WSSDL_BUS_HIP_LIB selects the library wssdl_bus_amd._lib loads
out=gpurun_out/r4_ab; mkdir -p $out
for rep in 1 2; do
  for v in new head; do
    if [ $v = new ]; then unset WSSDL_BUS_HIP_LIB; else export WSSDL_BUS_HIP_LIB=$PWD/tools/probes/libwssdl_head.so; fi
    timeout -k 10 120 python tools/nms_fused_ab.py --iters 50 > $out/ab_early_$v$rep.log 2>&1 || exit 1
    timeout -k 10 120 python tools/nms_fused_ab.py --iters 50 --pred-scale 0.3 --nms-thresh 0.3 > $out/ab_full_$v$rep.log 2>&1 || exit 1
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 10 --roofline-iters 2 > $out/ab_bench_$v$rep.log 2>&1 || exit 1
    echo "$v$rep early: $(grep -o 'proposal_layer_ms": [0-9.]*' $out/ab_early_$v$rep.log | tr '\n' ' ')"
    echo "$v$rep full:  $(grep -o 'proposal_layer_ms": [0-9.]*' $out/ab_full_$v$rep.log | tr '\n' ' ')"
    tail -1 $out/ab_bench_$v$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v$rep bench: hot path', d['hot_path']['gpu_ms_per_step'], 'proposal layer', d['roofline']['per_kernel']['proposal_layer']['avg_ms'])"
  done
done
