# same box, alternating: bench with the current library and with tools/probes/tmp_old.so (an older build placed there by hand)
mkdir -p gpurun_out/r3k
cp wssdl_bus_amd/libwssdl_bus_hip.so /tmp/cur.so
for rep in 1 2; do
  for v in cur old; do
    if [ $v = cur ]; then cp /tmp/cur.so wssdl_bus_amd/libwssdl_bus_hip.so; else cp tools/probes/tmp_old.so wssdl_bus_amd/libwssdl_bus_hip.so; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 8 > gpurun_out/r3k/ab_$v$rep.log 2>&1 || exit 1
    tail -1 gpurun_out/r3k/ab_$v$rep.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['hot_path']['gpu_ms_per_step'], d['roofline']['per_kernel']['proposal_layer']['avg_ms'])"
  done
done
cp /tmp/cur.so wssdl_bus_amd/libwssdl_bus_hip.so
