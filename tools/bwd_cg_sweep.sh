# roi_pool_backward: channels-per-workgroup sweep on the bench shapes
for cfgargs in "--config 3" "--config 3 --joint" "--config 2" "--config 5"; do
  for cg in 256 128 64; do
    echo "== $cfgargs cg=$cg"
    WSSDL_ROI_BWD_CG=$cg python3 tools/kernel_bench.py $cfgargs --iters 20 2>&1 | grep roi_pool_backward
  done
done
