# roi_pool_backward: tile / batch variants (WSSDL_ROI_BWD_VARIANT, see roi_pool.hip)
for cfgargs in "--config 3 --joint" "--config 3"; do
  for v in 0; do
    echo "== $cfgargs variant=$v"
    WSSDL_ROI_BWD_VARIANT=$v python3 tools/kernel_bench.py $cfgargs --iters 20 2>&1 | grep roi_pool_backward | cut -c1-120
  done
done
