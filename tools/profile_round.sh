#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/round/: gpu tests, smoke, bench
# line, rocprofv3 kernel-trace stats of the same bench command, PMC passes for the RoI kernels.
OUT=gpurun_out/round
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1
python bench.py --steps 10 --warmup 3 > $OUT/bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1
done
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; tail -1 $OUT/bench.log | cut -c1-600
python3 tools/pmc_summary.py $OUT wssdl | sort > $OUT/pmc_hotpath.txt
python3 tools/traffic_json.py $OUT $OUT/hotpath_traffic.json
