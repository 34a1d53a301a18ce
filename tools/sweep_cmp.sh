# NMS sweep timing inside the default bench (rocprofv3 kernel stats); optional lib variants
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "nms or proposal or detect or train" 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sw -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline > gpurun_out/prof_sw.log 2>&1
python3 - <<PY
import csv,glob
import os; f=max(glob.glob("gpurun_out/prof_sw/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("nms_", "topk", "rank_")): print("%-44s calls=%4s avg_us=%9.2f" % (r["Name"][:44], r["Calls"], float(r["AverageNs"])/1e3))
PY
python3 tools/kernel_bench.py --config 3 --iters 10 2>&1 | grep "proposal_layer\|nms_12000"
python3 tools/kernel_bench.py --config 5 --iters 10 2>&1 | grep "proposal_layer\|nms_"
