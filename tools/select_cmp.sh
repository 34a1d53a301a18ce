cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in default; do
  if [ $v = plain ]; then export WSSDL_BUS_HIP_LIB=$GRAFT_REPO_ROOT/wssdl_bus_amd/lib_plain.so; fi
  echo "== $v"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sel_$v -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline > gpurun_out/prof_sel_$v.log 2>&1
  python3 - <<PY
import csv,glob,os
f=max(glob.glob("gpurun_out/prof_sel_$v/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("subsample","roi_sample","topk_","rank_","mil_")): print("%-50s calls=%4s avg_us=%9.2f" % (r["Name"][:50], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
