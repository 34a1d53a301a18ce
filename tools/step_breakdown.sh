# kernel-time breakdown of the default bench step (rocprofv3 kernel trace, timed steps only)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_step -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline $BENCH_ARGS > gpurun_out/prof_step.log 2>&1
python3 - <<PY
import csv,glob,os
f=max(glob.glob("gpurun_out/prof_step/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# timed region = after the 4th roi_pool_bwd launch (3 warm-up steps + first timed)... use the last 5 backward launches
idx=[i for i,r in enumerate(rows) if "roi_pool_bwd" in r["Kernel_Name"]]
start=int(rows[idx[-5]]["Start_Timestamp"]); end=int(rows[idx[-1]]["Start_Timestamp"])
sel=[r for r in rows if start<=int(r["Start_Timestamp"])<end]
steps=4.0
agg={}
for r in sel:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    a=agg.setdefault(r["Kernel_Name"],[0.0,0]); a[0]+=d; a[1]+=1
print("kernel ms/step %.1f  wall ms/step %.1f" % (sum(v[0] for v in agg.values())/steps, (end-start)/1e6/steps))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][0])[:int(os.environ.get("TOPN","30"))]:
    print("%6.2f ms/step calls/step=%6.1f avg_us=%8.1f %s" % (v[0]/steps, v[1]/steps, v[0]/v[1]*1e3, k[:110]))
PY
