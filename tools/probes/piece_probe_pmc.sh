mkdir -p gpurun_out/r5g && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -w tools/probes/piece_bw_probe.hip -o /tmp/piece_bw_probe || exit 1
/tmp/piece_bw_probe
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/r5g/probe_pmc -- /tmp/piece_bw_probe > gpurun_out/r5g/probe_pmc.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r5g/probe_pmc/**/*counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    acc[(r['Dispatch_Id'],r['Kernel_Name'][:40])][r['Counter_Name']]=float(r['Counter_Value'])
t=glob.glob('gpurun_out/r5g/probe_pmc/**/*kernel_trace.csv',recursive=True)[0]
dur={r['Dispatch_Id']:(int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in csv.DictReader(open(t))}
for k,v in acc.items():
    if 'probe' in k[1]:
        ns=dur.get(k[0],0)
        print(k, 'ns',ns,'req %.3g'%v.get('TCP_TCC_READ_REQ_sum',0),'lat/req %.0f'%(v.get('TCP_TCC_READ_REQ_LATENCY_sum',0)/max(v.get('TCP_TCC_READ_REQ_sum',1),1)),'outstanding(@2.1GHz) %.0f'%(v.get('TCP_TCC_READ_REQ_LATENCY_sum',0)/max(ns*2.1,1)), 'pend %.3g'%v.get('TCP_PENDING_STALL_CYCLES_sum',0))
PY
