# the workload's own leg (bench.py --save-fixed-rois) against tools/probes/alter_leg_split.py on the set that run generated: are the leg's numbers the RoI set or the leg?
mkdir -p gpurun_out/r4_prof; timeout -k 10 400 python bench.py --workload vgg16_joint --steps 6 --warmup 3 --no-cpu-baseline --save-fixed-rois gpurun_out/r4_prof/vgg_rois_now.npy > gpurun_out/r4_prof/vgg_bench_now.log 2>&1; python - <<EOF
import json,numpy as np
d=json.loads([l for l in open("gpurun_out/r4_prof/vgg_bench_now.log") if l.startswith("{\"metric")][-1])
fs=d["roofline"]["fixed_set"]; print("bench leg:", {k:round(v["avg_ms"],4) for k,v in fs.items()}, d["roofline"]["launch"])
for f in ("gpurun_out/r4_prof/vgg_rois_now.npy","profiles/roofline_rois_vgg16_joint_r4128.npy"):
    r=np.load(f); w=(r[:,3]-r[:,1]+1)/16; h=(r[:,4]-r[:,2]+1)/16
    print(f, r.shape, "mean w,h in cells %.1f %.1f"%(w.mean(),h.mean()), "images", np.bincount(r[:,0].astype(int)))
EOF
timeout -k 10 300 python tools/probes/alter_leg_split.py gpurun_out/r4_prof/vgg_rois_now.npy 37,62,512 0:-1,8:-1,4:-1,16:-1,8:13,4:13,8:6,16:6,8:8 2>&1 | cut -c1-150 | head -10
