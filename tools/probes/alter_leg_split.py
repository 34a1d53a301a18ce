#!/usr/bin/env python3
"""The alternating workload's weak-step launch (N = 2, R = 4000, C = 1024) INSIDE the roofline leg (forward, list building and
backward interleaved, as bench.py times it): exact walk against the split form, per plan.
    python3 tools/probes/alter_leg_split.py [rois.npy [H,W,C [split:plan,...]]]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import roofline_leg  # noqa: E402
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402

rois = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "roofline_rois_resnet50_alter_weak_r4000.npy"))
H, W, C = (int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (38, 63, 1024)
combos = [tuple(int(x) for x in c.split(":")) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else \
    [(0, -1), (8, -1), (4, -1), (8, 13), (4, 13), (2, 13), (2, 7), (0, 13), (0, 7)]
N = int(rois[:, 0].max()) + 1
for rep in range(2):
    for split, plan in combos:
        cfg.ROI_POOL_BWD_SPLIT = split
        _lib.set_tuning("roi_bwd_plan", plan)
        out, meta = roofline_leg.run(rois, N, H, W, C, iters=20, warmup=5)
        b = out["roi_pool_backward"]
        print(json.dumps(dict(split=split, plan_forced=plan, plan=meta["backward_plan"], segments=meta["backward_segments"],
                              backward_ms=round(b["avg_ms"], 4),
                              frac_moved=round(b["min_moved_bytes"] / (b["avg_ms"] * 1e-3) / 8e12, 3),
                              ops={k: round(v["avg_ms"], 4) for k, v in out.items()})), flush=True)
_lib.set_tuning("roi_bwd_plan", -1)
