#!/usr/bin/env python3
"""Standalone NMS (12000 -> 2000) on a sparse and a dense box population; run it against the regular
library and against a build with -DWSSDL_SWEEP_ABLATE=1 (helper waves idle, wrong results) to see the
resolver + barrier floor of the sweep:
    WSSDL_BUS_HIP_LIB=build_abl/libsweep1.so WSSDL_HIPCC_EXTRA=-DWSSDL_SWEEP_ABLATE=1 python -m wssdl_bus_amd.build --force
    python tools/sweep_ablation.py; WSSDL_BUS_HIP_LIB=$PWD/build_abl/libsweep1.so python tools/sweep_ablation.py
Round 2: 0.166 / 0.205 ms against 0.153 / 0.172 ms with idle helpers -- the helpers are not what bounds it."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from kernel_bench import timeit
from wssdl_bus_amd.nms.hip_nms import hip_nms
rs = np.random.RandomState(3)
n = 12000
for spread, label in ((1000, "sparse"), (250, "dense")):
    c = rs.uniform(0, spread, size=(n, 2)) * [1.0, 0.6]
    wh = np.exp(rs.normal(4.5, 0.6, size=(n, 2)))
    d = torch.from_numpy(np.hstack((c - wh / 2, c + wh / 2, rs.permutation(n)[:, None] / float(n))).astype(np.float32)).cuda()
    k = hip_nms(d, 0.7, max_keep=2000)
    ms = timeit(lambda: hip_nms(d, 0.7, max_keep=2000), 30)
    print(label, "kept", len(k), "ms", round(ms, 4))
