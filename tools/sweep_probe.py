#!/usr/bin/env python3
"""Cost model of the NMS sweep: time vs chunks processed and boxes kept (HIP events)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wssdl_bus_amd.nms.hip_nms import hip_nms  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def boxes(n, spread, seed=3):
    rs = np.random.RandomState(seed)
    c = rs.uniform(0, spread, size=(n, 2)) * [1.0, 0.6]
    wh = np.exp(rs.normal(4.5, 0.6, size=(n, 2)))
    d = np.hstack((c - wh / 2, c + wh / 2, rs.permutation(n)[:, None] / float(n))).astype(np.float32)
    return torch.from_numpy(d).cuda()


for spread in (1000, 300, 100):
    dd = boxes(12000, spread)
    order = torch.argsort(dd[:, 4], descending=True)
    rank = torch.empty_like(order)
    rank[order] = torch.arange(len(order), device="cuda")
    for mk in (64, 300, 1000, 2000, 2880, 4000, 12000):
        keep = hip_nms(dd, 0.7, max_keep=mk)
        keep = torch.as_tensor(keep).cuda().long()
        last = int(rank[keep].max())
        us = timeit(lambda: hip_nms(dd, 0.7, max_keep=mk))
        print("spread %4d max_keep %5d kept %5d chunks %3d total_us %7.1f" % (spread, mk, len(keep), last // 64 + 1, us))
