#!/usr/bin/env python3
"""round 6: R >= 1024 with few channels -- window table + rows kernel (what the host layer does today) against one wave per
bin without a table, events around the library calls (the table launch counts)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op  # noqa: E402


def event_ms(fn, iters=30, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    saved = _lib.timeline.enabled, _lib.timeline.records
    try:
        _lib.timeline.reset(True)
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        tl = _lib.timeline.summary()
    finally:
        _lib.timeline.enabled, _lib.timeline.records = saved
    return tl["roi_pool_forward"]["avg_ms"] + tl.get("roi_pool_forward_windows", {"avg_ms": 0.0})["avg_ms"]


sets = {"test-sized": (np.load(os.path.join(ROOT, "profiles/roofline_rois_resnet101_1600_test_r300.npy")), (63, 100)),
        "train-sized": (np.load(os.path.join(ROOT, "profiles/roofline_rois_resnet18_sup_b2_r256.npy")), (38, 63))}
rng = np.random.RandomState(3)
for tag, (base, (H, W)) in sets.items():
    for C, Rs in ((256, (1024, 1500, 2000, 3000, 4500)), (512, (1024, 1500, 2000, 2300)), (1024, (1024, 1150))):
        for R in Rs:
            pick = base[rng.randint(0, base.shape[0], R)].copy()
            pick[:, 1:] += rng.uniform(-8, 8, (R, 4)).astype(np.float32)
            pick[:, 3] = np.maximum(pick[:, 3], pick[:, 1] + 4); pick[:, 4] = np.maximum(pick[:, 4], pick[:, 2] + 4)
            N = int(base[:, 0].max()) + 1
            rois = torch.from_numpy(pick).cuda()
            data = torch.randn((N, H, W, C), device="cuda", generator=torch.Generator("cuda").manual_seed(1))
            res = {}
            ref = None
            for name, min_rois, one in (("table+rows", 1024, 7), ("one-bin", 10 ** 9, 107), ("sliced/rows no table", 10 ** 9, 0)):
                op._WINDOW_TABLE_MIN_ROIS = min_rois
                with _lib.tuned(roi_fwd_one_bin=one, roi_fwd_blocks=0):
                    top, arg = op.roi_pool_compact(data, rois, 7, 7, 1.0 / 16)
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = (top.clone(), arg.clone())
                    assert torch.equal(top, ref[0]) and torch.equal(arg, ref[1]), (tag, C, R, name)
                    res[name] = event_ms(lambda: op.roi_pool_compact(data, rois, 7, 7, 1.0 / 16))
            op._WINDOW_TABLE_MIN_ROIS = 1024
            print("%-12s C %4d R %4d rows-waves %6d  " % (tag, C, R, R * 7 * (C // 256)) + "  ".join("%s %.4f" % kv for kv in res.items()), flush=True)
