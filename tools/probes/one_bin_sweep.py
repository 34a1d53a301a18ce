#!/usr/bin/env python3
"""round 6: where does one wave per bin (roi_fwd_one_bin = 7) beat the sliced kernel on forward launches too small for the
rows kernel?  R x C sweep, RoIs drawn from two committed proposal sets (train-sized windows / test-sized windows), HIP
events around the library call (_lib.timeline), outputs compared bit for bit."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.roi_pooling_layer.roi_pooling_op import roi_pool_compact  # noqa: E402


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
    return tl["roi_pool_forward"]["avg_ms"]


train = np.load(os.path.join(ROOT, "profiles/roofline_rois_resnet50_joint_b8.npy")) if os.path.exists(
    os.path.join(ROOT, "profiles/roofline_rois_resnet50_joint_b8.npy")) else None
sets = {"test-sized (R-101 set, 16 cells per bin)": (np.load(os.path.join(ROOT, "profiles/roofline_rois_resnet101_1600_test_r300.npy")), (63, 100)),
        "train-sized (R-18 set, 7 cells per bin)": (np.load(os.path.join(ROOT, "profiles/roofline_rois_resnet18_sup_b2_r256.npy")), (38, 63))}
rng = np.random.RandomState(3)
for tag, (base, (H, W)) in sets.items():
    for C in (256, 512, 1024, 2048):
        for R in (32, 128, 300, 600, 1000):
            pick = base[rng.randint(0, base.shape[0], R)].copy()
            pick[:, 1:] += rng.uniform(-8, 8, (R, 4)).astype(np.float32)
            pick[:, 3] = np.maximum(pick[:, 3], pick[:, 1] + 4); pick[:, 4] = np.maximum(pick[:, 4], pick[:, 2] + 4)
            N = int(base[:, 0].max()) + 1
            rois = torch.from_numpy(pick).cuda()
            data = torch.randn((N, H, W, C), device="cuda", generator=torch.Generator("cuda").manual_seed(1))
            res = {}
            ref = None
            for one in (0, 107):
                with _lib.tuned(roi_fwd_one_bin=one):
                    top, arg = roi_pool_compact(data, rois, 7, 7, 1.0 / 16)
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = (top.clone(), arg.clone())
                    assert torch.equal(top, ref[0]) and torch.equal(arg, ref[1]), (tag, C, R, one)
                    res[one] = event_ms(lambda: roi_pool_compact(data, rois, 7, 7, 1.0 / 16))
            print("%-42s C %4d R %4d waves(rows) %6d  sliced/rows %.4f  one-bin %.4f  ratio %.2f" % (
                tag, C, R, R * 7 * ((C + 255) // 256), res[0], res[107], res[107] / res[0]), flush=True)
