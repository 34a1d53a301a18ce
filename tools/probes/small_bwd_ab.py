#!/usr/bin/env python3
"""round 6: small backward launches -- lists + walk (prepare + walk, what the host layer does) against the kernel that filters
the RoIs itself (no lists), HIP events around the library calls, outputs compared bit for bit."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op  # noqa: E402


def event_ms(fn, names, iters=30, warmup=5):
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
    return [tl.get(k, {"avg_ms": 0.0})["avg_ms"] for k in names]


base18 = np.load(os.path.join(ROOT, "profiles/roofline_rois_resnet18_sup_b2_r256.npy"))
rng = np.random.RandomState(3)
for N, H, W, C, R in ((2, 38, 63, 256, 256), (2, 38, 63, 1024, 256), (4, 38, 63, 1024, 512), (1, 38, 63, 1024, 128), (8, 38, 63, 1024, 1024),
                      (2, 38, 63, 256, 64), (1, 63, 100, 1024, 300)):
    pick = base18[rng.randint(0, base18.shape[0], R)].copy()
    pick[:, 0] = rng.randint(0, N, R)
    pick[:, 1:] += rng.uniform(-8, 8, (R, 4)).astype(np.float32)
    pick[:, 3] = np.maximum(pick[:, 3], pick[:, 1] + 4); pick[:, 4] = np.maximum(pick[:, 4], pick[:, 2] + 4)
    rois = torch.from_numpy(pick).cuda()
    shape = (N, H, W, C)
    data = torch.randn(shape, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    top, arg8 = op.roi_pool_compact(data, rois, 7, 7, 1.0 / 16)
    diff = torch.randn(top.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(2))
    with _lib.tuned():
        from wssdl_bus_amd.fast_rcnn.config import cfg
        cfg.ROI_POOL_BWD_EXACT = True

        def lists():
            p = op.prepare_backward(shape, rois, 7, 7, 1.0 / 16)
            return op.roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=p, segments=p.segments)

        def nolists():
            return op.roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, use_workspace=False)
        a, b = lists(), nolists()
        assert torch.equal(a, b)
        pl, wl = event_ms(lists, ("roi_pool_backward_prepare", "roi_pool_backward"))
        _, wn = event_ms(nolists, ("roi_pool_backward_prepare", "roi_pool_backward"))
        best = None
        for v in range(6):
            with _lib.tuned(roi_bwdc_variant=v):
                _, t = event_ms(nolists, ("roi_pool_backward_prepare", "roi_pool_backward"), iters=15)
            best = (t, v) if best is None or t < best[0] else best
        cfg.ROI_POOL_BWD_EXACT = False
    print("N %d C %4d R %4d (%dx%d): lists %.4f + walk %.4f = %.4f   no lists %.4f (best variant %d: %.4f)" % (
        N, C, R, H, W, pl, wl, pl + wl, wn, best[1], best[0]), flush=True)
