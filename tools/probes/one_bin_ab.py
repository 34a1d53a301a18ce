#!/usr/bin/env python3
"""round 6: small forward launches, one wave per bin (roi_fwd_one_bin = 1) against the sliced kernel: same bits, time."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kernel_bench import timeit  # noqa: E402
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.roi_pooling_layer.roi_pooling_op import roi_pool_compact  # noqa: E402

for name, path, (H, W, C) in (("r101", "profiles/roofline_rois_resnet101_1600_test_r300.npy", (63, 100, 1024)),
                              ("r18", "profiles/roofline_rois_resnet18_sup_b2_r256.npy", (38, 63, 256)),
                              ("r50_128", None, (38, 63, 1024))):
    if path:
        rois = torch.from_numpy(np.load(os.path.join(ROOT, path))).cuda()
    else:
        g = torch.Generator().manual_seed(5)
        x1 = torch.rand(128, generator=g) * 800; y1 = torch.rand(128, generator=g) * 500
        rois = torch.stack((torch.zeros(128), x1, y1, x1 + 20 + torch.rand(128, generator=g) * 300,
                            y1 + 20 + torch.rand(128, generator=g) * 200), 1).cuda()
    N = int(rois[:, 0].max().item()) + 1
    data = torch.randn((N, H, W, C), device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    ref = None
    for rep in range(2):
        for one in (0, 7):
            with _lib.tuned(roi_fwd_one_bin=one):
                top, arg = roi_pool_compact(data, rois, 7, 7, 1.0 / 16)
                torch.cuda.synchronize()
                if ref is None:
                    ref = (top.clone(), arg.clone())
                assert torch.equal(top, ref[0]) and torch.equal(arg, ref[1]), (name, one)
                ms = timeit(lambda: roi_pool_compact(data, rois, 7, 7, 1.0 / 16), 50, warmup=5)
            print(name, "R", rois.shape[0], "one_bin", one, "ms %.4f" % ms, flush=True)
