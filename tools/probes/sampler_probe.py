#!/usr/bin/env python3
"""Device RoI sampler (wssdl_proposal_target_device) on synthetic proposal blobs of the bench's shapes; run it
under `rocprofv3 --kernel-trace --stats` to read roi_sample_kernel's duration per shape."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402
from wssdl_bus_amd.rpn_msr import proposal_target_layer_tf_bus as ptl  # noqa: E402

cfg.SAMPLING_RNG = "device"
rs = np.random.RandomState(0)
for n_images, per_image, sampled in ((8, 1064, 4), (2, 2000, 2), (3, 1376, 1), (1, 2000, 1)):
    rows = []
    for i in range(n_images):
        c = rs.uniform(50, 550, size=(per_image, 2))
        wh = rs.uniform(20, 300, size=(per_image, 2))
        rows.append(np.hstack((np.full((per_image, 1), i), c - wh / 2, c + wh / 2)))
    rois = torch.from_numpy(np.vstack(rows).astype(np.float32)).cuda()
    gt = torch.zeros((n_images, 20, 5), device="cuda")
    gt[:, 0] = torch.tensor([100.0, 80.0, 380.0, 300.0, 1.0], device="cuda")
    gt[:, 1] = torch.tensor([300.0, 260.0, 560.0, 520.0, 1.0], device="cuda")
    ng = torch.full((n_images,), 2, dtype=torch.int32, device="cuda")
    for _ in range(5):
        out = ptl._supervised_device(rois, gt, ng, list(range(sampled)), True, 2)
    torch.cuda.synchronize()
    lab = out[1].cpu().numpy().ravel()
    print(n_images, per_image, sampled, "fg", int((lab > 0).sum()), "bg", int((lab == 0).sum()), "pad", int((lab < 0).sum()), flush=True)
