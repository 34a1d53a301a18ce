#!/usr/bin/env python3
"""A/B of the proposal layer with the walking NMS (fused launch) and the grid + fixed-point NMS (wssdl_set_tuning nms_grid),
synthetic RPN outputs at three overlap regimes, outputs compared bit for bit.   python3 tools/nms_grid_ab.py [--images 8]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from kernel_bench import synth_rpn, timeit  # noqa: E402
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402
from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=8)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()
N = args.images
info = torch.tensor([[600, 1000, 1.0, 1.0]] * N, device="cuda")
prob, pred0 = synth_rpn(N, 38, 63, 9, 3)
for scale, thresh in ((1.0, 0.7), (0.5, 0.7), (0.3, 0.7), (0.1, 0.7), (1.0, 0.8), (1.0, 0.6)):
    cfg.TRAIN.RPN_NMS_THRESH = thresh
    pred = pred0 * scale
    ref = None
    for rep in range(args.reps):
        for grid in (0, 1):
            with _lib.tuned(nms_grid=grid):
                out = proposal_layer_padded(prob, pred, info, True)
                torch.cuda.synchronize()
                if ref is None:
                    ref = [t.clone() for t in out]
                same = all(torch.equal(a, b) for a, b in zip(out, ref))
                ms = timeit(lambda: proposal_layer_padded(prob, pred, info, True), args.iters, warmup=5)
            print(json.dumps(dict(nms_grid=grid, proposal_layer_ms=round(ms, 4), images=N, pred_scale=scale, nms_thresh=thresh,
                                  same_as_walk=same, kept=[int(v) for v in out[1].tolist()][:4])), flush=True)
            assert same, (scale, thresh, grid)
cfg.TRAIN.RPN_NMS_THRESH = 0.7
