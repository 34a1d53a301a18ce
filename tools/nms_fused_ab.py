#!/usr/bin/env python3
"""A/B of the proposal layer's one-pass NMS: fused mask + sweep launch against the two launches
(wssdl_set_tuning nms_fused), 8 images x 12000 candidates -> 2000, synthetic RPN maps.
    python3 tools/nms_fused_ab.py [--images 8] [--iters 30]"""
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
from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=8)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--pred-scale", type=float, default=1.0,
                help="scale of the box deltas: 0 = the anchors themselves (heavy suppression: the sweep walks every chunk)")
ap.add_argument("--nms-thresh", type=float, default=0.7, help="0.3 with --pred-scale 0.3: fewer than 2000 boxes survive, the sweep walks every chunk")
args = ap.parse_args()
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402
cfg.TRAIN.RPN_NMS_THRESH = args.nms_thresh
N = args.images
info = torch.tensor([[600, 1000, 1.0, 1.0]] * N, device="cuda")
prob, pred = synth_rpn(N, 38, 63, 9, 3)
pred = pred * args.pred_scale
ref = None
for rep in range(2):
    for fused, sparse in ((0, 0), (1, 0), (1, 8), (1, 16), (1, 24), (1, 64)):
        with _lib.tuned(nms_fused=fused, nms_sparse=sparse):
            out = proposal_layer_padded(prob, pred, info, True)
            if ref is None:
                ref = [t.clone() for t in out]
            assert all(torch.equal(a, b) for a, b in zip(out, ref))
            ms = timeit(lambda: proposal_layer_padded(prob, pred, info, True), args.iters, warmup=5)
        print(json.dumps(dict(nms_fused=fused, nms_sparse=sparse, proposal_layer_ms=round(ms, 4), images=N, pred_scale=args.pred_scale, nms_thresh=args.nms_thresh, kept=[int(v) for v in out[1].tolist()])))
