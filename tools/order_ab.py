#!/usr/bin/env python3
"""Proposal layer with the two orderings of its candidates (tuning topk_sort: 1 sorted runs + cross ranks,
0 select + sample sort), same inputs, HIP events around whole calls.  (profiles/r03_order_ab.log also has the
device-wide library sort that was measured and removed, as topk_sort 2.)

    python tools/order_ab.py [--iters 30]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
args = ap.parse_args()

rs = np.random.RandomState(5)
for N, (H, W), train in ((8, (38, 63), True), (2, (38, 63), True), (1, (63, 100), False), (1, (38, 63), False)):
    A = 9
    prob = torch.from_numpy(rs.uniform(0.01, 0.99, size=(N, H, W, 2 * A)).astype(np.float32)).cuda()
    pred = torch.from_numpy(rs.normal(0, 0.5, size=(N, H, W, 4 * A)).astype(np.float32)).cuda()
    info = torch.from_numpy(np.tile(np.array([[16 * H - 8, 16 * W - 8, 1.0, 1]], np.float32), (N, 1))).cuda()
    line = {"images": N, "anchors": H * W * A, "train": train}
    ref = None
    for mode in (1, 0):
        with _lib.tuned(topk_sort=mode):
            out = proposal_layer_padded(prob, pred, info, train)
            blob = out[0].clone()
            if ref is None:
                ref = blob
            assert torch.equal(ref, blob), (N, H, W, mode)
            for _ in range(3):
                proposal_layer_padded(prob, pred, info, train)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for _ in range(args.iters):
                proposal_layer_padded(prob, pred, info, train)
            e.record()
            torch.cuda.synchronize()
            line["ms_topk_sort_%d" % mode] = round(s.elapsed_time(e) / args.iters, 4)
    print(json.dumps(line), flush=True)
