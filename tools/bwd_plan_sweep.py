#!/usr/bin/env python3
"""Times every plan of the list-driven RoI-pool backward (tile shape x records in flight) on the proposals of
synthetic RPN maps, for a given launch size -- the data behind the plan the library picks by itself.

    python3 tools/bwd_plan_sweep.py --images 2 --channels 256 [--keep 128] [--plans 11,5,18,19]
"""
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
from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op  # noqa: E402
from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import compact_rois, proposal_layer_padded  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--keep", type=int, default=0, help="RoIs kept per image (0 = all proposals)")
    ap.add_argument("--plans", default="auto,11,5,18,19,23,22,21")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    N, H, W, C = args.images, 38, 63, args.channels
    info = torch.tensor([[600, 1000, 1.0, 1.0]] * N, device="cuda")
    prob, pred = synth_rpn(N, H, W, 9, 3)
    rois = compact_rois(*proposal_layer_padded(prob, pred, info, True))
    if args.keep:
        b = rois[:, 0]
        rois = torch.cat([rois[b == i][:args.keep] for i in range(N)])
    feat = torch.relu(torch.randn((N, H, W, C), device="cuda"))
    top, arg8 = op.roi_pool_compact(feat, rois, 7, 7, 1.0 / 16)
    diff = torch.randn_like(top)
    ref = None
    for p in args.plans.split(","):
        _lib.set_tuning("roi_bwd_plan", -1 if p == "auto" else int(p))
        plan = op.roi_pool_grad_prepare(tuple(feat.shape), rois, 7, 7, 1.0 / 16)
        g = op.roi_pool_grad_compact(tuple(feat.shape), rois, arg8, diff, 7, 7, 1.0 / 16, plan=plan)
        ref = g if ref is None else ref
        assert torch.equal(g, ref), p
        ms_prep = timeit(lambda: op.roi_pool_grad_prepare(tuple(feat.shape), rois, 7, 7, 1.0 / 16), args.iters)
        ms = timeit(lambda: op.roi_pool_grad_compact(tuple(feat.shape), rois, arg8, diff, 7, 7, 1.0 / 16, plan=plan),
                    args.iters)
        print(json.dumps(dict(plan=p, picked=plan.plan, N=N, C=C, R=int(rois.shape[0]), walk_ms=round(ms, 4),
                              prepare_ms=round(ms_prep, 4))))
    _lib.set_tuning("roi_bwd_plan", -1)


if __name__ == "__main__":
    main()
