#!/usr/bin/env python3
"""Stress of the fused mask + sweep launch's hand-over (row-block counters between workgroups): many
random batches (1-8 images, both NMS thresholds, test- and train-sized), each compared with the two-launch
form and with a repeat of itself.  Any timeout of a wait shows up as a zero count.
    python3 tools/nms_fused_stress.py [--rounds 150]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
from kernel_bench import synth_rpn  # noqa: E402
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402
from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=150)
ap.add_argument("--busy", action="store_true", help="keep a second stream busy with large GEMMs while the NMS runs")
args = ap.parse_args()
side = torch.cuda.Stream() if args.busy else None
big = torch.randn((8192, 8192), device="cuda") if args.busy else None
t0 = time.time()
bad = 0
for k in range(args.rounds):
    N = 1 + k % 8
    info = torch.tensor([[600, 1000, 1.0, 1.0]] * N, device="cuda")
    prob, pred = synth_rpn(N, 38, 63, 9, 100 + k)
    pred = pred * [0.0, 0.3, 1.0, 3.0][k % 4]
    cfg.TRAIN.RPN_NMS_THRESH = [0.7, 0.3, 0.5][k % 3]
    with _lib.tuned(nms_fused=0):
        ref = [t.clone() for t in proposal_layer_padded(prob, pred, info, True)]
    for rep in range(3):
        if side is not None:
            with torch.cuda.stream(side):
                for _ in range(3):
                    big @ big                       # ~1 ms each of a full chip, overlapping the launches below
        with _lib.tuned(nms_fused=1):
            out = proposal_layer_padded(prob, pred, info, True)
        if not all(torch.equal(a, b) for a, b in zip(out, ref)):
            bad += 1
            print("MISMATCH round", k, "rep", rep, "N", N, out[1].tolist(), ref[1].tolist(), flush=True)
    if k % 25 == 0:
        print("round", k, "kept", ref[1].tolist(), "elapsed %.1f s" % (time.time() - t0), flush=True)
cfg.TRAIN.RPN_NMS_THRESH = 0.7
print("rounds", args.rounds, "mismatches", bad)
sys.exit(1 if bad else 0)
