#!/usr/bin/env python3
"""Random post-detection cases (f3: per-class score filter, NMS 0.3, max_per_image cap; fast_rcnn/test_bus.py:360-401): the one-call
device op and the step-by-step form against the NumPy formulas with the oracle's NMS -- random row counts (0 ... 2500), 2 ... 8
classes, score thresholds, caps (none, small, larger than the survivors), pairwise distinct scores.
    python3 tools/post_detect_fuzz.py [--cases 60] [--seed 0]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import np_oracle as O  # noqa: E402
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402
from wssdl_bus_amd.fast_rcnn.test_bus import postprocess_detections  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rs = np.random.RandomState(args.seed)
bad = 0
for k in range(args.cases):
    R = int(rs.choice([0, 1, 2, 63, 64, 65, 300, 1000, 2500])) if k % 3 else int(rs.randint(1, 1200))
    K = int(rs.randint(2, 9))
    thresh = float(rs.choice([0.0, 0.05, 0.3, 0.6]))
    cap = int(rs.choice([0, 1, 5, 40, 100, 5000]))
    ctr = rs.uniform(50, 900, size=(R, 1, 2)) * [1.0, 0.6] + rs.normal(0, 6, size=(R, K, 2))
    wh = rs.uniform(20, 260, size=(R, K, 2))
    boxes = np.concatenate((ctr - wh / 2, ctr + wh / 2), axis=2).reshape(R, 4 * K).astype(np.float32)
    scores = ((rs.permutation(R * K) + 1).astype(np.float32) / np.float32(R * K + 1)).reshape(R, K)      # pairwise distinct
    want = {}
    for j in range(1, K):
        inds = np.where(scores[:, j] > thresh)[0]
        d = np.hstack((boxes[inds, 4 * j:4 * j + 4], scores[inds, j:j + 1])).astype(np.float32)
        want[j] = d[O.nms(d, cfg.TEST.NMS)] if len(d) else d
    alls = np.hstack([want[j][:, 4] for j in range(1, K)])
    if cap > 0 and len(alls) > cap:
        th = np.sort(alls)[-cap]
        for j in range(1, K):
            want[j] = want[j][want[j][:, 4] >= th]
    st, bt = torch.from_numpy(scores).cuda(), torch.from_numpy(boxes).cuda()
    ok = True
    for fused in (True, False):
        old = cfg.TEST.FUSED_POST_DETECTIONS
        try:
            cfg.TEST.FUSED_POST_DETECTIONS = fused
            got = postprocess_detections(st, bt, K, thresh=thresh, max_per_image=cap)
        finally:
            cfg.TEST.FUSED_POST_DETECTIONS = old
        for j in range(1, K):
            ok = ok and np.array_equal(got[j].cpu().numpy().reshape(-1, 5), want[j].reshape(-1, 5))
    if not ok:
        bad += 1
        print("MISMATCH case %d R %d K %d thresh %.2f cap %d" % (k, R, K, thresh, cap), flush=True)
    if (k + 1) % 20 == 0:
        print("case %d ok so far (%d mismatches)" % (k + 1, bad), flush=True)
print("cases %d mismatches %d" % (args.cases, bad))
sys.exit(1 if bad else 0)
