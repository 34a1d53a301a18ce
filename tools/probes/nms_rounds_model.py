#!/usr/bin/env python3
"""CPU model (round 6): how many rounds a fixed-point ("decide every box whose higher-scored neighbours are all decided")
form of the greedy NMS would need on the synthetic regimes of tools/nms_async_ab.py, and how sparse the neighbour lists
are.  Greedy NMS has one fixed point: kept(i) <=> no kept j < i with IoU(i, j) >= t; iterating the rule from "unknown" in
parallel reaches it in (longest alternating chain) rounds.  python3 tools/probes/nms_rounds_model.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import np_oracle as O  # noqa: E402  (a model, not the product)


def neighbours(boxes, thr):
    """lists of j < i with iou(i, j) >= thr (f64 rule of cpu_nms.pyx)"""
    n = boxes.shape[0]
    b = boxes.astype(np.float64)
    area = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    out = [None] * n
    for s in range(0, n, 512):
        e = min(n, s + 512)
        xx1 = np.maximum(b[s:e, None, 0], b[None, :e, 0]); yy1 = np.maximum(b[s:e, None, 1], b[None, :e, 1])
        xx2 = np.minimum(b[s:e, None, 2], b[None, :e, 2]); yy2 = np.minimum(b[s:e, None, 3], b[None, :e, 3])
        w = np.maximum(0.0, xx2 - xx1 + 1); h = np.maximum(0.0, yy2 - yy1 + 1)
        inter = w * h
        iou = inter / (area[s:e, None] + area[None, :e] - inter)
        m = iou >= thr
        for i in range(s, e):
            out[i] = np.nonzero(m[i - s, :i])[0]
    return out


def fixed_point(nb, block=None, post=2000):
    n = len(nb)
    st = np.zeros(n, np.int8)       # 0 unknown, 1 kept, 2 removed
    rounds = 0
    work = []
    while (st == 0).any():
        und = np.nonzero(st == 0)[0]
        new = st.copy()
        w = 0
        for i in und:
            s = st[nb[i]]
            w += len(s)
            if (s == 1).any():
                new[i] = 2
            elif (s == 2).all():
                new[i] = 1
        st = new
        rounds += 1
        work.append((len(und), w))
    return st, rounds, work


def main():
    g = torch.Generator().manual_seed(3)
    N, H, W, A = 2, 38, 63, 9
    logits = torch.randn((N, H, W, A, 2), generator=g)
    p = torch.softmax(logits, dim=-1)
    prob = torch.cat((p[..., 0], p[..., 1]), dim=-1).numpy()
    pred0 = (0.2 * torch.randn((N, H, W, 4 * A), generator=g)).numpy()
    base = O.generate_anchors(scales=np.array((8, 16, 32)))
    anchors = O.shifted_anchors(H, W, 16, base)
    info = np.array([600, 1000, 1.0], np.float32)
    for scale, thr in ((1.0, 0.7), (0.5, 0.5), (0.3, 0.3)):
        for i in range(N):
            st = O.proposal_stages_one_image(prob[i], pred0[i] * scale, info, anchors, A, 12000, 2000, thr, 16)
            boxes = st["sorted_boxes"]
            nb = neighbours(boxes, thr)
            nnz = sum(len(x) for x in nb)
            stat, rounds, work = fixed_point(nb)
            kept = np.nonzero(stat == 1)[0]
            assert np.array_equal(kept[:2000], st["keep"]), "fixed point != greedy"
            last = kept[min(len(kept), 2000) - 1]
            # rounds needed if only the prefix up to the 2000th kept box is iterated
            _, rounds_prefix, work_p = fixed_point(nb[:last + 1])
            print("scale %.1f thr %.1f image %d: n %d nnz %d (%.1f per box, max %d) kept %d (2000th at %d) rounds %d (prefix %d) "
                  "undecided per round %s" % (scale, thr, i, len(nb), nnz, nnz / len(nb), max(len(x) for x in nb), len(kept), last,
                                              rounds, rounds_prefix, [u for u, _ in work][:40]), flush=True)
            print("   entries read per round", [w for _, w in work][:40], flush=True)


if __name__ == "__main__":
    main()
