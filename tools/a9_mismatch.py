#!/usr/bin/env python3
"""a9 end to end against the reference's OWN rois (tests/golden/proposal_layer.npz): how many rows differ, where the
first difference is, and why -- for every golden case.  The device decodes with exp evaluated in f64 and rounded once,
NumPy with its f32 SIMD exp (~2.5 ulp): coordinates differ in the last bits, and an NMS decision whose IoU sits within
those bits of the threshold flips, which drops or adds one box and shifts every row behind it.

    python3 tools/a9_mismatch.py [--write]     --write: tests/golden/a9_pinned.json (what test_proposal_layer_golden asserts)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = ["res_38x63_train", "res_38x63_test", "vgg_37x62_train", "res_63x100_test"]
STRIDE, SCALES = [16, ], [8, 16, 32]


def iou_f32(a, b):
    """cpu_nms.pyx:43-66 arithmetic (f32) for one pair"""
    f = np.float32
    aa = (a[2] - a[0] + f(1)) * (a[3] - a[1] + f(1))
    ab = (b[2] - b[0] + f(1)) * (b[3] - b[1] + f(1))
    w = max(f(0), min(a[2], b[2]) - max(a[0], b[0]) + f(1))
    h = max(f(0), min(a[3], b[3]) - max(a[1], b[1]) + f(1))
    inter = f(w) * f(h)
    return float(inter / (aa + ab - inter))


def analyse(ref, blob, tol=1e-3):
    """rows of one image: (n_ref, n_got, rows that differ position-wise, first differing row, boxes only in ref, boxes only
    in got, IoU closest to the threshold between a box that flipped and the boxes kept before it)"""
    m = min(len(ref), len(blob))
    close = np.all(np.abs(ref[:m] - blob[:m]) <= tol, axis=1)
    first = int(np.argmin(close)) if not close.all() else -1
    # set difference (greedy matching in order)
    def only(a, b):
        out, j = [], 0
        used = np.zeros(len(b), bool)
        for i in range(len(a)):
            d = np.abs(b - a[i]).max(axis=1)
            d[used] = np.inf
            k = int(np.argmin(d)) if len(b) else -1
            if k >= 0 and d[k] <= tol:
                used[k] = True
            else:
                out.append(i)
        return out
    only_ref, only_got = only(ref, blob), only(blob, ref)
    nearest = None
    for rows, src, other in ((only_ref, ref, blob), (only_got, blob, ref)):
        for i in rows:
            # the box was kept on one side only: on the other side some earlier kept box suppressed it at IoU ~ thresh
            best = None
            for j in range(min(i + 8, len(other))):
                v = iou_f32(src[i, 1:].astype(np.float32), other[j, 1:].astype(np.float32))
                if best is None or abs(v - 0.7) < abs(best - 0.7):
                    best = v
            if best is not None and (nearest is None or abs(best - 0.7) < abs(nearest - 0.7)):
                nearest = best
    return dict(n_ref=int(len(ref)), n_got=int(len(blob)), rows_differing=int((~close).sum()) + abs(len(ref) - len(blob)),
                first_differing_row=first, only_in_reference=len(only_ref), only_in_device=len(only_got),
                iou_of_flipped_pair=(round(nearest, 7) if nearest is not None else None))


def measure():
    from test_gpu_parity import load_golden
    from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer
    g = load_golden("proposal_layer")
    out = {}
    for case in CASES:
        prob, pred, info = g[case + "/prob"], g[case + "/pred"], g[case + "/im_info"]
        train = bool(g[case + "/is_training"])
        blob = proposal_layer(prob, pred, info, train, False, STRIDE, SCALES)
        ref = g[case + "/rois"]
        per = []
        for i in range(prob.shape[0]):
            per.append(analyse(ref[ref[:, 0] == i], blob[blob[:, 0] == i]))
        out[case] = per
    return out


if __name__ == "__main__":
    res = measure()
    print(json.dumps(res, indent=1))
    if "--write" in sys.argv:
        with open(os.path.join(ROOT, "tests", "golden", "a9_pinned.json"), "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
            f.write("\n")
