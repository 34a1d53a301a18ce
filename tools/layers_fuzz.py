#!/usr/bin/env python3
"""Random anchor-target, proposal-target and IoU cases, product against the NumPy oracle (oracle/np_oracle.py, itself pinned by the reference's
goldens at three map shapes): random map shapes and image sizes, 1 ... 20 ground-truth rows of classes 0 / 1 / 2 (boxes partly or
wholly outside the image, tiny ones, duplicates), datasets SNUBH / SNUBH_FG, with the reference's NumPy RNG stream for the
sub-sampling.  Labels, sampled rows and weights bit for bit, regression targets within 1 ulp (anchors) / 4 ulp (RoIs: np.log in f32), IoU tables
bit for bit.
    python3 tools/layers_fuzz.py [--cases 60] [--seed 0]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from oracle import np_oracle as O, ref_kernels  # noqa: E402
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402
from wssdl_bus_amd.rpn_msr.anchor_target_layer_tf_bus import anchor_target_layer  # noqa: E402
from wssdl_bus_amd.rpn_msr.proposal_target_layer_tf_bus import proposal_target_layer, proposal_target_layer_joint  # noqa: E402
from wssdl_bus_amd.rpn_msr.anchor_target_layer_tf_bus import anchor_target_layer_joint  # noqa: E402
from wssdl_bus_amd.utils.cython_bbox import bbox_overlaps  # noqa: E402
from wssdl_bus_amd.utils.cython_bbox_ui import bbox_overlaps_ui  # noqa: E402

MAX_GT = 20


def ulp(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    a = np.where(a < 0, np.int64(-2 ** 31) - a, a)
    b = np.where(b < 0, np.int64(-2 ** 31) - b, b)
    return np.abs(a - b)


ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rs = np.random.RandomState(args.seed)
cfg.SAMPLING_RNG = "reference"
bad = 0
for k in range(args.cases):
    H, W = int(rs.randint(14, 71)), int(rs.randint(14, 111))       # (smaller maps: no anchor inside the image, the reference's argmax of an empty table raises)
    im_h, im_w = H * 16 - int(rs.randint(0, 16)), W * 16 - int(rs.randint(0, 16))
    n = int(rs.randint(1, MAX_GT + 1))
    rows = []
    for j in range(n):
        bw, bh = float(np.exp(rs.normal(np.log(150), 0.8))), float(np.exp(rs.normal(np.log(130), 0.8)))
        x1, y1 = rs.uniform(-0.15, 1.0) * im_w, rs.uniform(-0.15, 1.0) * im_h
        rows.append([x1, y1, x1 + bw, y1 + bh, int(rs.choice([0, 1, 1, 2]))])
    if n >= 3 and k % 4 == 0:
        rows[1] = list(rows[0])                               # a duplicate box
        rows[2][:4] = [10.0, 12.0, 14.0, 15.0]                # a tiny one
    if not any(r[4] > 0 for r in rows):
        rows[0][4] = 1
    gt = np.zeros((MAX_GT, 5), np.float32)
    gt[:n] = np.asarray(rows, np.float32)
    ng = np.array([n], np.int32)
    ii = np.array([[im_h, im_w, 1.0, 1]], np.float32)
    ds = "SNUBH_FG" if k % 3 == 0 else "SNUBH"
    score = np.zeros((1, H, W, 18), np.float32)
    seed = 1000 + k
    want = O.anchor_target_layer(score, gt[None], ng, ii, None, (16,), (8, 16, 32), ds, rng=np.random.RandomState(seed))
    got = anchor_target_layer(score, gt[None], ng, ii, None, [16], [8, 16, 32], ds, rng=np.random.RandomState(seed))
    ok = np.array_equal(np.asarray(got[0]), want[0]) and ulp(got[1], want[1]).max() <= 1 and \
        np.array_equal(np.asarray(got[2]), want[2]) and np.array_equal(np.asarray(got[3]), want[3])
    # IoU tables on the same ground truth against random boxes
    nb = int(rs.randint(1, 3000))
    xy = rs.uniform(-50, max(im_w, im_h), size=(nb, 2))
    boxes = np.ascontiguousarray(np.hstack((xy, xy + rs.uniform(0, 300, size=(nb, 2)))))
    q = np.ascontiguousarray(gt[:n, :4].astype(np.float64))
    ok_iou = np.array_equal(np.asarray(bbox_overlaps(boxes, q)), O.bbox_overlaps(boxes, q)) and \
        np.array_equal(np.asarray(bbox_overlaps_ui(boxes, q)), O.bbox_overlaps_ui(boxes, q))
    if ref_kernels.available():          # the reference's own compiled bbox.pyx / bbox_ui.pyx (oracle/_ref)
        ok_iou = ok_iou and np.array_equal(np.asarray(bbox_overlaps(boxes, q)), ref_kernels.bbox_overlaps(boxes, q)) and \
            np.array_equal(np.asarray(bbox_overlaps_ui(boxes, q)), ref_kernels.bbox_overlaps_ui(boxes, q))
    # proposal targets (alternating mode, train / weak / test) of N images: proposals around the ground truth and elsewhere
    Ni = int(rs.randint(1, 4))
    gts = np.zeros((Ni, MAX_GT, 5), np.float32)
    ngs = np.zeros((Ni,), np.int32)
    rl = []
    for i in range(Ni):
        m = int(rs.randint(1, n + 1))
        gts[i, :m] = gt[rs.permutation(n)[:m]]
        if not (gts[i, :m, 4] > 0).any():
            gts[i, 0, 4] = 1
        ngs[i] = m
        R = int(rs.randint(1, 400))
        src = gts[i, rs.randint(0, m, R), :4]
        jit = src + rs.normal(0, 25, size=(R, 4)) * (rs.uniform(size=(R, 1)) < 0.7)
        far = rs.uniform(0, 1, size=(R, 4)) * [im_w, im_h, im_w, im_h]
        b = np.where(rs.uniform(size=(R, 1)) < 0.6, jit, np.hstack((np.minimum(far[:, :2], far[:, 2:]), np.maximum(far[:, :2], far[:, 2:]))))
        rl.append(np.hstack((np.full((R, 1), i), b)).astype(np.float32))
    rois = np.concatenate(rl)
    ok_pt = True
    for tr, ws in ((True, False), (True, True), (False, False)):
        w_ = O.proposal_target_layer(rois, gts, ngs, 3, tr, ws, rng=np.random.RandomState(seed))
        g_ = proposal_target_layer(rois, gts, ngs, 3, tr, ws, rng=np.random.RandomState(seed))
        for j in range(5):
            a, e = np.asarray(g_[j]), w_[j]
            ok_pt = ok_pt and a.shape == e.shape and (ulp(a, e).max() <= 4 if j == 2 and a.size else np.array_equal(a, e))
    # combined mode: the first n_s images supervised, the others weak (rois appended only / all-ignore anchor labels)
    n_s = int(rs.randint(1, Ni + 1))
    old_ims = (cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH)
    cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = n_s, Ni - n_s
    oc = dict(IMS_PER_BATCH=n_s, WS_IMS_PER_BATCH=Ni - n_s)
    try:
        for tr in (True, False):
            w_ = O.proposal_target_layer_joint(rois, gts, ngs, 3, tr, rng=np.random.RandomState(seed), cfg=oc)
            g_ = proposal_target_layer_joint(rois, gts, ngs, 3, tr, rng=np.random.RandomState(seed))
            for j in range(5):
                a, e = np.asarray(g_[j]), w_[j]
                ok_pt = ok_pt and a.shape == e.shape and (ulp(a, e).max() <= 4 if j == 2 and a.size else np.array_equal(a, e))
        iis = np.repeat(ii, Ni, axis=0)
        scoreN = np.zeros((Ni, H, W, 18), np.float32)
        for tr in (True, False):
            sc_in = scoreN if tr else scoreN[:n_s]
            w_ = O.anchor_target_layer_joint(sc_in, gts[:len(sc_in)], ngs[:len(sc_in)], iis[:len(sc_in)], None, tr, (16,), (8, 16, 32), ds,
                                             rng=np.random.RandomState(seed), cfg=oc)
            g_ = anchor_target_layer_joint(sc_in, gts[:len(sc_in)], ngs[:len(sc_in)], iis[:len(sc_in)], None, tr, [16], [8, 16, 32], ds,
                                           rng=np.random.RandomState(seed))
            ok = ok and np.array_equal(np.asarray(g_[0]), w_[0]) and ulp(g_[1], w_[1]).max() <= 1 and \
                np.array_equal(np.asarray(g_[2]), w_[2]) and np.array_equal(np.asarray(g_[3]), w_[3])
    finally:
        cfg.TRAIN.IMS_PER_BATCH, cfg.TRAIN.WS_IMS_PER_BATCH = old_ims
    if not (ok and ok_iou and ok_pt):
        bad += 1
        print("MISMATCH case %d map %dx%d image %dx%d gt %d %s: anchor targets %s, IoU %s, proposal targets %s" % (k, H, W, im_h, im_w, n, ds, ok, ok_iou, ok_pt), flush=True)
    if (k + 1) % 20 == 0:
        print("case %d ok so far (%d mismatches)" % (k + 1, bad), flush=True)
print("cases %d mismatches %d" % (args.cases, bad))
sys.exit(1 if bad else 0)
