#!/usr/bin/env python3
"""Random multi-task-loss cases (a13), the fused device op against the oracle's f64 evaluation (values within north_star's 1e-5) and
against the chain of torch ops under autograd (gradients): random batch / map / anchor / row counts, weak images, padding rows.
    python3 tools/loss_fuzz.py [--cases 40] [--seed 0]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import test_gpu_loss as T  # noqa: E402
from wssdl_bus_amd.fast_rcnn import train_bus as TB  # noqa: E402
from wssdl_bus_amd.fast_rcnn.loss_op import multi_task_loss  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=40)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rs0 = np.random.RandomState(args.seed)
bad = 0
for k in range(args.cases):
    N = int(rs0.randint(1, 9))
    H, W, A = int(rs0.randint(3, 41)), int(rs0.randint(3, 65)), int(rs0.choice([3, 9]))
    weak_from = None if (N == 1 or k % 3 == 0) else int(rs0.randint(1, N))
    n_sup = N if weak_from is None else weak_from
    n_rows = int(rs0.randint(1, 129)) * n_sup
    pad = int(rs0.randint(0, 12)) if k % 4 == 1 else 0
    rows_total = n_rows + pad + (0 if weak_from is None else int(rs0.randint(1, 3000)))
    n_rows += pad
    rs = np.random.RandomState(1000 + k)
    rpn_cls, rpn_box, cls, box, rpn_data, roi_data = T._inputs(torch, rs, N, H, W, A, n_rows, rows_total, weak_from=weak_from, pad_rows=pad)
    if k % 3 == 2:
        # a trained network's regime: confident, correct scores -> cross-entropies of 1e-2 ... 1e-4, where f32 needs the log1p form
        with torch.no_grad():
            lab = roi_data[1].reshape(-1).long()
            rows = torch.nonzero(lab >= 0).reshape(-1)
            cls[rows, lab[rows]] += float(rs.uniform(6, 12))
            L = rpn_data[0].reshape(N, A, H, W).permute(0, 2, 3, 1)             # [N,H,W,A] labels
            boost = float(rs.uniform(6, 12))
            rpn_cls[..., :A] += boost * (L == 0)
            rpn_cls[..., A:] += boost * (L == 1)
    terms = multi_task_loss(rpn_cls, rpn_box, cls, box, rpn_data, roi_data, n_sup)
    want = T._oracle_terms(rpn_cls, rpn_box, cls, box, rpn_data, roi_data, n_sup)
    got = terms.detach().cpu().numpy().astype(np.float64)
    ok_v = bool(np.allclose(got, want, rtol=1e-5, atol=2e-9))            # (relative down to tiny losses)
    wts = torch.tensor(rs.uniform(0.2, 2.0, 4), dtype=torch.float32, device="cuda")
    (terms * wts).sum().backward()
    fused = [x.grad.clone() for x in (rpn_cls, rpn_box, cls, box)]
    for x in (rpn_cls, rpn_box, cls, box):
        x.grad = None
    # the chain of torch ops the op replaces, in f64 under autograd: the yardstick for the gradients (the same chain in f32
    # computes the label's component as p - 1, which cancels when p -> 1)
    d = [x.detach().double().requires_grad_(True) for x in (rpn_cls, rpn_box, cls, box)]
    n, h, w, c = rpn_cls.shape
    reshaped = d[0].permute(0, 3, 1, 2).reshape(n, 2, A * h, w).permute(0, 2, 3, 1)
    rd64 = tuple(t.double() if t.is_floating_point() else t for t in rpn_data)
    ro64 = tuple(t.double() if t.is_floating_point() else t for t in roi_data)
    ref = torch.stack([TB.rpn_cls_loss(reshaped, rd64[0]), TB.rpn_box_loss(d[1], rd64, n_sup),
                       TB.rcnn_cls_loss(d[2], ro64[1]), TB.rcnn_box_loss(d[3], ro64)])
    (ref * wts.double()).sum().backward()
    ok_g = True
    for f, x in zip(fused, d):
        g = x.grad
        scale = float(g.abs().max().clamp_min(1e-30))
        ok_g = ok_g and float((f.double() - g).abs().max()) <= 2e-5 * scale + 1e-12
    if not (ok_v and ok_g):
        bad += 1
        print("MISMATCH case %d N %d map %dx%d A %d rows %d of %d weak_from %s pad %d: values %s (%s vs %s) gradients %s" % (
            k, N, H, W, A, n_rows, rows_total, weak_from, pad, ok_v, got, want, ok_g), flush=True)
    if (k + 1) % 10 == 0:
        print("case %d ok so far (%d mismatches)" % (k + 1, bad), flush=True)
print("cases %d mismatches %d" % (args.cases, bad))
sys.exit(1 if bad else 0)
