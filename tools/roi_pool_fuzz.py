#!/usr/bin/env python3
"""Random RoI-pool cases, product against the C oracle, bit for bit: both contracts (f32 + i32 arg-max; the 1-byte training pair
where the shape is supported), both roundings, odd map sizes and channel counts (every dispatch branch: the wave-uniform
kernels, the sliced fall-backs, maps beyond 97 x 104 cells), pooled sizes other than 7 x 7, RoIs that are tiny, huge, partly
or wholly outside the image, with x2 < x1.  (A batch index outside [0, N) is left out: the reference op reads out of
bounds for it; the product's answer for it -- an empty RoI -- is tested in tests/test_gpu_edges.py.)
    python3 tools/roi_pool_fuzz.py [--cases 60] [--seed 0]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import c_oracle  # noqa: E402
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rs = np.random.RandomState(args.seed)
bad = 0
for k in range(args.cases):
    N = int(rs.randint(1, 5))
    H, W = int(rs.choice([5, 13, 37, 38, 63, 97, 110])), int(rs.choice([7, 19, 62, 63, 100, 104, 121]))
    C = int(rs.choice([32, 64, 96, 100, 128, 192, 256, 320, 512, 1024]))
    if H * W * C > 8_000_000:
        C = 64
    ph, pw = [(7, 7), (7, 7), (7, 7), (6, 6), (3, 5), (14, 14), (1, 1)][k % 7]
    R = int(rs.choice([1, 5, 64, 130, 600, 2500])) if k % 4 else int(rs.randint(1, 700))
    if R * ph * pw * C > 40_000_000:
        R = max(1, 40_000_000 // (ph * pw * C))
    im_h, im_w = H * 16, W * 16
    c = rs.uniform(-0.1, 1.1, size=(R, 2)) * [im_w, im_h]
    wh = np.exp(rs.normal(np.log(120), 1.0, size=(R, 2)))
    rois = np.hstack((rs.randint(0, N, (R, 1)), c - wh / 2, c + wh / 2)).astype(np.float32)
    if R > 6:
        rois[0, 1:] = [0, 0, im_w - 1, im_h - 1]                       # the whole image
        rois[1, 1:] = [5, 5, 6, 6]                                     # smaller than a cell
        rois[2, 1:] = [-300, -200, -100, -50]                          # wholly outside
        rois[3, 1:] = [200, 100, 100, 50]                              # x2 < x1
        rois[5, 1:] = [im_w - 3, im_h - 3, im_w + 400, im_h + 400]     # reaching far outside
    rois = rois[np.argsort(rois[:, 0], kind="stable")]
    f = np.maximum(rs.normal(size=(N, H, W, C)), 0).astype(np.float32)
    if k % 3 == 0:
        f[rs.uniform(size=f.shape) < 0.3] = 0.0                         # ties: the first maximum must win
    mode = "cpu" if k % 2 else "cuda"
    et, ea = c_oracle.roi_pool_forward(f, rois, ph, pw, 1.0 / 16, mode, threads=8)
    diff = rs.normal(size=et.shape).astype(np.float32)
    want = c_oracle.roi_pool_backward(diff, ea, rois, f.shape, ph, pw, 1.0 / 16)
    split = [7, 0, 4, 107, 7, 104][k % 6]          # small forward launches: waves per bin row (round 6; results must not depend on it)
    _lib.set_tuning("roi_fwd_one_bin", split)
    tag = "case %d N %d map %dx%dx%d R %d pooled %dx%d %s one_bin %d" % (k, N, H, W, C, R, ph, pw, mode, split)
    top, arg = op.roi_pool(f, rois, ph, pw, 1.0 / 16, rounding=mode)
    ok = np.array_equal(top, et) and np.array_equal(arg, ea)
    g = op.roi_pool_grad(f, rois, arg, diff, ph, pw, 1.0 / 16)
    ok2 = np.array_equal(g, want)
    ok3 = ok4 = ok5 = True
    if op.compact_supported(H, W, C, ph, pw):
        ft, rt, dt = torch.from_numpy(f).cuda(), torch.from_numpy(rois).cuda(), torch.from_numpy(diff).cuda()
        top8, arg8 = op.roi_pool_compact(ft, rt, ph, pw, 1.0 / 16, rounding=mode)
        if op.compact_overflowed(ft.device):
            op._flags(ft.device).flags.zero_()                         # a window beyond 15 x 16 cells: the 1-byte pair refuses it (tested elsewhere)
        else:
            ok3 = np.array_equal(top8.cpu().numpy(), et) and \
                np.array_equal(op.expand_argmax(arg8, rt, f.shape, ph, pw, 1.0 / 16, rounding=mode).cpu().numpy(), ea)
            g8 = op.roi_pool_grad_compact(f.shape, rt, arg8, dt, ph, pw, 1.0 / 16, rounding=mode, segments=1)
            ok4 = np.array_equal(g8.cpu().numpy(), want)
            # the bin-owner form (round 5): two owner plans per case -- every (bin, cell) pair applied exactly once (bit-equal
            # on integer-valued gradients), real-valued ones repeatable and within 1e-5 of the scale
            if ph <= 8 and pw <= 8 and C % 4 == 0:
                n_own = _lib.lib().wssdl_roi_pool_backward_owner_plan_count()
                ints = rs.randint(-8, 9, size=et.shape).astype(np.float32)
                want_i = c_oracle.roi_pool_backward(ints, ea, rois, f.shape, ph, pw, 1.0 / 16)
                it = torch.from_numpy(ints).cuda()
                for o in rs.choice(n_own, size=2, replace=False):
                    plan = op.roi_pool_grad_prepare_owner(f.shape, rt, ph, pw, 1.0 / 16, int(o), rounding=mode)
                    plan.owner_segments = int(rs.randint(1, 5))          # round 6: 1-4 waves per tile stream
                    gi = op.roi_pool_grad_compact(f.shape, rt, arg8, it, ph, pw, 1.0 / 16, rounding=mode, plan=plan)
                    ga = op.roi_pool_grad_compact(f.shape, rt, arg8, dt, ph, pw, 1.0 / 16, rounding=mode, plan=plan)
                    gb = op.roi_pool_grad_compact(f.shape, rt, arg8, dt, ph, pw, 1.0 / 16, rounding=mode, plan=plan)
                    ok5 = ok5 and np.array_equal(gi.cpu().numpy(), want_i) and bool(torch.equal(ga, gb)) and \
                        float(np.abs(ga.cpu().numpy() - want).max()) <= 1e-5 * max(float(np.abs(want).max()), 1e-30)
                    if op.flags_raised():
                        ok5 = False
    if not (ok and ok2 and ok3 and ok4 and ok5):
        bad += 1
        print("MISMATCH %s: i32 forward %s backward %s, 1-byte forward %s backward %s, owner form %s" % (tag, ok, ok2, ok3, ok4, ok5), flush=True)
    if (k + 1) % 20 == 0:
        print("case %d ok so far (%d mismatches)" % (k + 1, bad), flush=True)
print("cases %d mismatches %d" % (args.cases, bad))
sys.exit(1 if bad else 0)
