#!/usr/bin/env python3
"""Random cases for the block-table RoI-pool forward (csrc/roi_pool_blocks.hip) against the C oracle, bit for bit: random
map sizes / channel counts the form takes, 1024-3000 RoIs from tiny to larger than the image (inside the 1-byte code's
range), both roundings, maps with ties (ReLU zeros, a handful of distinct values), negative maps, maps with cells the
reference's scan never takes (NaN, +-inf, -FLT_MAX) and with -0.0; bin rows sorted or in RoI order, one or two waves per
bin row.  top is compared as bits, the arg-max after expansion to the reference's i32 indices.
    python3 tools/roi_blocks_fuzz.py [--cases 40] [--seed 0]"""
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
ap.add_argument("--cases", type=int, default=40)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rs = np.random.RandomState(args.seed)
L = _lib.lib()
bad = 0
for k in range(args.cases):
    N = int(rs.randint(1, 5))
    H, W = int(rs.randint(8, 98)), int(rs.randint(8, 105))
    C = int(rs.choice([256, 256, 512, 1024, 2048]))
    while N * H * W * C > 12_000_000:
        N = max(1, N - 1)
        if N == 1:
            H, W = max(8, H // 2), max(8, W // 2)
    R = int(rs.randint(1024, 3001))
    R = max(1024, min(R, 170_000_000 // (49 * C)))
    assert L.wssdl_roi_pool_forward_blocks_bytes(R, N, H, W, C, 7, 7) > 0, (R, N, H, W, C)
    im_h, im_w = H * 16, W * 16
    style = k % 4
    c = rs.uniform(-0.05, 1.05, size=(R, 2)) * [im_w, im_h]
    scale = [60, 200, 400, 120][style]
    wh = np.exp(rs.normal(np.log(scale), [0.3, 0.5, 0.6, 1.2][style], size=(R, 2)))
    x1y1 = np.clip(c - wh / 2, -40, None)
    x2y2 = np.minimum(c + wh / 2, [im_w + 40, im_h + 40])
    rois = np.hstack((rs.randint(0, N, (R, 1)), x1y1, x2y2)).astype(np.float32)
    rois[0, 1:] = [0, 0, im_w - 1, im_h - 1]
    rois[1, 1:] = [5, 5, 6, 6]
    rois[2, 1:] = [-300, -200, -100, -50]
    rois[3, 1:] = [200, 100, 100, 50]
    if k % 2:
        rois = rois[np.argsort(rois[:, 0], kind="stable")]
    shape = (N, H, W, C)
    kind = ["relu", "few values", "negative", "odd cells", "minus zero", "normal"][k % 6]
    f = rs.normal(size=shape).astype(np.float32)
    if kind == "relu":
        f = np.maximum(f, 0)
    elif kind == "few values":
        f = rs.randint(-2, 3, size=shape).astype(np.float32) * np.float32(0.5)
    elif kind == "negative":
        f = -np.abs(f) - np.float32(1.0)
    elif kind == "odd cells":
        p = rs.uniform(size=shape)
        f[p < 0.15] = np.nan
        f[(p >= 0.15) & (p < 0.3)] = -np.inf
        f[(p >= 0.3) & (p < 0.45)] = -np.finfo(np.float32).max
        f[(p >= 0.45) & (p < 0.47)] = np.inf
        f[:, : H // 3] = np.nan
    elif kind == "minus zero":
        f = np.maximum(f, 0)
        z = f == 0
        f[z] = np.where(rs.uniform(size=int(z.sum())) < 0.5, np.float32(-0.0), np.float32(0.0))
    mode = "cpu" if (k // 2) % 2 else "cuda"
    et, ea = c_oracle.roi_pool_forward(f, rois, 7, 7, 1.0 / 16, mode, threads=16)
    ft, rt = torch.from_numpy(f).cuda(), torch.from_numpy(rois).cuda()
    sort, parts = int(rs.randint(0, 2)), int(rs.randint(1, 3))
    with _lib.tuned(roi_fwd_blocks=1, roi_fwd_blocks_sort=sort, roi_fwd_blocks_parts=parts):
        _lib.timeline.reset(True)
        top, arg8 = op.roi_pool_compact(ft, rt, 7, 7, 1.0 / 16, rounding=mode)
        torch.cuda.synchronize()
        ran = "roi_pool_forward_blocks_prepare" in _lib.timeline.summary()
        _lib.timeline.reset(False)
    tag = "case %d %s N %d map %dx%dx%d R %d %s sort %d parts %d style %d" % (k, kind, N, H, W, C, R, mode, sort, parts, style)
    if op.compact_overflowed(ft.device):
        op._flags(ft.device).flags.zero_()                 # a window beyond 15 x 16 cells: the 1-byte pair refuses it
        print("skipped (window overflow) " + tag, flush=True)
        continue
    ok_t = np.array_equal(top.cpu().numpy().view(np.uint32), et.view(np.uint32))
    ok_a = np.array_equal(op.expand_argmax(arg8, rt, shape, 7, 7, 1.0 / 16, rounding=mode).cpu().numpy(), ea)
    if not (ran and ok_t and ok_a):
        bad += 1
        print("MISMATCH %s: block path ran %s, top %s, argmax %s" % (tag, ran, ok_t, ok_a), flush=True)
    if (k + 1) % 10 == 0:
        print("case %d ok so far (%d mismatches)" % (k + 1, bad), flush=True)
print("cases %d mismatches %d" % (args.cases, bad))
sys.exit(1 if bad else 0)
