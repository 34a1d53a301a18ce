#!/usr/bin/env python3
"""Random NMS cases, product (wssdl_nms / wssdl_nms_new through hip_nms) against the C oracle's keep lists (and the reference's own Cython build, oracle/_ref, when present): sizes 1 ... 26000 (beyond 20480 boxes: the general sweep instead of the pipelined one)
(not multiples of 64 on purpose), tight clusters, duplicates, degenerate boxes, thresholds 0.05 ... 0.95, max_keep cuts.
Scores are distinct (the order of equal scores is the product's own rule, tested elsewhere).
    python3 tools/nms_fuzz.py [--cases 400] [--seed 0]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import c_oracle, ref_kernels  # noqa: E402
from wssdl_bus_amd.nms.hip_nms import hip_nms  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=400)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT", help="wssdl_set_tuning before the run, e.g. nms_fused=0")
ap.add_argument("--high-thresholds", action="store_true", help="thresholds 0.6 ... 0.95 mostly (the RPN range)")
args = ap.parse_args()
for kv in args.tune:
    from wssdl_bus_amd import _lib
    _lib.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
rs = np.random.RandomState(args.seed)
bad = n_ref = 0
for k in range(args.cases):
    n = int(rs.choice([1, 2, 63, 64, 65, 127, 129, 300, 1000, 2047, 2049, 4097, 6000, 9001, 12000, 13000, 20480, 20481, 26000])) if k % 3 else int(rs.randint(1, 13001))
    kind = k % 5
    spread = [1000.0, 300.0, 100.0, 30.0, 600.0][kind]
    c = rs.uniform(0, spread, size=(n, 2)) * [1.0, 0.6]
    wh = np.exp(rs.normal(4.0, 0.7, size=(n, 2)))
    if kind == 3:                                        # near-duplicates of a few boxes
        base = rs.randint(0, max(n // 50, 1), size=n)
        c, wh = c[base] + rs.normal(0, 1.0, (n, 2)), wh[base]
    d = np.hstack((c - wh / 2, c + wh / 2, np.zeros((n, 1)))).astype(np.float32)
    if kind == 4 and n > 8:                              # degenerate: zero / negative extent, exact duplicates
        d[:4, 2:4] = d[:4, 0:2]
        d[4:6, 2:4] = d[4:6, 0:2] - 3
        d[6] = d[7]
    d[:, 4] = (rs.permutation(n) + 1).astype(np.float32) / np.float32(n + 1)
    thresh = float(rs.choice([0.6, 0.65, 0.7, 0.8, 0.95, 0.5] if args.high_thresholds else [0.05, 0.3, 0.5, 0.7, 0.95]))
    rule = "nms_new" if k % 4 == 0 else "nms"
    want = (c_oracle.nms_new if rule == "nms_new" else c_oracle.cpu_nms)(d, thresh)
    # ... and, where oracle/_ref is built, the reference's own compiled Cython says the same (cpu_nms.pyx / utils/nms.pyx)
    ref_fn = (ref_kernels.nms_new if rule == "nms_new" else ref_kernels.cpu_nms)
    if ref_fn is not None:
        ref_keep = [int(v) for v in ref_fn(d, thresh)]
        if ref_keep != want:
            bad += 1
            print("ORACLE != REFERENCE BUILD case %d: n %d thresh %.2f rule %s" % (k, n, thresh, rule))
        n_ref += 1
    mk = None if k % 2 else int(rs.choice([1, 64, 300, 2000]))
    got = hip_nms(d, thresh, max_keep=mk, rule=rule)
    if mk is not None:
        want = want[:mk]
    if got != want:
        bad += 1
        print("MISMATCH case %d: n %d kind %d thresh %.2f rule %s max_keep %s: got %d kept, want %d" % (k, n, kind, thresh, rule, mk, len(got), len(want)))
    if (k + 1) % 100 == 0:
        print("case %d ok so far (%d mismatches)" % (k + 1, bad), flush=True)
print("cases %d mismatches %d (%d of them also against the reference's own Cython build)" % (args.cases, bad, n_ref))
sys.exit(1 if bad else 0)
