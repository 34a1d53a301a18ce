#!/usr/bin/env python3
"""The roofline leg of bench.py: the RoI-pool pair of the training path on ONE fixed RoI set.

The RoI count the train step feeds RoI pooling with depends on what the randomly initialised RPN
leaves after NMS, so it moves from run to run.  This leg pins it: exactly
n_sup * 128 + n_ws * 2000 rows (8512 for the default workload, BASELINE configs[2]) taken from
the network's own proposals -- the sampled rows of the supervised images, the kept proposals of
the weak images topped up to 2000 from the pre-NMS list in score order -- saved once to
profiles/roofline_rois_r8512.npy (tools/make_roofline_rois.py) and loaded here.  bench.py runs
this leg after its timed steps; tools/profile_round.sh runs it alone under rocprofv3 (kernel
trace, then one PMC pass per counter group), so that kernel times, HBM traffic and the BENCH line
all refer to the same launches.

    python3 tools/roofline_leg.py [--iters 20] [--warmup 3]      -> one JSON line
"""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ROIS_PATH = os.path.join(ROOT, "profiles", "roofline_rois_r8512.npy")
KERNEL_SOURCES = ("roi_pool.hip", "roi_pool.hip.h", "roi_pool_compact.hip", "roi_pool_blocks.hip", "roi_pool_walk.hip")


def kernel_source_id():
    """Identifies the kernels a traffic measurement belongs to (the GPU box has no .git): a hash of the
    RoI-pool sources without their full-line comments and blank lines."""
    h = hashlib.sha1()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "wssdl_bus_amd", "csrc", f), "rb") as fh:
            for line in fh:
                t = line.strip()
                if t and not t.startswith(b"//"):
                    h.update(t + b"\n")
    return h.hexdigest()[:16]


def alg_bytes(op, N, H, W, C, R, PH=7, PW=7):
    """Algorithmic bytes per launch, SURVEY.md section 8(d) (the reference's layout: f32 top + i32 argmax)."""
    if op == "roi_pool_forward":
        return N * H * W * C * 4 + R * 20 + R * PH * PW * C * 8
    if op == "roi_pool_backward":
        return R * PH * PW * C * 8 + N * H * W * C * 4
    return 0


def moved_bytes(op, N, H, W, C, R, PH=7, PW=7):
    """Bytes the training path has to move at least (1-byte arg-max, no re-reads)."""
    if op == "roi_pool_forward":
        return N * H * W * C * 4 + R * 20 + R * PH * PW * C * 5
    if op == "roi_pool_backward":
        return R * PH * PW * C * 5 + N * H * W * C * 4
    return 0


def load_rois(path=ROIS_PATH):
    import numpy as np
    rois = np.load(path)
    assert rois.ndim == 2 and rois.shape[1] == 5 and rois.dtype == np.float32
    return rois, "profiles/%s sha1 %s" % (os.path.basename(path), hashlib.sha1(rois.tobytes()).hexdigest()[:12])


def run(rois_np, N, H, W, C, iters=20, warmup=3, seed=3):
    """Times roi_pool_forward (compact), roi_pool_backward_prepare and roi_pool_backward (walk) on the
    given RoI set with HIP events on the launch stream (_lib.timed); returns {op: {avg_ms, calls, ...}}."""
    import torch
    from wssdl_bus_amd import _lib
    from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(seed)
    feat = torch.relu(torch.randn((N, H, W, C), device=dev, generator=g))
    rois = torch.from_numpy(rois_np).to(dev)
    R = rois.shape[0]
    shape = (N, H, W, C)
    compact = op.compact_supported(H, W, C, 7, 7)
    saved_enabled, saved_records = _lib.timeline.enabled, _lib.timeline.records
    try:
        if compact:
            top, arg = op.roi_pool_compact(feat, rois, 7, 7, 1.0 / 16)
        else:
            top, arg = op.roi_pool(feat, rois, 7, 7, 1.0 / 16)
        diff = torch.randn(top.shape, device=dev, generator=g)
        plan = op.prepare_backward(shape, rois, 7, 7, 1.0 / 16) if compact else None
        segs = plan.segments if compact else 1

        def one():
            if compact:
                op.roi_pool_compact(feat, rois, 7, 7, 1.0 / 16)
                p = op.prepare_backward(shape, rois, 7, 7, 1.0 / 16)
                return op.roi_pool_grad_compact(shape, rois, arg, diff, 7, 7, 1.0 / 16, plan=p, segments=p.segments)
            op.roi_pool(feat, rois, 7, 7, 1.0 / 16)
            return op.roi_pool_grad(feat, rois, arg, diff, 7, 7, 1.0 / 16)
        for _ in range(warmup):
            one()
        torch.cuda.synchronize()
        _lib.timeline.reset(True)
        for _ in range(iters):
            one()
        torch.cuda.synchronize()
        tl = _lib.timeline.summary()
    finally:
        _lib.timeline.enabled, _lib.timeline.records = saved_enabled, saved_records
    out = {}
    # the block-table forward is one op of three launches (tables + bin-row order, then the pooling kernel): quoted
    # together, like the bin-owner backward's walk + merge; the parts stay visible in `parts_ms`
    prep = tl.pop("roi_pool_forward_blocks_prepare", None)
    if prep is not None and "roi_pool_forward" in tl:
        f = tl["roi_pool_forward"]
        f["parts_ms"] = dict(tables_and_order=prep["avg_ms"], pooling=f["avg_ms"])
        f["avg_ms"] += prep["avg_ms"]
    for name, d in tl.items():
        ab = alg_bytes(name, N, H, W, C, R)
        out[name] = dict(avg_ms=d["avg_ms"], calls=d["calls"], alg_bytes_per_launch=ab,
                         GBps=(ab / (d["avg_ms"] * 1e-3) / 1e9) if ab else None)
        if ab and compact:
            out[name]["min_moved_bytes"] = moved_bytes(name, N, H, W, C, R)
        if "parts_ms" in d:
            out[name]["parts_ms"] = d["parts_ms"]
    meta = dict(N=N, H=H, W=W, C=C, R=R, argmax_bytes=1 if compact else 4,
                backward_plan=(plan.plan if plan is not None else None),
                backward_owner_plan=(plan.owner if plan is not None else None),
                backward_variant=(plan.variant if plan is not None else "i32 pair"),
                backward_segments=segs, kernel_source_id=kernel_source_id(),
                forward_variant=("block tables (k = 2, 3, 4) + bin rows in (image, first window row) order: 3 launches"
                                 if prep is not None else
                                 "one wave per bin (small launch)" if (compact and R * 7 * ((C + 255) // 256) < 32768 and C % 256 == 0
                                                                       and _lib.get_tuning("roi_fwd_one_bin") > 0) else
                                 "rows kernel" if R * 7 * ((C + 255) // 256) >= 32768 else "sliced kernel (small launch)"))
    return out, meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--roi-bwd-plan", type=int, default=-1, help="force a backward plan (wssdl_set_tuning)")
    ap.add_argument("--check", action="store_true",
                    help="before timing: the pair's outputs on this RoI set against the C oracle (all 8512 RoIs: top, arg-max "
                         "and the exact walk bit for bit, the bin-owner form at its tolerance; tests/test_gpu_roi_compact.py)")
    ap.add_argument("--rois", default="", metavar="PATH.npy", help="another RoI set (float32 [R,5]) instead of the fixed one")
    ap.add_argument("--map", default="38,63,1024", help="H,W,C of the feature map the set belongs to")
    ap.add_argument("--i32", action="store_true",
                    help="the op pair of the reference's contract (f32 top + i32 arg-max, wssdl_roi_pool_forward / _backward_ws) "
                         "instead of the training path's 1-byte pair")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT",
                    help="wssdl_set_tuning(KEY, INT) before the run, e.g. roi_fwd_blocks=1 (repeatable)")
    ap.add_argument("--window-table-min-rois", type=int, default=-1,
                    help="experiment: the smallest RoI list that gets a window table (and with it the block-table forward)")
    args = ap.parse_args()
    import torch
    assert torch.cuda.is_available()
    if args.window_table_min_rois >= 0:
        from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op
        roi_pooling_op._WINDOW_TABLE_MIN_ROIS = args.window_table_min_rois
    if args.check:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from test_gpu_roi_compact import roofline_set_parity
        checked, plans = roofline_set_parity(torch, owners=(8,))
        print(json.dumps(dict(check="top, argmax, bottom_diff == C oracle", rois_per_image=checked, plans=plans)))
    rois, tag = load_rois(args.rois) if args.rois else load_rois()
    if args.roi_bwd_plan >= 0:
        from wssdl_bus_amd import _lib
        _lib.set_tuning("roi_bwd_plan", args.roi_bwd_plan)
    if args.i32:
        from wssdl_bus_amd.fast_rcnn.config import cfg
        cfg.ROI_POOL_COMPACT_ARGMAX = False
    for kv in args.tune:
        from wssdl_bus_amd import _lib
        key, val = kv.split("=")
        _lib.set_tuning(key, int(val))
    N = int(rois[:, 0].max()) + 1
    H, W, C = (int(v) for v in args.map.split(","))
    ops, meta = run(rois, N, H, W, C, args.iters, args.warmup)
    print(json.dumps(dict(roi_set=tag, meta=meta, ops=ops)))


if __name__ == "__main__":
    main()
