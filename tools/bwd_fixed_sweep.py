#!/usr/bin/env python3
"""Times plans of the list-driven RoI-pool backward (tile shape x records in flight x channels per
lane) on the FIXED roofline RoI set (profiles/roofline_rois_r8512.npy) and checks every one bit for
bit against plan 11.

    python3 tools/bwd_fixed_sweep.py --plans 11,13,16 [--iters 20] [--one PLAN]
    python3 tools/bwd_fixed_sweep.py --plans 13 --owner 8,0,1 [--images 4,5] [--rois set.npy --map 37,62 --channels 512]
    python3 tools/bwd_fixed_sweep.py --i32 --plans 9,11,13        (the declared op's i32 arg-max: exact walk and owner_i32)

--owner times plans of the bin-owner form (wssdl_roi_pool_backward_compact_owner: walk + halo merge) after the exact plans and
checks them against plan 11 at the owner form's tolerance + repeatability.  --one / --one-owner run a single plan a few
times and nothing else: the form to put under `rocprofv3 --pmc ...` for the HBM traffic / SQ counters of one variant
(tools/pmc_owner.sh, tools/pmc_sq_owner.sh).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import torch  # noqa: E402

from roofline_leg import load_rois, moved_bytes  # noqa: E402
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.roi_pooling_layer import roi_pooling_op as op  # noqa: E402


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plans", default="11,13,16,5,9,10")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--one", default="")
    ap.add_argument("--channels", type=int, default=1024)
    ap.add_argument("--images", default="", help="subset of the set's images, e.g. 0,4,5 (1 supervised + 2 weak: "
                    "the low-image-count regime of the VGG-16 / alternating workloads); re-indexed from 0")
    ap.add_argument("--rois", default="", help="another RoI set (float32 [R,5] .npy) instead of the default workload's")
    ap.add_argument("--map", default="38,63", help="feature-map height,width")
    ap.add_argument("--segments", default="1", help="segments of the split walk to time per plan, e.g. 1,2,4,8 (1 = the "
                    "exact walk; > 1 is compared with it by the largest difference relative to max |bottom_diff|)")
    ap.add_argument("--i32", action="store_true", help="time wssdl_roi_pool_backward_ws (i32 arg-max, prepare + walk in "
                    "one call) per plan instead of the 1-byte pair; checked against the tile-owner kernel's result")
    ap.add_argument("--owner", default="", help="owner plans (bin-owner form, wssdl_roi_pool_backward_compact_owner) to time "
                    "after the exact plans, e.g. 0,1,4: checked against plan 11 by the largest difference relative to "
                    "max |bottom_diff| and for repeatability")
    ap.add_argument("--owner-segments", default="1", help="waves per tile stream of the owner plans to time, e.g. 1,2,4 "
                    "(round 6: wssdl_roi_pool_backward_compact_owner_split)")
    ap.add_argument("--one-owner", default="", help="like --one for an owner plan (the form to put under rocprofv3 --pmc)")
    ap.add_argument("--dup", type=int, default=1, help="repeat the set's images DUP times as further images (N x DUP images, "
                    "R x DUP RoIs): time(DUP = 2) - time(DUP = 1) is the bulk rate without the launch's ramp and tail")
    ap.add_argument("--denormals", action="store_true",
                    help="scale top_diff so that sums pass through the f32 denormal range (checks that every plan, "
                         "the ds_add_f32 ones included, still equals plan 11 bit for bit)")
    args = ap.parse_args()
    rois_np, tag = load_rois(args.rois) if args.rois else load_rois()
    if args.images:
        import numpy as np
        keep = [int(x) for x in args.images.split(",")]
        rois_np = np.concatenate([np.concatenate([np.full((int((rois_np[:, 0] == k).sum()), 1), i, np.float32),
                                                  rois_np[rois_np[:, 0] == k][:, 1:]], axis=1)
                                  for i, k in enumerate(keep)]).astype(np.float32)
    if args.dup > 1:
        import numpy as np
        n0 = int(rois_np[:, 0].max()) + 1
        rois_np = np.concatenate([rois_np + np.array([[k * n0, 0, 0, 0, 0]], np.float32) for k in range(args.dup)]).astype(np.float32)
    H, W = (int(v) for v in args.map.split(","))
    N, C = int(rois_np[:, 0].max()) + 1, args.channels
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(3)
    feat = torch.relu(torch.randn((N, H, W, C), device=dev, generator=g))
    rois = torch.from_numpy(rois_np).to(dev)
    R = rois.shape[0]
    shape = (N, H, W, C)
    top, arg8 = op.roi_pool_compact(feat, rois, 7, 7, 1.0 / 16)
    diff = torch.randn(top.shape, device=dev, generator=g)
    if args.denormals:
        diff = diff * 1e-38          # |values| ~ 1e-38: below and around FLT_MIN = 1.18e-38
    del top
    mb = moved_bytes("roi_pool_backward", N, H, W, C, R)

    def run(plan_id):
        with _lib.tuned(roi_bwd_plan=plan_id):
            plan = op.roi_pool_grad_prepare(shape, rois, 7, 7, 1.0 / 16)
        assert plan.plan == plan_id
        return plan

    if args.i32:
        top, arg = op.roi_pool(feat, rois, 7, 7, 1.0 / 16)
        del top
        ref = torch.empty(shape, dtype=torch.float32, device=dev)
        L = _lib.lib()
        _lib.check(L.wssdl_roi_pool_backward(_lib.ptr(diff), _lib.ptr(arg), _lib.ptr(rois), R, N, H, W, C, 7, 7, 1.0 / 16,
                                             _lib.ptr(ref), _lib.stream()), "wssdl_roi_pool_backward")
        ab = R * 49 * C * 8 + N * H * W * C * 4
        for p in ["auto"] + [int(x) for x in args.plans.split(",")]:
            _lib.set_tuning("roi_bwd_plan", -1 if p == "auto" else p)
            got = op.roi_pool_grad(feat, rois, arg, diff, 7, 7, 1.0 / 16)
            same = bool(torch.equal(got, ref))
            del got
            ms = timeit(lambda: op.roi_pool_grad(feat, rois, arg, diff, 7, 7, 1.0 / 16), args.iters)
            print(json.dumps(dict(i32=True, plan=p, prepare_plus_walk_ms=round(ms, 4), equal_to_tile_owner_kernel=same,
                                  alg8d_TBps=round(ab / ms / 1e9, 3), frac_8d=round(ab / ms / 1e9 / 8.0, 3))), flush=True)
            assert same, p
        _lib.set_tuning("roi_bwd_plan", -1)
        scale = float(ref.abs().max())
        for o in (0, 1):
            nws = L.wssdl_roi_pool_backward_workspace_bytes(R, N, H, W, 7, 7)
            nscr = L.wssdl_roi_pool_backward_owner_scratch_bytes(N, H, W, C, o)
            ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
            scr = torch.empty((nscr,), dtype=torch.uint8, device=dev)
            out = torch.empty(shape, dtype=torch.float32, device=dev)

            def call():
                _lib.check(L.wssdl_roi_pool_backward_owner_i32(_lib.ptr(diff), _lib.ptr(arg), _lib.ptr(rois), R, N, H, W, C, 7, 7,
                                                               1.0 / 16, _lib.ptr(out), _lib.ptr(ws), nws, o, _lib.ptr(scr), nscr,
                                                               _lib.stream()), "wssdl_roi_pool_backward_owner_i32")
            call()
            rel = float((out - ref).abs().max()) / scale
            ms = timeit(call, args.iters)
            print(json.dumps(dict(i32=True, owner=o, prepare_plus_walk_plus_merge_ms=round(ms, 4), max_diff_over_max_abs=rel,
                                  alg8d_TBps=round(ab / ms / 1e9, 3), frac_8d=round(ab / ms / 1e9 / 8.0, 3))), flush=True)
            assert rel <= 1e-5, o
        return
    if args.one_owner:
        plan = op.roi_pool_grad_prepare_owner(shape, rois, 7, 7, 1.0 / 16, int(args.one_owner))
        for _ in range(5):
            op.roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=plan)
        torch.cuda.synchronize()
        print(json.dumps(dict(one_owner=args.one_owner, R=R, C=C)))
        return
    if args.one:
        plan = run(int(args.one))
        for _ in range(5):
            op.roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=plan)
        torch.cuda.synchronize()
        print(json.dumps(dict(one=args.one, R=R, C=C)))
        return

    ref = op.roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=run(11))
    # clocks up: with 10 launches the first plan of the list measured ~8 % slow (plan 11 0.58-0.60 ms in first place, 0.516-0.526
    # later in the same run), which is how plan 13 once seemed faster than plan 11
    timeit(lambda: op.roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=run(11)), 100)
    scale = float(ref.abs().max())
    for p in (int(x) for x in args.plans.split(",")):
        plan = run(p)
        with _lib.tuned(roi_bwd_plan=p):
            ms_prep = timeit(lambda: op.roi_pool_grad_prepare(shape, rois, 7, 7, 1.0 / 16), 10)
        for seg in (int(x) for x in args.segments.split(",")):
            got = op.roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=plan, segments=seg)
            same = bool(torch.equal(got, ref))
            rel = float((got - ref).abs().max()) / scale
            again = op.roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=plan, segments=seg)
            repeatable = bool(torch.equal(got, again))
            del got, again
            ms = timeit(lambda: op.roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=plan, segments=seg),
                        args.iters)
            print(json.dumps(dict(plan=p, segments=seg, walk_ms=round(ms, 4), prepare_ms=round(ms_prep, 4),
                                  equal_to_plan11=same, max_diff_over_max_abs=rel, repeatable=repeatable,
                                  moved_TBps=round(mb / ms / 1e9, 3), frac_moved=round(mb / ms / 1e9 / 8.0, 3))), flush=True)
            assert repeatable and (same if seg == 1 else rel <= 1e-5), (p, seg)
    for o in (int(x) for x in args.owner.split(",") if x != ""):
      for oseg in (int(x) for x in args.owner_segments.split(",") if x != ""):
        plan = op.roi_pool_grad_prepare_owner(shape, rois, 7, 7, 1.0 / 16, o)
        plan.owner_segments = oseg
        ms_prep = timeit(lambda: op.roi_pool_grad_prepare_owner(shape, rois, 7, 7, 1.0 / 16, o), 10)
        got = op.roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=plan)
        rel = float((got - ref).abs().max()) / scale
        again = op.roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=plan)
        repeatable = bool(torch.equal(got, again))
        del got, again
        ms = timeit(lambda: op.roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=plan), args.iters)
        print(json.dumps(dict(owner=o, owner_segments=oseg, walk_plus_merge_ms=round(ms, 4), prepare_ms=round(ms_prep, 4),
                              max_diff_over_max_abs=rel, repeatable=repeatable,
                              moved_TBps=round(mb / ms / 1e9, 3), frac_moved=round(mb / ms / 1e9 / 8.0, 3))), flush=True)
        assert repeatable and rel <= 1e-5, (o, oseg)
    assert not op.flags_raised()


if __name__ == "__main__":
    main()
