#!/usr/bin/env python3
"""Per-op timing of the hot-path kernels at the BASELINE config sizes (HIP events on
the launch stream).  Prints one JSON line per op with achieved algorithmic GB/s.

    python tools/kernel_bench.py [--config 3] [--iters 20]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.fast_rcnn.config import cfg  # noqa: E402
from wssdl_bus_amd.roi_pooling_layer.roi_pooling_op import (roi_pool, roi_pool_grad, roi_pool_compact,  # noqa: E402
                                                            roi_pool_grad_compact, compact_supported)
from wssdl_bus_amd.rpn_msr.anchor_target_layer_tf_bus import anchor_target_layer  # noqa: E402
from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded, compact_rois  # noqa: E402
from wssdl_bus_amd.nms.hip_nms import hip_nms  # noqa: E402


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def synth_rpn(N, H, W, A, seed):
    g = torch.Generator("cuda").manual_seed(seed)
    logits = torch.randn((N, H, W, A, 2), device="cuda", generator=g)
    p = torch.softmax(logits, dim=-1)
    prob = torch.cat((p[..., 0], p[..., 1]), dim=-1).contiguous()
    pred = (0.2 * torch.randn((N, H, W, 4 * A), device="cuda", generator=g)).contiguous()
    return prob, pred


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--joint", action="store_true",
                    help="RoI set of the combined mini-batch: first half of the images keep 128 RoIs "
                         "(supervised, sampled), the rest keep all their proposals (weak)")
    args = ap.parse_args()
    A = 9
    if args.config == 5:
        N, H, W, C, im_h, im_w, train = 1, 63, 100, 1024, 1000, 1600, False
    elif args.config == 2:
        N, H, W, C, im_h, im_w, train = 2, 38, 63, 256, 600, 1000, True
    else:
        N, H, W, C, im_h, im_w, train = 8, 38, 63, 1024, 600, 1000, True
    info = torch.tensor([[im_h, im_w, 1.0, 1.0]] * N, device="cuda")
    prob, pred = synth_rpn(N, H, W, A, 3)
    out = []

    # proposal layer (all images)
    rois_p, counts = proposal_layer_padded(prob, pred, info, train)
    ms = timeit(lambda: proposal_layer_padded(prob, pred, info, train), args.iters)
    cnt = counts.cpu().numpy()
    out.append(dict(op="proposal_layer", ms=ms, images=N, rois=int(cnt.sum()),
                    images_per_s=N / ms * 1e3))
    rois = compact_rois(rois_p, counts)
    if args.joint:
        b = rois[:, 0]
        keep = torch.zeros_like(b, dtype=torch.bool)
        for i in range(N):
            idx = torch.nonzero(b == i).flatten()
            keep[idx if i >= N // 2 else idx[:128]] = True
        rois = rois[keep].contiguous()
    R = rois.shape[0]

    # RoI pool forward / backward
    feat = torch.relu(torch.randn((N, H, W, C), device="cuda"))
    top, arg = roi_pool(feat, rois, 7, 7, 1.0 / 16)
    ms = timeit(lambda: roi_pool(feat, rois, 7, 7, 1.0 / 16), args.iters)
    byt = N * H * W * C * 4 + R * 20 + R * 49 * C * 8
    out.append(dict(op="roi_pool_forward", ms=ms, R=R, C=C, alg_bytes=byt, GBps=byt / ms / 1e6))
    diff = torch.randn_like(top)
    ms = timeit(lambda: roi_pool_grad(feat, rois, arg, diff, 7, 7, 1.0 / 16), args.iters)
    byt = R * 49 * C * 8 + N * H * W * C * 4
    out.append(dict(op="roi_pool_backward", ms=ms, R=R, C=C, alg_bytes=byt, GBps=byt / ms / 1e6))

    # training path: 1-byte arg-max (same algorithmic bytes: the metric counts the reference's layout)
    if compact_supported(H, W, C, 7, 7):
        top_c, arg8 = roi_pool_compact(feat, rois, 7, 7, 1.0 / 16)
        assert os.environ.get("KB_NO_CHECK") or torch.equal(top_c, top)
        for fv in os.environ.get("KB_FWD_VARIANTS", "0").split(","):
            _lib.set_tuning("roi_fwd_variant", int(fv))
            t2, a2 = roi_pool_compact(feat, rois, 7, 7, 1.0 / 16)
            assert os.environ.get("KB_NO_CHECK") or (torch.equal(t2, top) and torch.equal(a2, arg8))
            ms = timeit(lambda: roi_pool_compact(feat, rois, 7, 7, 1.0 / 16), args.iters, warmup=5)
            byt = N * H * W * C * 4 + R * 20 + R * 49 * C * 8
            out.append(dict(op="roi_pool_forward_compact[v%s]" % fv, ms=ms, R=R, C=C, alg_bytes=byt, GBps=byt / ms / 1e6,
                            moved_bytes=N * H * W * C * 4 + R * 20 + R * 49 * C * 5))
        _lib.set_tuning("roi_fwd_variant", 0)
        ref_g = roi_pool_grad(feat, rois, arg, diff, 7, 7, 1.0 / 16)
        from wssdl_bus_amd.roi_pooling_layer.roi_pooling_op import roi_pool_grad_prepare
        shape = tuple(feat.shape)
        for plan_id in os.environ.get("KB_BWD_PLANS", "auto").split(","):
            # "auto" = the plan the library picks for this launch size
            _lib.set_tuning("roi_bwd_plan", -1 if plan_id == "auto" else int(plan_id))
            plan = roi_pool_grad_prepare(shape, rois, 7, 7, 1.0 / 16)
            g = roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=plan)
            assert os.environ.get("KB_NO_CHECK") or torch.equal(g, ref_g), plan_id
            ms_p = timeit(lambda: roi_pool_grad_prepare(shape, rois, 7, 7, 1.0 / 16), args.iters, warmup=5)
            ms = timeit(lambda: roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, plan=plan), args.iters,
                        warmup=5)
            byt = R * 49 * C * 8 + N * H * W * C * 4
            out.append(dict(op="roi_pool_backward_compact[plan %s]" % plan_id, ms=ms, prepare_ms=ms_p, R=R, C=C,
                            alg_bytes=byt, GBps=byt / ms / 1e6, GBps_with_prepare=byt / (ms + ms_p) / 1e6))
        _lib.set_tuning("roi_bwd_plan", -1)
        ms = timeit(lambda: roi_pool_grad_compact(shape, rois, arg8, diff, 7, 7, 1.0 / 16, use_workspace=False),
                    args.iters, warmup=5)
        out.append(dict(op="roi_pool_backward_compact[no lists]", ms=ms, R=R, C=C, alg_bytes=byt, GBps=byt / ms / 1e6))

    # anchor targets
    gt = torch.zeros((N, 20, 5), device="cuda")
    gt[:, 0] = torch.tensor([100.0, 80.0, 380.0, 300.0, 1.0], device="cuda")
    gt[:, 1] = torch.tensor([500.0, 60.0, 900.0, 420.0, 0.0], device="cuda")
    ng = torch.full((N,), 2, dtype=torch.int32, device="cuda")
    score = torch.empty((N, H, W, 2 * A), device="cuda")
    for mode in ("device", "reference"):
        cfg.SAMPLING_RNG = mode
        ms = timeit(lambda: anchor_target_layer(score, gt, ng, info, None, [16], [8, 16, 32], "SNUBH"),
                    args.iters)
        byt = N * (13 * A * H * W * 4)
        out.append(dict(op="anchor_target_layer[%s rng]" % mode, ms=ms, images=N, alg_bytes=byt,
                        GBps=byt / ms / 1e6))
    cfg.SAMPLING_RNG = "reference"

    # standalone NMS, 12000 boxes
    rs = np.random.RandomState(3)
    n = 12000
    c = rs.uniform(0, 1000, size=(n, 2)) * [1.0, 0.6]
    wh = np.exp(rs.normal(4.5, 0.6, size=(n, 2)))
    d = np.hstack((c - wh / 2, c + wh / 2, rs.permutation(n)[:, None] / float(n))).astype(np.float32)
    dd = torch.from_numpy(d).cuda()
    ms = timeit(lambda: hip_nms(dd, 0.7, max_keep=2000), args.iters)
    byt = n * 20 + 2 * n * ((n + 63) // 64) * 8 // 2
    out.append(dict(op="nms_12000_keep2000", ms=ms, alg_bytes=byt, GBps=byt / ms / 1e6))
    # what this box sustains: device-to-device copy (read + write) and a pure write stream
    nb = 1 << 30
    a = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    b = torch.empty((nb,), dtype=torch.uint8, device="cuda")
    ms = timeit(lambda: b.copy_(a), 10)
    out.append(dict(op="d2d_copy_1GiB", ms=ms, alg_bytes=2 * nb, GBps=2 * nb / ms / 1e6))
    ms = timeit(lambda: b.zero_(), 10)
    out.append(dict(op="fill_1GiB", ms=ms, alg_bytes=nb, GBps=nb / ms / 1e6))
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
