#!/usr/bin/env python3
"""Per-workgroup timeline of the RoI-pool backward (tuning tool, needs a trace build:
WSSDL_BUS_HIP_LIB=.../libwssdl_trace.so built with WSSDL_HIPCC_EXTRA=-DWSSDL_BWDC_TRACE=1).
Prints: kernel span, sum of workgroup times / (span x slots) = slot utilisation, the longest
workgroups and how record / bin counts relate to workgroup time."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wssdl_bus_amd import _lib  # noqa: E402
from wssdl_bus_amd.roi_pooling_layer.roi_pooling_op import roi_pool_compact, roi_pool_grad_compact  # noqa: E402
from wssdl_bus_amd.rpn_msr.proposal_layer_tf_bus import proposal_layer_padded, compact_rois  # noqa: E402
from kernel_bench import synth_rpn  # noqa: E402


def main():
    N, H, W, C = 8, 38, 63, 1024
    info = torch.tensor([[600, 1000, 1.0, 1.0]] * N, device="cuda")
    prob, pred = synth_rpn(N, H, W, 9, 3)
    rois_p, counts = proposal_layer_padded(prob, pred, info, True)
    rois = compact_rois(rois_p, counts)
    b = rois[:, 0]
    keep = torch.zeros_like(b, dtype=torch.bool)
    for i in range(N):
        idx = torch.nonzero(b == i).flatten()
        keep[idx if i >= N // 2 else idx[:128]] = True
    rois = rois[keep].contiguous()
    feat = torch.relu(torch.randn((N, H, W, C), device="cuda"))
    top, arg8 = roi_pool_compact(feat, rois, 7, 7, 1.0 / 16)
    diff = torch.randn_like(top)
    L = _lib.lib()
    nblk = 1 << 18
    trace = torch.zeros((nblk, 4), dtype=torch.int64, device="cuda")
    L.wssdl_debug_set_trace.argtypes = [ctypes.c_void_p]
    for _ in range(3):
        roi_pool_grad_compact(tuple(feat.shape), rois, arg8, diff, 7, 7, 1.0 / 16)
    torch.cuda.synchronize()
    L.wssdl_debug_set_trace(ctypes.c_void_p(trace.data_ptr()))
    roi_pool_grad_compact(tuple(feat.shape), rois, arg8, diff, 7, 7, 1.0 / 16)
    torch.cuda.synchronize()
    L.wssdl_debug_set_trace(None)
    t = trace.cpu().numpy()
    t = t[t[:, 1] > 0]
    start, end, rec, bins = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    t0 = start.min()
    dur = (end - start) * 10e-3            # us (100 MHz)
    span = (end.max() - t0) * 10e-3
    print("workgroups %d  kernel span %.1f us  sum(wg time) %.0f us  mean %.1f  median %.1f  max %.1f us"
          % (len(t), span, dur.sum(), dur.mean(), np.median(dur), dur.max()))
    for slots in (1024, 1792, 2048):
        print("  utilisation at %d slots: %.2f" % (slots, dur.sum() / (span * slots)))
    print("records per wg: mean %.0f max %d;  bins per wg: mean %.0f max %d" % (rec.mean(), rec.max(), bins.mean(), bins.max()))
    o = np.argsort(-dur)[:8]
    for i in o:
        print("  wg: start %.1f us dur %.1f us records %d bins %d -> %.2f us/record" %
              ((start[i] - t0) * 10e-3, dur[i], rec[i], bins[i], dur[i] / max(rec[i], 1)))
    A = np.stack([rec, bins, np.ones_like(rec)], 1).astype(np.float64)
    coef = np.linalg.lstsq(A, dur, rcond=None)[0]
    print("fit: wg time = %.3f us/record + %.3f us/bin + %.1f us" % tuple(coef))
    last = np.sort((end - t0) * 10e-3)
    print("end times: 50%% %.1f  90%% %.1f  99%% %.1f  100%% %.1f us" %
          tuple(last[[len(last) // 2, int(len(last) * .9), int(len(last) * .99), -1]]))
    # concurrency over time
    ev = np.concatenate([np.stack([start - t0, np.ones_like(start)], 1), np.stack([end - t0, -np.ones_like(end)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    conc = np.cumsum(ev[:, 1])
    for frac in (0.1, 0.25, 0.5, 0.75, 0.9):
        k = np.searchsorted(ev[:, 0], frac * (end.max() - t0))
        print("  workgroups resident at %.0f%% of the span: %d" % (frac * 100, conc[min(k, len(conc) - 1)]))


if __name__ == "__main__":
    main()
