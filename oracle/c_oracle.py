"""ctypes binding of ``oracle/liboracle.so`` (oracle/c/oracle_kernels.c).

TEST INFRASTRUCTURE ONLY -- see the header of oracle_kernels.c.  Builds the
library with gcc on first use when it is missing or stale.
"""
import ctypes
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "c", "oracle_kernels.c")
_LIB = os.path.join(_HERE, "liboracle.so")
_lock = threading.Lock()
_lib = None


def build(force=False):
    if (not force and os.path.exists(_LIB)
            and os.path.getmtime(_LIB) >= os.path.getmtime(_SRC)):
        return _LIB
    subprocess.check_call(
        ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c11",
         "-shared", _SRC, "-o", _LIB, "-lm"])
    return _LIB


def lib():
    global _lib
    with _lock:
        if _lib is None:
            build()
            L = ctypes.CDLL(_LIB)
            dp = ctypes.POINTER(ctypes.c_double)
            fp = ctypes.POINTER(ctypes.c_float)
            ip = ctypes.POINTER(ctypes.c_int32)
            lp = ctypes.POINTER(ctypes.c_int64)
            i64, i32 = ctypes.c_int64, ctypes.c_int
            L.orc_bbox_overlaps.argtypes = [dp, i64, dp, i64, dp]
            L.orc_bbox_overlaps.restype = None
            L.orc_bbox_overlaps_ui.argtypes = [dp, i64, dp, i64, dp]
            L.orc_bbox_overlaps_ui.restype = None
            L.orc_cpu_nms.argtypes = [fp, i64, lp, ctypes.c_double, lp]
            L.orc_cpu_nms.restype = i64
            L.orc_nms_new.argtypes = [fp, i64, lp, ctypes.c_double, lp]
            L.orc_nms_new.restype = i64
            L.orc_roi_pool_forward.argtypes = [fp, i32, i32, i32, i32, fp, i32, i32, i32,
                                               ctypes.c_float, i32, fp, ip]
            L.orc_roi_pool_forward.restype = None
            L.orc_roi_pool_forward_range.argtypes = [fp, i32, i32, i32, i32, fp, i32, i32, i32, i32,
                                                     ctypes.c_float, i32, fp, ip]
            L.orc_roi_pool_forward_range.restype = None
            for name in ("orc_roi_pool_backward", "orc_roi_pool_backward_scatter"):
                f = getattr(L, name)
                f.argtypes = [fp, ip, fp, i32, i32, i32, i32, i32, i32, i32, ctypes.c_float, fp]
                f.restype = None
            L.orc_roi_pool_backward_scatter_channels.argtypes = [fp, ip, fp, i32, i32, i32, i32, i32, i32, i32,
                                                                 ctypes.c_float, i32, i32, fp]
            L.orc_roi_pool_backward_scatter_channels.restype = None
            _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def bbox_overlaps(boxes, query):
    # the Cython kernels index columns 0..3 of 2-D arrays of any width
    boxes = np.ascontiguousarray(np.asarray(boxes)[:, :4], dtype=np.float64)
    query = np.ascontiguousarray(np.asarray(query)[:, :4], dtype=np.float64)
    out = np.zeros((boxes.shape[0], query.shape[0]), dtype=np.float64)
    lib().orc_bbox_overlaps(_p(boxes, ctypes.c_double), boxes.shape[0],
                            _p(query, ctypes.c_double), query.shape[0],
                            _p(out, ctypes.c_double))
    return out


def bbox_overlaps_ui(boxes, query):
    # the Cython kernels index columns 0..3 of 2-D arrays of any width
    boxes = np.ascontiguousarray(np.asarray(boxes)[:, :4], dtype=np.float64)
    query = np.ascontiguousarray(np.asarray(query)[:, :4], dtype=np.float64)
    out = np.zeros((boxes.shape[0], query.shape[0]), dtype=np.float64)
    lib().orc_bbox_overlaps_ui(_p(boxes, ctypes.c_double), boxes.shape[0],
                               _p(query, ctypes.c_double), query.shape[0],
                               _p(out, ctypes.c_double))
    return out


def cpu_nms(dets, thresh):
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    order = np.ascontiguousarray(dets[:, 4].argsort()[::-1], dtype=np.int64)  # cpu_nms.pyx:25
    keep = np.empty(max(n, 1), dtype=np.int64)
    nk = lib().orc_cpu_nms(_p(dets, ctypes.c_float), n, _p(order, ctypes.c_int64),
                           float(thresh), _p(keep, ctypes.c_int64))
    return [int(k) for k in keep[:nk]]


def nms_new(dets, thresh):
    """utils/nms.pyx:70-123."""
    dets = np.ascontiguousarray(dets, dtype=np.float32)
    n = dets.shape[0]
    order = np.ascontiguousarray(dets[:, 4].argsort()[::-1], dtype=np.int64)  # nms.pyx:78
    keep = np.empty(max(n, 1), dtype=np.int64)
    nk = lib().orc_nms_new(_p(dets, ctypes.c_float), n, _p(order, ctypes.c_int64),
                           float(thresh), _p(keep, ctypes.c_int64))
    return [int(k) for k in keep[:nk]]


_MODES = {"cuda": 0, "cpu": 1, 0: 0, 1: 1}


def roi_pool_forward(bottom, rois, pooled_h, pooled_w, spatial_scale, mode="cuda", threads=1):
    bottom = np.ascontiguousarray(bottom, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32).reshape(-1, 5)
    N, H, W, C = bottom.shape
    R = rois.shape[0]
    # roi_pooling_op.cc:150-152 / roi_pooling_op_gpu.cu.cc:41-43 index the batch unchecked: an index outside [0, N) reads
    # out of bounds there (and here).  Not a defined input of the reference; the product's answer for it (an empty RoI) is
    # its own, tested in tests/test_gpu_edges.py.
    if R and ((rois[:, 0] < 0).any() or (rois[:, 0].astype(np.int64) >= N).any()):
        raise ValueError("roi_pool_forward oracle: batch index outside [0, %d) -- undefined in the reference op" % N)
    top = np.empty((R, pooled_h, pooled_w, C), dtype=np.float32)
    arg = np.empty((R, pooled_h, pooled_w, C), dtype=np.int32)
    L = lib()
    if threads <= 1 or R < 2 * threads:
        L.orc_roi_pool_forward(_p(bottom, ctypes.c_float), N, H, W, C, _p(rois, ctypes.c_float), R,
                               pooled_h, pooled_w, float(spatial_scale), _MODES[mode],
                               _p(top, ctypes.c_float), _p(arg, ctypes.c_int32))
    else:
        # shard the RoI range over host threads (ctypes releases the GIL); the
        # reference shards the flat output range over TF's intra-op pool
        # (roi_pooling_op.cc:198-203).
        bounds = np.linspace(0, R, threads + 1).astype(int)
        ts = []
        for t in range(threads):
            a = (_p(bottom, ctypes.c_float), N, H, W, C, _p(rois, ctypes.c_float),
                 int(bounds[t]), int(bounds[t + 1]), pooled_h, pooled_w, float(spatial_scale),
                 _MODES[mode], _p(top, ctypes.c_float), _p(arg, ctypes.c_int32))
            th = threading.Thread(target=L.orc_roi_pool_forward_range, args=a)
            th.start()
            ts.append(th)
        for th in ts:
            th.join()
    return top, arg


def roi_pool_backward(top_diff, argmax, rois, bottom_shape, pooled_h, pooled_w, spatial_scale,
                      literal=False, threads=1):
    top_diff = np.ascontiguousarray(top_diff, dtype=np.float32)
    argmax = np.ascontiguousarray(argmax, dtype=np.int32)
    rois = np.ascontiguousarray(rois, dtype=np.float32).reshape(-1, 5)
    N, H, W, C = bottom_shape
    if threads > 1 and not literal and C >= threads:
        # channels are independent: one host thread per channel range (ctypes releases the GIL)
        out = np.zeros((N, H, W, C), dtype=np.float32)
        bounds = np.linspace(0, C, threads + 1).astype(int)
        ts = []
        for t in range(threads):
            a = (_p(top_diff, ctypes.c_float), _p(argmax, ctypes.c_int32), _p(rois, ctypes.c_float),
                 rois.shape[0], N, H, W, C, pooled_h, pooled_w, float(spatial_scale), int(bounds[t]),
                 int(bounds[t + 1]), _p(out, ctypes.c_float))
            th = threading.Thread(target=lib().orc_roi_pool_backward_scatter_channels, args=a)
            th.start()
            ts.append(th)
        for th in ts:
            th.join()
        return out
    out = np.empty((N, H, W, C), dtype=np.float32)
    f = lib().orc_roi_pool_backward if literal else lib().orc_roi_pool_backward_scatter
    f(_p(top_diff, ctypes.c_float), _p(argmax, ctypes.c_int32), _p(rois, ctypes.c_float),
      rois.shape[0], N, H, W, C, pooled_h, pooled_w, float(spatial_scale),
      _p(out, ctypes.c_float))
    return out
