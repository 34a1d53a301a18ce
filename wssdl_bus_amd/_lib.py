"""ctypes binding of ``libwssdl_bus_hip.so`` (C ABI: ``include/wssdl_bus_hip.h``).

The product path has NO fallback: if the library is missing this module raises,
and every op in this package goes through it.  PyTorch is only used here for
device memory (``tensor.data_ptr()``) and the current HIP stream.
"""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WSSDL_BUS_HIP_LIB") or os.path.join(_HERE, "libwssdl_bus_hip.so")

OK, ERR_INVALID_ARGUMENT, ERR_WORKSPACE, ERR_LAUNCH = 0, 1, 2, 3
ROUND_CUDA, ROUND_CPU = 0, 1
DATASET_SNUBH, DATASET_SNUBH_FG, DATASET_FG_ONLY = 0, 1, 2
MAX_ANCHORS, MAX_GT = 32, 64

# every symbol include/wssdl_bus_hip.h declares: (restype, argtypes)
_vp, _i, _i64, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double
_sz, _u64 = ctypes.c_size_t, ctypes.c_uint64
SYMBOLS = {
    "wssdl_version": (ctypes.c_char_p, []),
    "wssdl_last_error": (ctypes.c_char_p, []),
    "wssdl_set_tuning": (_i, [ctypes.c_char_p, _i]),
    "wssdl_get_tuning": (_i, [ctypes.c_char_p, _vp]),
    "wssdl_generate_anchors_host": (_i, [_i, _vp, _i, _vp, _i, _vp]),
    "wssdl_shifted_anchors": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "wssdl_bbox_overlaps": (_i, [_vp, _i64, _i, _vp, _i64, _i, _vp, _vp]),
    "wssdl_bbox_overlaps_ui": (_i, [_vp, _i64, _i, _vp, _i64, _i, _vp, _vp]),
    "wssdl_nms_workspace_bytes": (_sz, [_i]),
    "wssdl_nms": (_i, [_vp, _i, _d, _i, _vp, _vp, _vp, _sz, _vp]),
    "wssdl_nms_new": (_i, [_vp, _i, _d, _i, _vp, _vp, _vp, _sz, _vp]),
    "wssdl_proposal_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "wssdl_proposal_layer": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _d, _f,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "wssdl_proposal_layer_from_logits": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _d, _f,
                                              _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "wssdl_proposal_compact": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp]),
    "wssdl_anchor_workspace_bytes": (_sz, [_i]),
    "wssdl_anchor_labels": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _i, _i, _d, _d, _i,
                                 _vp, _vp, _vp, _vp, _sz, _vp]),
    "wssdl_anchor_subsample_device": (_i, [_vp, _i, _i, _i, _d, _u64, _vp, _vp]),
    "wssdl_anchor_targets": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _d,
                                  _vp, _vp, _vp, _vp, _vp]),
    "wssdl_roi_gt_assign": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp]),
    "wssdl_roi_candidates": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _vp, _vp]),
    "wssdl_roi_sample_device": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _d, _d, _d, _u64, _vp, _vp, _vp, _vp]),
    "wssdl_proposal_target_device_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "wssdl_proposal_target_device": (_i, [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _d, _d, _d, _u64, _i, _vp, _vp,
                                          _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "wssdl_roi_targets": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               _vp]),
    "wssdl_roi_pool_forward": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "wssdl_roi_pool_backward": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "wssdl_roi_pool_backward_ws": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _sz, _vp]),
    "wssdl_roi_pool_compact_supported": (_i, [_i, _i, _i, _i, _i]),
    "wssdl_roi_pool_forward_compact": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "wssdl_roi_pool_forward_windows_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "wssdl_roi_pool_forward_windows": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp, _vp]),
    "wssdl_roi_pool_forward_compact_windows": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp]),
    "wssdl_roi_pool_forward_blocks_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "wssdl_roi_pool_forward_blocks_auto": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "wssdl_roi_pool_forward_windows_blocks": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp, _vp, _sz, _vp]),
    "wssdl_roi_pool_forward_blocks_prepare": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "wssdl_roi_pool_forward_compact_blocks": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp, _vp, _vp]),
    "wssdl_roi_pool_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "wssdl_roi_pool_backward_plan_count": (_i, []),
    "wssdl_roi_pool_backward_status_offset": (_sz, [_i, _i, _i, _i, _i, _i]),
    "wssdl_roi_pool_backward_prepare": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _vp, _vp]),
    "wssdl_roi_pool_backward_compact": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _sz, _i,
                                             _vp]),
    "wssdl_roi_pool_backward_split_segments": (_i, [_i, _i, _i, _i, _i]),
    "wssdl_roi_pool_backward_split_plan": (_i, []),
    "wssdl_roi_pool_backward_split_scratch_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "wssdl_roi_pool_backward_compact_split": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _sz, _i, _i,
                                                   _vp, _sz, _vp]),
    "wssdl_roi_pool_backward_owner_plan_count": (_i, []),
    "wssdl_roi_pool_backward_owner_plan": (_i, [_i, _i, _i, _i, _i]),
    "wssdl_roi_pool_backward_owner_plan_for": (_i, [_i, _i, _i, _i, _i, _i, _i]),
    "wssdl_roi_pool_backward_owner_scratch_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "wssdl_roi_pool_backward_owner_prepare": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _sz, _i, _vp]),
    "wssdl_roi_pool_backward_compact_owner": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _sz, _i,
                                                   _vp, _sz, _vp]),
    "wssdl_roi_pool_backward_owner_segments": (_i, [_i, _i, _i, _i, _i]),
    "wssdl_roi_pool_backward_owner_split_scratch_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "wssdl_roi_pool_backward_compact_owner_split": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _sz, _i, _i, _vp, _sz, _vp]),
    "wssdl_roi_pool_backward_owner_i32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp, _sz, _i, _vp, _sz, _vp]),
    "wssdl_roi_argmax_expand": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "wssdl_image_prep_workspace_bytes": (_sz, []),
    "wssdl_image_prep": (_i, [_vp, _i, _i, _i, _i, _i, _f, _i, _f, _d, _vp, _vp, _sz, _vp]),
    "wssdl_image_to_blob": (_i, [_vp, _i, _i, _i, _d, _i, _vp, _i, _i, _i, _i, _vp]),
    "wssdl_image_adjust_f64": (_i, [_vp, _i, _i, _i, _i64, _i, _d, _i, _d, _d, _vp, _vp, _sz, _vp]),
    "wssdl_image_warp_workspace_bytes": (_sz, []),
    "wssdl_image_warp": (_i, [_vp, _i, _i, _i, _i, _vp, _i, _i, _i, _d, _i, _vp, _vp, _sz, _vp]),
    "wssdl_image_resize": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "wssdl_flip_boxes": (_i, [_vp, _i, _i, _f, _vp]),
    "wssdl_post_detections_workspace_bytes": (_sz, [_i, _i]),
    "wssdl_post_detections": (_i, [_vp, _vp, _i, _i, _f, _d, _i, _vp, _vp, _vp, _sz, _vp]),
    "wssdl_mil_select": (_i, [_vp, _i, _i, _vp, _i, _f, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "wssdl_mil_loss_forward": (_i, [_vp, _i, _i, _vp, _i, _f, _vp, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _vp]),
    "wssdl_mil_loss_backward": (_i, [_vp, _i, _i, _vp, _i, _f, _vp, _i, _vp, _vp, _f, _vp, _vp, _vp]),
    "wssdl_multi_task_loss_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "wssdl_multi_task_loss_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                                           _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "wssdl_multi_task_loss_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i,
                                            _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp,
                                            _vp, _vp, _vp, _vp, _vp, _vp]),
}

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


class HipCallError(RuntimeError):
    pass


def lib():
    """The loaded library.  Raises HipLibraryMissing when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                "%s not found: build it with `python -m wssdl_bus_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)        # AttributeError if the .so lacks a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


_STATUS = {1: "invalid argument", 2: "workspace too small", 3: "HIP launch error"}


def check(rc, what):
    if rc != OK:
        err = lib().wssdl_last_error().decode()
        raise HipCallError("%s failed: %s%s" % (what, _STATUS.get(rc, "status %d" % rc),
                                                (" (%s)" % err) if err else ""))


def set_tuning(key, value):
    """wssdl_set_tuning: the library's knobs are set here, never through the environment."""
    check(lib().wssdl_set_tuning(key.encode(), int(value)), "wssdl_set_tuning(%s)" % key)


def get_tuning(key):
    v = ctypes.c_int(0)
    check(lib().wssdl_get_tuning(key.encode(), ctypes.byref(v)), "wssdl_get_tuning(%s)" % key)
    return int(v.value)


class tuned(object):
    """with tuned(roi_bwd_plan=11): ...   -- sets knobs and restores them."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: get_tuning(k) for k in self.kv}
        for k, v in self.kv.items():
            set_tuning(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_tuning(k, v)
        return False


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def host_ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipCallError("expected a tensor on the GPU, got device %s" % t.device)


def to_device(x, dtype, device=None):
    """numpy array / tensor -> contiguous GPU tensor of `dtype` (the py_func
    boundary of the reference hands the layers numpy copies, network.py:216)."""
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(x))
    dev = device if device is not None else (t.device if t.is_cuda else torch.device("cuda", torch.cuda.current_device()))
    return t.to(device=dev, dtype=dtype).contiguous()


def wants_numpy(*xs):
    """True when the caller passed numpy data (then outputs go back as numpy)."""
    return not any(isinstance(x, torch.Tensor) for x in xs if x is not None)


def generate_anchors_host(base_size, ratios, scales):
    r = np.ascontiguousarray(ratios, dtype=np.float64).ravel()
    s = np.ascontiguousarray(scales, dtype=np.float64).ravel()
    out = np.zeros((len(r) * len(s), 4), dtype=np.float64)
    n = lib().wssdl_generate_anchors_host(int(base_size), host_ptr(r), len(r), host_ptr(s), len(s),
                                          host_ptr(out))
    if n < 0:
        check(-n, "wssdl_generate_anchors_host")
    return out


# --------------------------------------------------------------------------
# Optional HIP-event instrumentation of the C-ABI calls (used by bench.py's
# `roofline` leg).  Events are recorded on the stream the kernels are launched
# on (torch's current stream), so they bracket exactly the enqueued kernels.
class _Timeline(object):
    def __init__(self):
        self.enabled = False
        self.records = []           # (name, start_event, end_event, meta)

    def reset(self, enabled):
        self.enabled = enabled
        self.records = []

    def summary(self):
        """name -> dict(calls, total_ms, avg_ms, metas).  Call after a synchronize."""
        out = {}
        for name, s, e, meta in self.records:
            d = out.setdefault(name, dict(calls=0, total_ms=0.0, metas=[]))
            d["calls"] += 1
            d["total_ms"] += s.elapsed_time(e)
            d["metas"].append(meta)
        for d in out.values():
            d["avg_ms"] = d["total_ms"] / max(d["calls"], 1)
        return out


timeline = _Timeline()


class timed(object):
    """with timed('roi_pool_forward', meta): <enqueue kernels>"""

    def __init__(self, name, meta=None):
        self.name, self.meta = name, meta

    def __enter__(self):
        if timeline.enabled:
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
        return self

    def __exit__(self, *exc):
        if timeline.enabled:
            self.e.record()
            timeline.records.append((self.name, self.s, self.e, self.meta))
        return False
