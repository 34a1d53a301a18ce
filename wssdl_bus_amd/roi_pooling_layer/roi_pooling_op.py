"""RoiPool / RoiPoolGrad on the GPU.

Reference: the TF custom op loaded by code/lib/roi_pooling_layer/roi_pooling_op.py:4-7
(``roi_pool(bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale)``
-> ``(top_data, argmax)``; ``roi_pool_grad(bottom_data, bottom_rois, argmax, grad, ...)``
-> ``bottom_diff``), registered in roi_pooling_op.cc:31-63, gradient wiring in
roi_pooling_op_grad.py:24-44 (gradient w.r.t. the feature map only).

Layouts are the op's: bottom_data [N,H,W,C] f32 (NHWC), bottom_rois [R,5] f32,
top_data / argmax [R,PH,PW,C].
"""
import ctypes
import os

import torch

from .. import _lib
from ..fast_rcnn.config import cfg

_ROUNDING = {"cuda": _lib.ROUND_CUDA, "cpu": _lib.ROUND_CPU, 0: 0, 1: 1}


def _check_inputs(bottom_data, bottom_rois):
    # roi_pooling_op.cc:97-102 (OP_REQUIRES rank checks -> InvalidArgument)
    if bottom_data.dim() != 4:
        raise ValueError("data must be 4-dimensional")
    if bottom_rois.dim() != 2:
        raise ValueError("rois must be 2-dimensional")
    if bottom_rois.shape[1] != 5:
        raise ValueError("rois must be [R, 5]")


def roi_pool(bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale, name=None,
             rounding=None):
    """Forward op: returns ``(top_data, argmax)`` like the TF op's two outputs."""
    as_np = _lib.wants_numpy(bottom_data, bottom_rois)
    data = _lib.to_device(bottom_data, torch.float32)
    rois = _lib.to_device(bottom_rois, torch.float32, data.device)
    _check_inputs(data, rois)
    N, H, W, C = data.shape
    R = rois.shape[0]
    mode = _ROUNDING[cfg.ROI_POOL_ROUNDING if rounding is None else rounding]
    top = torch.empty((R, pooled_height, pooled_width, C), dtype=torch.float32, device=data.device)
    arg = torch.empty((R, pooled_height, pooled_width, C), dtype=torch.int32, device=data.device)
    with torch.cuda.device(data.device), _lib.timed("roi_pool_forward", dict(N=N, H=H, W=W, C=C, R=R)):
        _lib.check(_lib.lib().wssdl_roi_pool_forward(
            _lib.ptr(data), N, H, W, C, _lib.ptr(rois), R, int(pooled_height), int(pooled_width),
            float(spatial_scale), mode, _lib.ptr(top), _lib.ptr(arg), _lib.stream()),
            "wssdl_roi_pool_forward")
    if as_np:
        return top.cpu().numpy(), arg.cpu().numpy()
    return top, arg


def roi_pool_grad(bottom_data, bottom_rois, argmax, grad, pooled_height, pooled_width,
                  spatial_scale, name=None):
    """Backward op: gradient w.r.t. bottom_data, shape of bottom_data."""
    as_np = _lib.wants_numpy(bottom_data, bottom_rois, argmax, grad)
    g = _lib.to_device(grad, torch.float32)
    rois = _lib.to_device(bottom_rois, torch.float32, g.device)
    arg = _lib.to_device(argmax, torch.int32, g.device)
    shape = tuple(bottom_data.shape)
    if len(shape) != 4:
        raise ValueError("data must be 4-dimensional")
    if rois.dim() != 2:
        raise ValueError("rois must be 2-dimensional")
    if arg.dim() != 4:
        raise ValueError("argmax_data must be 4-dimensional")
    if g.dim() != 4:
        raise ValueError("out_backprop must be 4-dimensional")
    N, H, W, C = shape
    out = torch.empty(shape, dtype=torch.float32, device=g.device)
    L = _lib.lib()
    with torch.cuda.device(g.device):
        # the list-driven kernels read the i32 arg-max too when they get a workspace for their lists
        # (wssdl_roi_pool_backward_ws; shapes they do not take run the kernel of wssdl_roi_pool_backward)
        nws = L.wssdl_roi_pool_backward_workspace_bytes(rois.shape[0], N, H, W, int(pooled_height), int(pooled_width))
        ws = torch.empty((nws,), dtype=torch.uint8, device=g.device) if nws else None
        with _lib.timed("roi_pool_backward", dict(N=N, H=H, W=W, C=C, R=rois.shape[0])):
            _lib.check(L.wssdl_roi_pool_backward_ws(
                _lib.ptr(g), _lib.ptr(arg), _lib.ptr(rois), rois.shape[0], N, H, W, C,
                int(pooled_height), int(pooled_width), float(spatial_scale), _lib.ptr(out), _lib.ptr(ws), nws,
                _lib.stream()), "wssdl_roi_pool_backward_ws")
    return out.cpu().numpy() if as_np else out


# ----------------------------------------------------- training path: 1-byte arg-max ---
# The arg-max tensor is an internal hand-off between the op and its gradient
# (network.py:206-210 keeps top_data only; roi_pooling_op_grad.py:24-44 feeds argmax back), so the
# autograd pair below agrees on one byte per element instead of four whenever the library
# supports the shape (include/wssdl_bus_hip.h, "training path").  top_data and the gradient are
# bit-identical to the i32 pair; `expand_argmax` rebuilds the reference's i32 indices.


def compact_supported(H, W, C, pooled_height, pooled_width):
    return bool(cfg.ROI_POOL_COMPACT_ARGMAX) and bool(
        _lib.lib().wssdl_roi_pool_compact_supported(int(H), int(W), int(C), int(pooled_height),
                                                    int(pooled_width)))


# Device-side error flags of the compact pair.  The kernels can only raise them; somebody has to
# read them.  [0] = the forward met a bin window larger than 15 x 16 cells (a RoI reaching far
# outside the feature map: its 1-byte code cannot say where the maximum was), [1] = the backward's
# lists did not fit their workspace.  Either one means a wrong gradient, so they are never dropped:
#   * poll_flags()  -- no synchronisation: looks at the copy started by the previous poll once its
#                      event has completed, raises if a flag was up, starts the next copy.  The
#                      train step calls it before every optimiser step (one step of delay).
#   * check_flags() -- synchronises, raises.  Tests, bench.py after its timed region.
#   * cfg.ROI_POOL_FLAG_CHECK = 'eager': RoiPoolFunction reads the forward flag right away (one
#     host read-back per call) and re-runs an overflowing call on the i32 pair, which takes any RoI.
_OVERFLOW_MSG = ("RoI pooling (1-byte arg-max path): a bin window exceeded 15 x 16 cells -- a RoI reaches far "
                 "outside the feature map.  Its arg-max and gradient are invalid; clip the RoIs or use the i32 "
                 "pair (cfg.ROI_POOL_COMPACT_ARGMAX = False or cfg.ROI_POOL_FLAG_CHECK = 'eager').")
_LISTS_MSG = "RoI pooling backward: the per-tile lists overflowed their workspace; bottom_diff is short."
_NMS_TIMEOUT_MSG = ("proposal layer: the NMS sweep of an image gave up waiting for the mask blocks of the fused launch "
                    "(roi count -1 = WSSDL_NMS_TIMED_OUT: the GPU was held by another process or kernel).  That "
                    "image's proposals are incomplete; the step must not be used.")


class _DeviceFlags(object):
    def __init__(self, dev):
        # [0] forward window overflow, [1] backward lists overflow, [2] an NMS sweep timed out (raised by
        # proposal_layer_tf_bus.padded_blob, the one consumer of the roi counts that never reads them back)
        self.flags = torch.zeros((3,), dtype=torch.int32, device=dev)
        self.host = torch.zeros((3,), dtype=torch.int32).pin_memory()
        self.event = None

    @staticmethod
    def _raise(v, strict=True):
        # the time-out first: its handling (two launches from here on, the step counted as tainted) must not be lost
        # when one of the other two flags raises in the same poll
        if int(v[2]) != 0 and not strict:
            # the sync-free training path: that step ran with no proposals for the image whose sweep gave up; from
            # here on mask and sweep run as two launches (no waits between workgroups, identical results)
            from ..rpn_msr.proposal_layer_tf_bus import note_nms_timeout
            note_nms_timeout("deferred flag of the padded path")
            _lib.set_tuning("nms_fused", 0)
            _tainted["steps"] += 1
        if int(v[0]) != 0:
            raise _lib.HipCallError(_OVERFLOW_MSG)
        if int(v[1]) != 0:
            raise _lib.HipCallError(_LISTS_MSG)
        if int(v[2]) != 0 and strict:
            raise _lib.HipCallError(_NMS_TIMEOUT_MSG)

    def read_and_clear(self):
        v = self.flags.cpu()
        self.flags.zero_()
        self.event = None
        return v

    def poll(self):
        if self.event is not None and self.event.query():
            self.event = None
            if int(self.host.abs().sum()) != 0:
                # reported once: the bits this raise names are cleared, bits raised since the copy stay up
                self.flags.bitwise_and_(torch.bitwise_not(self.host.clone().to(self.flags.device)))
            self._raise(self.host, strict=False)
        if self.event is None:
            self.host.copy_(self.flags, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()


_device_flags = {}
_tainted = {"steps": 0}


def _flags(dev):
    f = _device_flags.get(dev)
    if f is None:
        f = _device_flags[dev] = _DeviceFlags(dev)
    return f


def _overflow_flag(dev):
    return _flags(dev).flags[0:1]


def note_roi_counts(counts):
    """Sync-free: raises the deferred flag [2] when any per-image roi count is negative (WSSDL_NMS_TIMED_OUT)."""
    f = _flags(counts.device)
    f.flags[2:3] |= (counts.min() < 0).to(torch.int32)


def poll_flags():
    """Sync-free check of the flags raised up to the previous poll (see above)."""
    for f in _device_flags.values():
        f.poll()


def check_flags():
    """Synchronising check: raises HipCallError if any compact RoI-pool call raised a flag, or if an earlier sync-free
    poll swallowed an NMS time-out (a step that was applied with an image's proposals missing: tainted_steps()).  The
    train loop calls this before it writes a snapshot."""
    for f in list(_device_flags.values()):
        _DeviceFlags._raise(f.read_and_clear())
    if _tainted["steps"]:
        n, _tainted["steps"] = _tainted["steps"], 0
        raise _lib.HipCallError("%d optimiser step(s) since the last check ran with an image's proposals missing (NMS sweep "
                                "time-out seen by the sync-free poll; the process switched to two NMS launches).  " % n
                                + _NMS_TIMEOUT_MSG)


def tainted_steps():
    """Steps applied since the last check_flags() although an NMS sweep had timed out (sync-free path)."""
    return _tainted["steps"]


def flags_raised():
    """True (and the flags are cleared) when any flag is up.  Synchronises; for tests."""
    up = False
    for f in list(_device_flags.values()):
        up = bool(f.read_and_clear().any()) or up
    return up


def compact_overflowed(device=None):
    """True when any compact forward on `device` met a window larger than 15 x 16 cells.  Synchronises."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    f = _device_flags.get(dev)
    return bool(f is not None and int(f.flags[0].item()) != 0)


_WINDOW_TABLE_MIN_ROIS = 1024


def roi_pool_compact(data, rois, pooled_height, pooled_width, spatial_scale, rounding=None):
    """GPU tensors in, ``(top_data f32, argmax8 u8)`` out."""
    _check_inputs(data, rois)
    N, H, W, C = data.shape
    R = rois.shape[0]
    mode = _ROUNDING[cfg.ROI_POOL_ROUNDING if rounding is None else rounding]
    top = torch.empty((R, pooled_height, pooled_width, C), dtype=torch.float32, device=data.device)
    arg8 = torch.empty((R, pooled_height, pooled_width, C), dtype=torch.uint8, device=data.device)
    L = _lib.lib()
    with torch.cuda.device(data.device):
        # (a launch of its own only pays off on a train-sized RoI list)
        nwin = L.wssdl_roi_pool_forward_windows_bytes(R, H, W, C, int(pooled_height), int(pooled_width)) \
            if (R >= _WINDOW_TABLE_MIN_ROIS and _lib.get_tuning("roi_fwd_variant") == 0) else 0
        if nwin:
            # the RoI geometry once per (roi, bin row) into a table, then the pooling kernel reads it with
            # scalar loads (two launches, timed apart: the second is the kernel the roofline is quoted on)
            table = torch.empty((nwin,), dtype=torch.uint8, device=data.device)
            # many proposals per image (large, overlapping windows): block-maximum tables of this step's feature map,
            # four table reads per bin instead of a scan of its cells (csrc/roi_pool_blocks.hip; same bits)
            want_blocks = cfg.get("ROI_POOL_FWD_BLOCKS", "auto")
            if want_blocks == "auto":
                want_blocks = L.wssdl_roi_pool_forward_blocks_auto(R, N, H, W, C, int(pooled_height), int(pooled_width))
            nblk = L.wssdl_roi_pool_forward_blocks_bytes(R, N, H, W, C, int(pooled_height), int(pooled_width)) \
                if want_blocks else 0
            if nblk:
                blocks = torch.empty((nblk,), dtype=torch.uint8, device=data.device)
                with _lib.timed("roi_pool_forward_windows", dict(R=R)):
                    _lib.check(L.wssdl_roi_pool_forward_windows_blocks(
                        _lib.ptr(rois), R, N, H, W, C, int(pooled_height), int(pooled_width), float(spatial_scale), mode,
                        _lib.ptr(table), nwin, _lib.ptr(_overflow_flag(data.device)), _lib.ptr(blocks), nblk,
                        _lib.stream()), "wssdl_roi_pool_forward_windows_blocks")
                with _lib.timed("roi_pool_forward_blocks_prepare", dict(N=N, H=H, W=W, C=C, R=R)):
                    _lib.check(L.wssdl_roi_pool_forward_blocks_prepare(
                        _lib.ptr(data), N, H, W, C, R, int(pooled_height), int(pooled_width), _lib.ptr(table),
                        _lib.ptr(blocks), nblk, _lib.stream()), "wssdl_roi_pool_forward_blocks_prepare")
                with _lib.timed("roi_pool_forward", dict(N=N, H=H, W=W, C=C, R=R, argmax_bytes=1, blocks=1)):
                    _lib.check(L.wssdl_roi_pool_forward_compact_blocks(
                        _lib.ptr(data), N, H, W, C, R, int(pooled_height), int(pooled_width), _lib.ptr(table),
                        _lib.ptr(blocks), nblk, _lib.ptr(top), _lib.ptr(arg8), _lib.stream()),
                        "wssdl_roi_pool_forward_compact_blocks")
                return top, arg8
            with _lib.timed("roi_pool_forward_windows", dict(R=R)):
                _lib.check(L.wssdl_roi_pool_forward_windows(
                    _lib.ptr(rois), R, N, H, W, C, int(pooled_height), int(pooled_width), float(spatial_scale), mode,
                    _lib.ptr(table), nwin, _lib.ptr(_overflow_flag(data.device)), _lib.stream()),
                    "wssdl_roi_pool_forward_windows")
            with _lib.timed("roi_pool_forward", dict(N=N, H=H, W=W, C=C, R=R, argmax_bytes=1)):
                _lib.check(L.wssdl_roi_pool_forward_compact_windows(
                    _lib.ptr(data), N, H, W, C, _lib.ptr(rois), R, int(pooled_height), int(pooled_width),
                    float(spatial_scale), mode, _lib.ptr(table), _lib.ptr(top), _lib.ptr(arg8), _lib.stream()),
                    "wssdl_roi_pool_forward_compact_windows")
        else:
            with _lib.timed("roi_pool_forward", dict(N=N, H=H, W=W, C=C, R=R, argmax_bytes=1)):
                _lib.check(L.wssdl_roi_pool_forward_compact(
                    _lib.ptr(data), N, H, W, C, _lib.ptr(rois), R, int(pooled_height), int(pooled_width),
                    float(spatial_scale), mode, _lib.ptr(top), _lib.ptr(arg8), _lib.ptr(_overflow_flag(data.device)),
                    _lib.stream()), "wssdl_roi_pool_forward_compact")
    return top, arg8


class BackwardPlan(object):
    """What wssdl_roi_pool_backward_prepare left behind: the workspace holding the per-tile lists
    and the plan id the walk kernel must be launched with (-1: no lists, the backward filters
    the RoIs itself)."""

    def __init__(self, workspace, nbytes, plan, owner=-1):
        # owner >= 0: the lists were built for that plan of the bin-owner form (roi_pool_grad_prepare_owner)
        self.workspace, self.nbytes, self.plan, self.owner = workspace, nbytes, plan, owner
        self.owner_segments = 1        # > 1: the owner form with that many waves per tile stream (round 6)
        self.segments, self.variant = 1, "exact walk: the reference's summation order, bit for bit"


def roi_pool_grad_prepare(shape, rois, pooled_height, pooled_width, spatial_scale, rounding=None, segments=1):
    """Build the lists that drive the compact backward.  They depend on the RoIs and the shapes
    only, so the autograd pair does this right behind the forward (off the backward's path).
    `segments` > 1 (the split form will walk them): the lists are built for the plan that form wants."""
    if segments > 1 and _lib.get_tuning("roi_bwd_plan") < 0:
        with _lib.tuned(roi_bwd_plan=int(_lib.lib().wssdl_roi_pool_backward_split_plan())):
            return roi_pool_grad_prepare(shape, rois, pooled_height, pooled_width, spatial_scale, rounding)
    N, H, W, C = shape
    mode = _ROUNDING[cfg.ROI_POOL_ROUNDING if rounding is None else rounding]
    L = _lib.lib()
    R = rois.shape[0]
    plan = ctypes.c_int32(-1)
    with torch.cuda.device(rois.device):
        nws = L.wssdl_roi_pool_backward_workspace_bytes(R, N, H, W, int(pooled_height), int(pooled_width))
        ws = torch.empty((nws,), dtype=torch.uint8, device=rois.device) if nws else None
        with _lib.timed("roi_pool_backward_prepare", dict(N=N, H=H, W=W, C=C, R=R)):
            _lib.check(L.wssdl_roi_pool_backward_prepare(
                _lib.ptr(rois), R, N, H, W, C, int(pooled_height), int(pooled_width), float(spatial_scale),
                mode, _lib.ptr(ws), nws, ctypes.byref(plan), _lib.stream()), "wssdl_roi_pool_backward_prepare")
            if plan.value >= 0:
                # the status block's error word joins the deferred flags (device-side OR, no read-back)
                off = L.wssdl_roi_pool_backward_status_offset(R, N, H, W, int(pooled_height), int(pooled_width))
                _flags(rois.device).flags[1:2].bitwise_or_(ws[off + 4:off + 8].view(torch.int32))
    return BackwardPlan(ws, nws, int(plan.value))


def owner_plan(shape, R, pooled_height=7, pooled_width=7):
    """The owner plan the bin-owner form of the list-driven backward should use for this launch, or -1 to keep
    the exact walk: cfg.ROI_POOL_BWD_OWNER = 'auto' (the library's rule by launch shape), an int plan id, or -1 / False
    for never.  >= 0 is deterministic but NOT bit-identical to the reference's summation order (like the split form)."""
    v = cfg.get("ROI_POOL_BWD_OWNER", "auto")
    N, H, W, C = shape
    if cfg.get("ROI_POOL_BWD_EXACT", False):
        return -1
    if v == "auto":
        # (-1 too when the owner form does not take the launch: pooled sizes above 8, very large R * C)
        return int(_lib.lib().wssdl_roi_pool_backward_owner_plan_for(int(R), N, H, W, C, int(pooled_height),
                                                                      int(pooled_width)))
    if v is False or v is None:
        return -1
    return int(v)


def owner_segments(shape, R):
    """Waves per tile stream of the bin-owner form for this launch: cfg.ROI_POOL_BWD_OWNER_SEGMENTS = 'auto' (the
    library's rule, wssdl_roi_pool_backward_owner_segments) or an int; 1 = the plain owner form."""
    v = cfg.get("ROI_POOL_BWD_OWNER_SEGMENTS", "auto")
    N, H, W, C = shape
    if v == "auto":
        return max(1, int(_lib.lib().wssdl_roi_pool_backward_owner_segments(int(R), N, H, W, C)))
    return max(1, int(v))


def roi_pool_grad_prepare_owner(shape, rois, pooled_height, pooled_width, spatial_scale, owner, rounding=None):
    """Lists of the bin-owner form (every bin listed once, by the tile of its window's first cell)."""
    N, H, W, C = shape
    mode = _ROUNDING[cfg.ROI_POOL_ROUNDING if rounding is None else rounding]
    L = _lib.lib()
    R = rois.shape[0]
    with torch.cuda.device(rois.device):
        nws = L.wssdl_roi_pool_backward_workspace_bytes(R, N, H, W, int(pooled_height), int(pooled_width))
        ws = torch.empty((nws,), dtype=torch.uint8, device=rois.device)
        with _lib.timed("roi_pool_backward_prepare", dict(N=N, H=H, W=W, C=C, R=R, owner=int(owner))):
            _lib.check(L.wssdl_roi_pool_backward_owner_prepare(
                _lib.ptr(rois), R, N, H, W, C, int(pooled_height), int(pooled_width), float(spatial_scale),
                mode, _lib.ptr(ws), nws, int(owner), _lib.stream()), "wssdl_roi_pool_backward_owner_prepare")
            off = L.wssdl_roi_pool_backward_status_offset(R, N, H, W, int(pooled_height), int(pooled_width))
            _flags(rois.device).flags[1:2].bitwise_or_(ws[off + 4:off + 8].view(torch.int32))
    return BackwardPlan(ws, nws, -1, owner=int(owner))


def prepare_backward(shape, rois, pooled_height, pooled_width, spatial_scale, rounding=None):
    """The lists of the form the backward of this launch will take (cfg.ROI_POOL_BWD_EXACT / _OWNER / _SPLIT): the
    bin-owner form where the library suggests it, else the split form where it suggests that, else the exact walk.
    The returned plan carries `.variant` (a description for logs and bench lines) and `.segments`."""
    R = rois.shape[0]
    own = owner_plan(shape, R, pooled_height, pooled_width)
    if own >= 0:
        plan = roi_pool_grad_prepare_owner(shape, rois, pooled_height, pooled_width, spatial_scale, own, rounding)
        plan.segments = 1
        plan.owner_segments = owner_segments(shape, R)
        plan.variant = "bin-owner walk, owner plan %d%s: deterministic, not bit-ordered (<= 1e-6 of the exact walk)" % (
            own, (", %d waves per tile stream" % plan.owner_segments) if plan.owner_segments > 1 else "")
        return plan
    segs = split_segments(shape, R)
    plan = roi_pool_grad_prepare(shape, rois, pooled_height, pooled_width, spatial_scale, rounding, segments=segs)
    if plan.plan < 0:
        segs = 1
    plan.segments = segs
    plan.variant = ("exact walk: the reference's summation order, bit for bit" if segs <= 1 else
                    "split walk, %d segments: deterministic, not bit-ordered (<= 1e-6 of the exact walk)" % segs)
    return plan


def split_segments(shape, R):
    """How many segments the list-driven backward cuts a tile's slot stream into for this launch:
    cfg.ROI_POOL_BWD_SPLIT = 'auto' (the library's rule, wssdl_roi_pool_backward_split_segments: at most 4 images with
    >= 1000 RoIs each -> 8 segments, 4 for two / three images x 1024 channels, else 1), an int, or
    0 / 1 for the exact walk.  > 1 is deterministic but NOT bit-identical to the reference's summation order."""
    v = cfg.get("ROI_POOL_BWD_SPLIT", "auto")
    N, H, W, C = shape
    if cfg.get("ROI_POOL_BWD_EXACT", False):
        return 1
    if v == "auto":
        return int(_lib.lib().wssdl_roi_pool_backward_split_segments(int(R), N, H, W, C))
    return max(1, int(v))


def roi_pool_grad_compact(shape, rois, arg8, grad, pooled_height, pooled_width, spatial_scale,
                          rounding=None, use_workspace=True, plan=None, segments=1):
    """bottom_diff from the 1-byte arg-max.  `plan` = what roi_pool_grad_prepare returned for these
    RoIs (prepared here when None and use_workspace).  `segments` > 1: the split form
    (wssdl_roi_pool_backward_compact_split; deterministic, not bit-ordered -- see split_segments)."""
    N, H, W, C = shape
    mode = _ROUNDING[cfg.ROI_POOL_ROUNDING if rounding is None else rounding]
    out = torch.empty(shape, dtype=torch.float32, device=grad.device)
    L = _lib.lib()
    R = rois.shape[0]
    if plan is None and use_workspace:
        plan = roi_pool_grad_prepare(shape, rois, pooled_height, pooled_width, spatial_scale, rounding)
    if plan is not None and plan.owner >= 0 and getattr(plan, "owner_segments", 1) > 1:
        nseg = int(plan.owner_segments)
        with torch.cuda.device(grad.device):
            nscr = L.wssdl_roi_pool_backward_owner_split_scratch_bytes(N, H, W, C, plan.owner, nseg)
            scratch = torch.empty((nscr,), dtype=torch.uint8, device=grad.device)
            with _lib.timed("roi_pool_backward", dict(N=N, H=H, W=W, C=C, R=R, argmax_bytes=1, owner=plan.owner,
                                                      owner_segments=nseg)):
                _lib.check(L.wssdl_roi_pool_backward_compact_owner_split(
                    _lib.ptr(grad), _lib.ptr(arg8), _lib.ptr(rois), R, N, H, W, C,
                    int(pooled_height), int(pooled_width), float(spatial_scale), mode, _lib.ptr(out),
                    _lib.ptr(plan.workspace), plan.nbytes, plan.owner, nseg, _lib.ptr(scratch), nscr,
                    _lib.stream()), "wssdl_roi_pool_backward_compact_owner_split")
        return out
    if plan is not None and plan.owner >= 0:
        with torch.cuda.device(grad.device):
            nscr = L.wssdl_roi_pool_backward_owner_scratch_bytes(N, H, W, C, plan.owner)
            scratch = torch.empty((nscr,), dtype=torch.uint8, device=grad.device)
            with _lib.timed("roi_pool_backward", dict(N=N, H=H, W=W, C=C, R=R, argmax_bytes=1, owner=plan.owner)):
                _lib.check(L.wssdl_roi_pool_backward_compact_owner(
                    _lib.ptr(grad), _lib.ptr(arg8), _lib.ptr(rois), R, N, H, W, C,
                    int(pooled_height), int(pooled_width), float(spatial_scale), mode, _lib.ptr(out),
                    _lib.ptr(plan.workspace), plan.nbytes, plan.owner, _lib.ptr(scratch), nscr,
                    _lib.stream()), "wssdl_roi_pool_backward_compact_owner")
        return out
    if plan is None or plan.plan < 0:
        plan = BackwardPlan(None, 0, -1)
    segments = int(segments) if plan.plan >= 0 else 1
    if segments > 1:
        with torch.cuda.device(grad.device):
            nscr = L.wssdl_roi_pool_backward_split_scratch_bytes(N, H, W, C, segments)
            scratch = torch.empty((nscr,), dtype=torch.uint8, device=grad.device)
            with _lib.timed("roi_pool_backward", dict(N=N, H=H, W=W, C=C, R=R, argmax_bytes=1, plan=plan.plan,
                                                      segments=segments)):
                _lib.check(L.wssdl_roi_pool_backward_compact_split(
                    _lib.ptr(grad), _lib.ptr(arg8), _lib.ptr(rois), R, N, H, W, C,
                    int(pooled_height), int(pooled_width), float(spatial_scale), mode, _lib.ptr(out),
                    _lib.ptr(plan.workspace), plan.nbytes, plan.plan, segments, _lib.ptr(scratch), nscr,
                    _lib.stream()), "wssdl_roi_pool_backward_compact_split")
        return out
    with torch.cuda.device(grad.device):
        with _lib.timed("roi_pool_backward", dict(N=N, H=H, W=W, C=C, R=R, argmax_bytes=1, plan=plan.plan)):
            _lib.check(L.wssdl_roi_pool_backward_compact(
                _lib.ptr(grad), _lib.ptr(arg8), _lib.ptr(rois), R, N, H, W, C,
                int(pooled_height), int(pooled_width), float(spatial_scale), mode, _lib.ptr(out),
                _lib.ptr(plan.workspace), plan.nbytes, plan.plan, _lib.stream()),
                "wssdl_roi_pool_backward_compact")
    return out


def expand_argmax(arg8, rois, shape, pooled_height, pooled_width, spatial_scale, rounding=None):
    """1-byte codes -> the reference's i32 argmax (flat NHWC index inside the image, -1 empty)."""
    N, H, W, C = shape
    mode = _ROUNDING[cfg.ROI_POOL_ROUNDING if rounding is None else rounding]
    out = torch.empty(tuple(arg8.shape), dtype=torch.int32, device=arg8.device)
    with torch.cuda.device(arg8.device):
        _lib.check(_lib.lib().wssdl_roi_argmax_expand(
            _lib.ptr(arg8), _lib.ptr(rois), rois.shape[0], H, W, C, int(pooled_height),
            int(pooled_width), float(spatial_scale), mode, _lib.ptr(out), _lib.stream()),
            "wssdl_roi_argmax_expand")
    return out


_announced_forms = set()


def _announce_backward_form(shape, R, plan):
    """Once per process and launch class: which form the training gradient takes (the default is tolerance parity --
    the bin-owner / split forms -- not the reference's bit order; cfg.ROI_POOL_BWD_EXACT = True switches it)."""
    key = (shape[0], shape[3], plan.variant)
    if key in _announced_forms or not cfg.get("ROI_POOL_ANNOUNCE_BWD_FORM", True):
        return
    _announced_forms.add(key)
    import sys
    print("[wssdl_bus_amd] RoI-pool backward, %d images x %d channels, %d RoIs: %s" % (shape[0], shape[3], R, plan.variant),
          file=sys.stderr)


class RoiPoolFunction(torch.autograd.Function):
    """autograd wiring of the pair above (roi_pooling_op_grad.py:24-44): the
    gradient flows to the feature map only; rois get None.  The second output is the arg-max
    the backward will read: u8 codes on the compact path, the reference's i32 otherwise."""

    @staticmethod
    def forward(ctx, bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale,
                rounding):
        data = bottom_data.contiguous()
        rois = bottom_rois.contiguous()
        _check_inputs(data, rois)
        rounding = cfg.ROI_POOL_ROUNDING if rounding is None else rounding     # fixed for the pair
        ctx.compact = data.is_cuda and data.dtype == torch.float32 and rois.dtype == torch.float32 and \
            compact_supported(data.shape[1], data.shape[2], data.shape[3], pooled_height, pooled_width)
        ctx.plan = None
        top = arg = None
        if ctx.compact:
            top, arg = roi_pool_compact(data, rois, pooled_height, pooled_width, spatial_scale, rounding)
            if cfg.get('ROI_POOL_FLAG_CHECK', 'deferred') == 'eager' and compact_overflowed(data.device):
                # a RoI the 1-byte codes cannot describe: this call runs on the i32 pair (any RoI)
                _flags(data.device).flags[0:1].zero_()
                ctx.compact = False
                top = arg = None
        if ctx.compact:
            if data.requires_grad or bottom_data.requires_grad:
                # the backward's lists depend on the RoIs only: build them now, behind the forward
                ctx.plan = prepare_backward(tuple(data.shape), rois, pooled_height, pooled_width, spatial_scale, rounding)
                _announce_backward_form(tuple(data.shape), rois.shape[0], ctx.plan)
        else:
            top, arg = roi_pool(data, rois, pooled_height, pooled_width, spatial_scale,
                                rounding=rounding)
        ctx.save_for_backward(rois, arg)
        ctx.geom = (tuple(data.shape), pooled_height, pooled_width, spatial_scale, rounding)
        ctx.mark_non_differentiable(arg)
        return top, arg

    @staticmethod
    def backward(ctx, grad_top, _grad_arg):
        rois, arg = ctx.saved_tensors
        shape, ph, pw, scale, rounding = ctx.geom
        if ctx.compact:
            bottom_diff = roi_pool_grad_compact(shape, rois, arg, grad_top.contiguous(), ph, pw, scale,
                                                rounding, plan=ctx.plan, segments=getattr(ctx.plan, "segments", 1))
        else:
            bottom_diff = roi_pool_grad(torch.empty(shape, device="meta"), rois, arg,
                                        grad_top.contiguous(), ph, pw, scale)
        return bottom_diff, None, None, None, None, None


def roi_pool_autograd(bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale,
                      rounding=None, return_argmax=True):
    """Differentiable roi_pool: ``(top_data, argmax)``.  `argmax` is the reference's i32 tensor
    (expanded from the 1-byte codes when the compact path ran); pass return_argmax=False -- as the
    network layer does, which like network.py:206-210 keeps top_data only -- to skip that."""
    top, arg = RoiPoolFunction.apply(bottom_data, bottom_rois, pooled_height, pooled_width,
                                     spatial_scale, rounding)
    if not return_argmax:
        return top, None
    if arg.dtype == torch.uint8:
        arg = expand_argmax(arg, bottom_rois.contiguous(), tuple(bottom_data.shape), pooled_height,
                            pooled_width, spatial_scale, rounding)
    return top, arg
