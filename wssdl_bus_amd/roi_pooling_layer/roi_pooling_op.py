"""RoiPool / RoiPoolGrad on the GPU.

Reference: the TF custom op loaded by code/lib/roi_pooling_layer/roi_pooling_op.py:4-7
(``roi_pool(bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale)``
-> ``(top_data, argmax)``; ``roi_pool_grad(bottom_data, bottom_rois, argmax, grad, ...)``
-> ``bottom_diff``), registered in roi_pooling_op.cc:31-63, gradient wiring in
roi_pooling_op_grad.py:24-44 (gradient w.r.t. the feature map only).

Layouts are the op's: bottom_data [N,H,W,C] f32 (NHWC), bottom_rois [R,5] f32,
top_data / argmax [R,PH,PW,C].
"""
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
    with torch.cuda.device(g.device), _lib.timed("roi_pool_backward", dict(N=N, H=H, W=W, C=C, R=rois.shape[0])):
        _lib.check(_lib.lib().wssdl_roi_pool_backward(
            _lib.ptr(g), _lib.ptr(arg), _lib.ptr(rois), rois.shape[0], N, H, W, C,
            int(pooled_height), int(pooled_width), float(spatial_scale), _lib.ptr(out),
            _lib.stream()), "wssdl_roi_pool_backward")
    return out.cpu().numpy() if as_np else out


class RoiPoolFunction(torch.autograd.Function):
    """autograd wiring of the pair above (roi_pooling_op_grad.py:24-44): the
    gradient flows to the feature map only; rois get None."""

    @staticmethod
    def forward(ctx, bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale,
                rounding):
        data = bottom_data.contiguous()
        rois = bottom_rois.contiguous()
        top, arg = roi_pool(data, rois, pooled_height, pooled_width, spatial_scale,
                            rounding=rounding)
        ctx.save_for_backward(rois, arg)
        ctx.geom = (tuple(data.shape), pooled_height, pooled_width, spatial_scale)
        ctx.mark_non_differentiable(arg)
        return top, arg

    @staticmethod
    def backward(ctx, grad_top, _grad_arg):
        rois, arg = ctx.saved_tensors
        shape, ph, pw, scale = ctx.geom
        bottom_diff = roi_pool_grad(torch.empty(shape, device="meta"), rois, arg,
                                    grad_top.contiguous(), ph, pw, scale)
        return bottom_diff, None, None, None, None, None


def roi_pool_autograd(bottom_data, bottom_rois, pooled_height, pooled_width, spatial_scale,
                      rounding=None):
    """Differentiable roi_pool: ``(top_data, argmax)``."""
    return RoiPoolFunction.apply(bottom_data, bottom_rois, pooled_height, pooled_width,
                                 spatial_scale, rounding)
