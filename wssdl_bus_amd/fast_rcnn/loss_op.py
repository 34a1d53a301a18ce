"""The four supervised loss terms as ONE device op with its own backward.

Reference: code/lib/fast_rcnn/train_bus.py:186-235 (alternating mode) == :605-647 (combined
mode, where the RPN box term covers the first IMS_PER_BATCH images): ~40 TF element-wise and
reduction ops; here one forward launch (+ a one-workgroup finish) and one backward launch of
`csrc/loss.hip`.  The op takes the layers as the network produces them -- `rpn_cls_score`
[N,H,W,2A] instead of its `rpn_cls_score_reshape` view (network.py:283-291 is an index map) -- so
neither the reshape nor the NCHW -> NHWC permutes of the targets are materialised.
"""
import torch

from .. import _lib

TERMS = ("rpn_cross_entropy", "rpn_loss_box", "cross_entropy", "loss_box")


def _i32(t):
    return t if t.dtype == torch.int32 else t.to(torch.int32)


class _MultiTaskLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rpn_cls_score, rpn_bbox_pred, cls_score, bbox_pred, rpn_labels, rpn_tg, rpn_inw, rpn_outw,
                labels, tg, inw, outw, n_box_images):
        _lib.require_cuda(rpn_cls_score, rpn_bbox_pred, cls_score, bbox_pred, rpn_labels, rpn_tg, rpn_inw, rpn_outw,
                          labels, tg, inw, outw)
        L = _lib.lib()
        N, H, W, A2 = rpn_cls_score.shape
        A = A2 // 2
        if tuple(rpn_bbox_pred.shape) != (N, H, W, 4 * A):
            raise _lib.HipCallError("rpn_bbox_pred %s does not match rpn_cls_score %s" %
                                    (tuple(rpn_bbox_pred.shape), tuple(rpn_cls_score.shape)))
        if rpn_labels.numel() != N * A * H * W or tuple(rpn_tg.shape) != (N, 4 * A, H, W):
            raise _lib.HipCallError("rpn-data shapes do not match the score map")
        K = cls_score.shape[1]
        n_rows = labels.numel()
        if n_rows > cls_score.shape[0] or tuple(tg.shape) != (n_rows, 4 * K) or bbox_pred.shape[1] != 4 * K:
            raise _lib.HipCallError("roi-data shapes do not match the head outputs")
        f32 = torch.float32
        t = [x.contiguous() if x.dtype == f32 else x.to(f32).contiguous()
             for x in (rpn_cls_score, rpn_bbox_pred, cls_score, bbox_pred, rpn_tg, rpn_inw, rpn_outw, tg, inw, outw)]
        rpn_cls_score, rpn_bbox_pred, cls_score, bbox_pred, rpn_tg, rpn_inw, rpn_outw, tg, inw, outw = t
        rpn_labels = _i32(rpn_labels).contiguous()
        labels = _i32(labels).reshape(-1).contiguous()
        dev = rpn_cls_score.device
        nb = int(n_box_images) if n_box_images is not None else N
        ws = torch.empty((L.wssdl_multi_task_loss_workspace_bytes(N, H, W, A),), dtype=torch.uint8, device=dev)
        losses = torch.empty((4,), dtype=f32, device=dev)
        with torch.cuda.device(dev), _lib.timed("multi_task_loss", dict(N=N, H=H, W=W, A=A, rows=n_rows)):
            _lib.check(L.wssdl_multi_task_loss_forward(
                _lib.ptr(rpn_cls_score), _lib.ptr(rpn_labels), _lib.ptr(rpn_bbox_pred), _lib.ptr(rpn_tg),
                _lib.ptr(rpn_inw), _lib.ptr(rpn_outw), N, nb, H, W, A, _lib.ptr(cls_score), _lib.ptr(labels),
                _lib.ptr(bbox_pred), _lib.ptr(tg), _lib.ptr(inw), _lib.ptr(outw), n_rows, K, _lib.ptr(losses),
                _lib.ptr(ws), ws.numel(), _lib.stream()), "wssdl_multi_task_loss_forward")
        ctx.save_for_backward(rpn_cls_score, rpn_labels, rpn_bbox_pred, rpn_tg, rpn_inw, rpn_outw, cls_score, labels,
                              bbox_pred, tg, inw, outw, ws)
        ctx.dims = (N, nb, H, W, A, n_rows, K)
        return losses

    @staticmethod
    def backward(ctx, grad_losses):
        (rpn_cls_score, rpn_labels, rpn_bbox_pred, rpn_tg, rpn_inw, rpn_outw, cls_score, labels, bbox_pred, tg, inw,
         outw, ws) = ctx.saved_tensors
        N, nb, H, W, A, n_rows, K = ctx.dims
        L = _lib.lib()
        gl = grad_losses.to(torch.float32).contiguous()
        g_rpn_cls = torch.empty_like(rpn_cls_score)
        g_rpn_box = torch.empty_like(rpn_bbox_pred)
        g_cls = torch.empty_like(cls_score)
        g_box = torch.empty_like(bbox_pred)
        with torch.cuda.device(gl.device), _lib.timed("multi_task_loss_backward", dict(N=N, H=H, W=W, A=A, rows=n_rows)):
            _lib.check(L.wssdl_multi_task_loss_backward(
                _lib.ptr(rpn_cls_score), _lib.ptr(rpn_labels), _lib.ptr(rpn_bbox_pred), _lib.ptr(rpn_tg),
                _lib.ptr(rpn_inw), _lib.ptr(rpn_outw), N, nb, H, W, A, _lib.ptr(cls_score), _lib.ptr(labels),
                _lib.ptr(bbox_pred), _lib.ptr(tg), _lib.ptr(inw), _lib.ptr(outw), n_rows, cls_score.shape[0], K,
                _lib.ptr(gl), _lib.ptr(ws), _lib.ptr(g_rpn_cls), _lib.ptr(g_rpn_box), _lib.ptr(g_cls),
                _lib.ptr(g_box), _lib.stream()), "wssdl_multi_task_loss_backward")
        return (g_rpn_cls, g_rpn_box, g_cls, g_box) + (None,) * 9


def multi_task_loss(rpn_cls_score, rpn_bbox_pred, cls_score, bbox_pred, rpn_data, roi_data, n_box_images=None):
    """-> tensor [4]: rpn_cross_entropy, rpn_loss_box, cross_entropy, loss_box (TERMS).
    rpn_data = (labels, targets, inside_w, outside_w) of the anchor-target layer, roi_data =
    (rois, labels, targets, inside_w, outside_w) of the proposal-target layer."""
    return _MultiTaskLoss.apply(rpn_cls_score, rpn_bbox_pred, cls_score, bbox_pred, rpn_data[0], rpn_data[1],
                                rpn_data[2], rpn_data[3], roi_data[1], roi_data[2], roi_data[3], roi_data[4],
                                n_box_images)
