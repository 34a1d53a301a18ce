"""Box transforms as device tensor ops (reference: code/lib/fast_rcnn/bbox_transform.py:10-77).

On the RPN hot path these formulas run *inside* the HIP kernels (proposal decode, anchor /
RoI targets).  The stand-alone functions here serve the post-detection step of the test
path (fast_rcnn/test_bus.py), where the head's class-wise deltas are decoded; f32 PyTorch
ops in the reference's operation order."""
import torch


def bbox_transform(ex_rois, gt_rois):
    ew = ex_rois[:, 2] - ex_rois[:, 0] + 1.0
    eh = ex_rois[:, 3] - ex_rois[:, 1] + 1.0
    ecx = ex_rois[:, 0] + 0.5 * ew
    ecy = ex_rois[:, 1] + 0.5 * eh
    gw = gt_rois[:, 2] - gt_rois[:, 0] + 1.0
    gh = gt_rois[:, 3] - gt_rois[:, 1] + 1.0
    gcx = gt_rois[:, 0] + 0.5 * gw
    gcy = gt_rois[:, 1] + 0.5 * gh
    return torch.stack(((gcx - ecx) / ew, (gcy - ecy) / eh, torch.log(gw / ew), torch.log(gh / eh)), dim=1)


def bbox_transform_inv(boxes, deltas):
    """boxes [R,4], deltas [R,4K] -> [R,4K] (bbox_transform.py:30-61)."""
    if boxes.shape[0] == 0:
        return torch.zeros((0, deltas.shape[1]), dtype=deltas.dtype, device=deltas.device)
    boxes = boxes.to(deltas.dtype)
    w = boxes[:, 2] - boxes[:, 0] + 1.0
    h = boxes[:, 3] - boxes[:, 1] + 1.0
    cx = boxes[:, 0] + 0.5 * w
    cy = boxes[:, 1] + 0.5 * h
    dx, dy, dw, dh = deltas[:, 0::4], deltas[:, 1::4], deltas[:, 2::4], deltas[:, 3::4]
    pcx = dx * w[:, None] + cx[:, None]
    pcy = dy * h[:, None] + cy[:, None]
    pw = torch.exp(dw) * w[:, None]
    ph = torch.exp(dh) * h[:, None]
    out = torch.zeros_like(deltas)
    out[:, 0::4] = pcx - 0.5 * pw
    out[:, 1::4] = pcy - 0.5 * ph
    out[:, 2::4] = pcx + 0.5 * pw
    out[:, 3::4] = pcy + 0.5 * ph
    return out


def clip_boxes(boxes, im_shape):
    """bbox_transform.py:63-77: every coordinate into [0, im-1]; im_shape = (h, w)."""
    boxes[:, 0::4] = torch.clamp(boxes[:, 0::4], max=float(im_shape[1]) - 1).clamp_min(0)
    boxes[:, 1::4] = torch.clamp(boxes[:, 1::4], max=float(im_shape[0]) - 1).clamp_min(0)
    boxes[:, 2::4] = torch.clamp(boxes[:, 2::4], max=float(im_shape[1]) - 1).clamp_min(0)
    boxes[:, 3::4] = torch.clamp(boxes[:, 3::4], max=float(im_shape[0]) - 1).clamp_min(0)
    return boxes
