"""Box <-> delta transforms as device tensor ops.

Same maths as the reference's ``fast_rcnn/bbox_transform.py`` (:10-28 encode, :30-61 decode,
:63-77 clip), written over a (width, height, centre) helper.  On the RPN hot path these
formulas run *inside* the HIP kernels (proposal decode, anchor / RoI targets); the stand-alone
functions serve the post-detection step of the test path (fast_rcnn/test_bus.py), where the
head's class-wise deltas are decoded.  f32 PyTorch ops, +1 pixel convention.
"""
import torch


def _whc(b):
    """(w, h, cx, cy) of [R,4] corner boxes; centre = corner + half extent, as the reference."""
    wh = b[:, 2:4] - b[:, 0:2] + 1.0
    ctr = b[:, 0:2] + 0.5 * wh
    return wh[:, 0], wh[:, 1], ctr[:, 0], ctr[:, 1]


def bbox_transform(ex_rois, gt_rois):
    """Regression targets (dx, dy, dw, dh) that map ex_rois onto gt_rois."""
    ew, eh, ex, ey = _whc(ex_rois)
    gw, gh, gx, gy = _whc(gt_rois)
    return torch.stack(((gx - ex) / ew, (gy - ey) / eh, (gw / ew).log(), (gh / eh).log()), dim=1)


def bbox_transform_inv(boxes, deltas):
    """Decode class-wise deltas [R,4K] on boxes [R,4] -> corner boxes [R,4K]."""
    if boxes.shape[0] == 0:
        return deltas.new_zeros((0, deltas.shape[1]))
    w, h, cx, cy = (t.unsqueeze(1) for t in _whc(boxes.to(deltas.dtype)))
    d = deltas.reshape(deltas.shape[0], -1, 4)                 # [R,K,(dx,dy,dw,dh)]
    pcx = d[..., 0] * w + cx
    pcy = d[..., 1] * h + cy
    half_w = 0.5 * (d[..., 2].exp() * w)
    half_h = 0.5 * (d[..., 3].exp() * h)
    out = torch.stack((pcx - half_w, pcy - half_h, pcx + half_w, pcy + half_h), dim=-1)
    return out.reshape(deltas.shape)


def clip_boxes(boxes, im_shape):
    """Every coordinate into [0, im-1]; im_shape = (height, width).  In place, like the reference."""
    v = boxes.view(boxes.shape[0], -1, 4)
    xmax, ymax = float(im_shape[1]) - 1.0, float(im_shape[0]) - 1.0
    v[..., 0::2].clamp_(min=0.0, max=xmax)
    v[..., 1::2].clamp_(min=0.0, max=ymax)
    return boxes
