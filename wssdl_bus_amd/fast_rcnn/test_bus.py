"""Inference path: im_detect + per-class NMS.

Reference: code/lib/fast_rcnn/test_bus.py:146-240 (im_detect), :360-401 (per-class NMS with
``utils.cython_nms.nms`` -- the same greedy rule as cpu_nms -- and the max_per_image cap).
The proposal layer, RoI pooling and every NMS here run on the HIP library; decoding the
class-wise deltas is a handful of tensor ops."""
import numpy as np
import torch

from ..nms.hip_nms import hip_nms
from .bbox_transform import bbox_transform_inv
from .config import cfg


def _clip_boxes(boxes, im_shape):
    """test_bus.py:129-139: x1,y1 >= 0; x2 <= w-1; y2 <= h-1 (one-sided, unlike clip_boxes)."""
    boxes[:, 0::4] = boxes[:, 0::4].clamp_min(0)
    boxes[:, 1::4] = boxes[:, 1::4].clamp_min(0)
    boxes[:, 2::4] = boxes[:, 2::4].clamp(max=float(im_shape[1]) - 1)
    boxes[:, 3::4] = boxes[:, 3::4].clamp(max=float(im_shape[0]) - 1)
    return boxes


@torch.no_grad()
def im_detect(net, data, im_info):
    """One image through the test wiring of the network (Resnet_test_bus.py / VGGnet_test_bus.py).
    data [1,H,W,3] NHWC, im_info [1,>=3] (h, w, scale, ...).  Returns (scores [R,K], pred_boxes
    [R,4K]) in original-image coordinates, like the reference."""
    assert data.shape[0] == 1, "Only single-image batch implemented"         # test_bus.py:209
    was_training = net.training
    net.eval()
    try:
        layers = net(data, im_info, None, None, is_training=False, is_ws=False, test_net=True)
    finally:
        net.train(was_training)
    rois = layers['rpn_rois']
    scale = float(im_info[0, 2])
    boxes = rois[:, 1:5] / scale
    scores = layers['cls_prob']
    if cfg.TEST.BBOX_REG:
        pred = bbox_transform_inv(boxes, layers['bbox_pred'])
        pred = _clip_boxes(pred, (float(im_info[0, 0]) / scale, float(im_info[0, 1]) / scale))
    else:
        pred = boxes.repeat(1, scores.shape[1])
    return scores, pred


@torch.no_grad()
def post_detections_device(scores, boxes, num_classes, thresh=0.05, max_per_image=300):
    """The whole post-detection step as one C-ABI call (wssdl_post_detections: one set of launches for all
    classes, no read-back): returns (dets [K-1, R, 5] f32, counts [K-1] i32) on the GPU; rows
    dets[j-1, :counts[j-1]] are class j's detections in descending score order."""
    from .. import _lib
    s = scores.to(torch.float32).contiguous()
    b = boxes.to(torch.float32).contiguous()
    R, K = s.shape
    L = _lib.lib()
    dets = torch.empty((K - 1, max(R, 1), 5), dtype=torch.float32, device=s.device)
    counts = torch.empty((K - 1,), dtype=torch.int32, device=s.device)
    with torch.cuda.device(s.device):
        n = L.wssdl_post_detections_workspace_bytes(R, K)
        ws = torch.empty((n,), dtype=torch.uint8, device=s.device)
        _lib.check(L.wssdl_post_detections(_lib.ptr(s), _lib.ptr(b), R, K, float(thresh), float(cfg.TEST.NMS),
                                           int(max_per_image), _lib.ptr(dets), _lib.ptr(counts), _lib.ptr(ws), n,
                                           _lib.stream()), "wssdl_post_detections")
    return dets, counts


@torch.no_grad()
def postprocess_detections(scores, boxes, num_classes, thresh=0.05, max_per_image=300):
    """test_bus.py:360-401: per class j >= 1 keep scores > thresh, NMS at cfg.TEST.NMS, then cap
    the image at max_per_image detections over all classes.  Returns {j: dets [n,5]} (GPU).
    On the GPU this is one device op (post_detections_device) followed by ONE read-back of the per-class
    counts; the class-agnostic variant and CPU tensors take the step-by-step form below."""
    if scores.is_cuda and not cfg.TEST.CLS_AGNOSTIC_NMS and scores.shape[1] == num_classes and num_classes <= 65 \
            and cfg.TEST.get("FUSED_POST_DETECTIONS", True):
        dets, counts = post_detections_device(scores, boxes, num_classes, thresh, max_per_image)
        n = counts.cpu().tolist()
        return {j: dets[j - 1, :n[j - 1]] for j in range(1, num_classes)}
    out = {}
    for j in range(1, num_classes):
        inds = torch.nonzero(scores[:, j] > thresh).reshape(-1)
        dets = torch.cat((boxes[inds, j * 4:(j + 1) * 4], scores[inds, j:j + 1]), dim=1).to(torch.float32)
        if dets.shape[0]:
            keep = hip_nms(dets.contiguous(), cfg.TEST.NMS)
            dets = dets[keep]
        out[j] = dets
    if cfg.TEST.CLS_AGNOSTIC_NMS:
        alld = torch.cat([torch.cat((out[j], torch.full((out[j].shape[0], 1), float(j), device=scores.device)), 1)
                          for j in range(1, num_classes)], 0)
        if alld.shape[0]:
            # the reference hands all six columns to nms(); column 4 is the score either way
            keep = hip_nms(alld[:, :5].contiguous(), cfg.TEST.NMS)
            alld = alld[keep]
        for j in range(1, num_classes):
            out[j] = alld[alld[:, 5] == j][:, :5]
    if max_per_image > 0:
        image_scores = torch.cat([out[j][:, 4] for j in range(1, num_classes)])
        if image_scores.numel() > max_per_image:
            image_thresh = torch.sort(image_scores).values[-max_per_image]
            for j in range(1, num_classes):
                out[j] = out[j][out[j][:, 4] >= image_thresh]
    return out
