"""anchor_target_layer / anchor_target_layer_ws / anchor_target_layer_joint on the GPU.

Reference: code/lib/rpn_msr/anchor_target_layer_tf_bus.py:19-303, :306-325, :328-628.
Same call signatures and output layouts:
  rpn_labels               [N, 1, A*H, W] f32 in {-1, 0, 1}
  rpn_bbox_targets         [N, 4A, H, W]  f32
  rpn_bbox_inside_weights  [N, 4A, H, W]  f32
  rpn_bbox_outside_weights [N, 4A, H, W]  f32

The label assignment (inside filter, f64 IoU, thresholds, per-gt arg-max ties) and
the target / weight blobs are HIP kernels.  The two random sub-samplings
(:202-217) follow ``cfg.SAMPLING_RNG``:
  'reference' -- drawn on the host from ``numpy.random`` (the reference's global
                 legacy stream: same seed + same call order => bit-identical
                 labels); costs one device->host copy of the int8 labels;
  'device'    -- exact-k random subset on the GPU, no host round trip.
"""
import numpy as np
import numpy.random as npr
import torch

from .. import _lib
from ..fast_rcnn.config import cfg
from .generate_anchors import generate_anchors

DEBUG = False

_DATASETS = {"SNUBH": _lib.DATASET_SNUBH, "SNUBH_FG": _lib.DATASET_SNUBH_FG}
_device_calls = [0]


def _shape_hw(rpn_cls_score):
    shp = rpn_cls_score.shape if hasattr(rpn_cls_score, "shape") else tuple(rpn_cls_score)
    return int(shp[0]), int(shp[1]), int(shp[2])


def _device_of(*xs):
    for x in xs:
        if isinstance(x, torch.Tensor) and x.is_cuda:
            return x.device
    return torch.device("cuda", torch.cuda.current_device())


def _subsample_reference(labels_pre, rng):
    """anchor_target_layer_tf_bus.py:202-217 on a host copy of the labels, consuming
    `rng` (numpy.random by default) exactly like the reference."""
    lab = labels_pre.cpu().numpy().copy()
    num_fg_max = int(cfg.TRAIN.RPN_FG_FRACTION * cfg.TRAIN.RPN_BATCHSIZE)
    for i in range(lab.shape[0]):
        l = lab[i]
        fg_inds = np.where(l == 1)[0]
        if len(fg_inds) > num_fg_max:
            l[rng.choice(fg_inds, size=(len(fg_inds) - num_fg_max), replace=False)] = -1
        num_bg = cfg.TRAIN.RPN_BATCHSIZE - np.sum(l == 1)
        bg_inds = np.where(l == 0)[0]
        if len(bg_inds) > num_bg:
            l[rng.choice(bg_inds, size=(len(bg_inds) - num_bg), replace=False)] = -1
    return torch.from_numpy(lab).to(labels_pre.device)


def anchor_labels(gt_boxes, num_gt_boxes, im_info, n_images, height, width, _feat_stride,
                  anchor_scales, dataset, device=None):
    """Stage 1: labels before sub-sampling.  Returns (labels_pre [n,K*A] i8,
    argmax_gt [n,K*A] i32, counts [n,4] i32 = (#inside,#fg,#bg,0), gt, anchors)."""
    dev = device if device is not None else _device_of(gt_boxes, im_info)
    gt = _lib.to_device(gt_boxes, torch.float32, dev)
    ng = _lib.to_device(num_gt_boxes, torch.int32, dev)
    info = _lib.to_device(im_info, torch.float32, dev)
    if gt.dim() != 3 or gt.shape[2] != 5:
        raise ValueError("gt_boxes must be [N, MAX_GT, 5]")
    if gt.shape[1] > _lib.MAX_GT:
        raise ValueError("at most %d gt boxes per image are supported" % _lib.MAX_GT)
    if min(gt.shape[0], ng.shape[0], info.shape[0]) < n_images:
        raise ValueError("need gt_boxes / num_gt_boxes / im_info rows for %d images" % n_images)
    anchors = generate_anchors(scales=np.array(anchor_scales))
    A = anchors.shape[0]
    total = height * width * A
    stride = int(np.asarray(_feat_stride).ravel()[0])
    L = _lib.lib()
    with torch.cuda.device(dev):
        labels = torch.empty((n_images, total), dtype=torch.int8, device=dev)
        argmax = torch.empty((n_images, total), dtype=torch.int32, device=dev)
        counts = torch.empty((n_images, 4), dtype=torch.int32, device=dev)
        ws = torch.empty((L.wssdl_anchor_workspace_bytes(n_images),), dtype=torch.uint8, device=dev)
        _lib.check(L.wssdl_anchor_labels(
            _lib.ptr(gt), gt.shape[1], _lib.ptr(ng), _lib.ptr(info), info.shape[1], n_images,
            height, width, _lib.host_ptr(anchors), A, stride,
            _DATASETS.get(str(dataset), _lib.DATASET_FG_ONLY),
            float(cfg.TRAIN.RPN_POSITIVE_OVERLAP), float(cfg.TRAIN.RPN_NEGATIVE_OVERLAP),
            int(bool(cfg.TRAIN.RPN_CLOBBER_POSITIVES)), _lib.ptr(labels), _lib.ptr(argmax),
            _lib.ptr(counts), _lib.ptr(ws), ws.numel(), _lib.stream()), "wssdl_anchor_labels")
    return labels, argmax, counts, gt, anchors


def anchor_targets(labels, argmax, gt, anchors, n_images, n_out, height, width, _feat_stride,
                   device):
    """Stage 3: final labels -> the four output blobs for n_out images (images
    n_images..n_out-1 are all-ignore)."""
    A = anchors.shape[0]
    stride = int(np.asarray(_feat_stride).ravel()[0])
    iw = np.ascontiguousarray(cfg.TRAIN.RPN_BBOX_INSIDE_WEIGHTS, dtype=np.float32)
    with torch.cuda.device(device):
        rpn_labels = torch.empty((n_out, 1, A * height, width), dtype=torch.float32, device=device)
        tg = torch.empty((n_out, 4 * A, height, width), dtype=torch.float32, device=device)
        inw = torch.empty_like(tg)
        outw = torch.empty_like(tg)
        _lib.check(_lib.lib().wssdl_anchor_targets(
            _lib.ptr(labels), _lib.ptr(argmax), _lib.ptr(gt), gt.shape[1] if gt is not None else 1,
            n_images, n_out, height, width, _lib.host_ptr(anchors), A, stride, _lib.host_ptr(iw),
            float(cfg.TRAIN.RPN_POSITIVE_WEIGHT), _lib.ptr(rpn_labels), _lib.ptr(tg),
            _lib.ptr(inw), _lib.ptr(outw), _lib.stream()), "wssdl_anchor_targets")
    return rpn_labels, tg, inw, outw


def _run(rpn_cls_score, gt_boxes, num_gt_boxes, im_info, n_images, n_out, _feat_stride,
         anchor_scales, dataset, rng):
    as_np = _lib.wants_numpy(gt_boxes, im_info, rpn_cls_score if hasattr(rpn_cls_score, "dtype") else None)
    _, height, width = _shape_hw(rpn_cls_score)
    dev = _device_of(rpn_cls_score, gt_boxes, im_info)
    with _lib.timed("anchor_target_layer", dict(n_images=n_images, n_out=n_out, H=height, W=width)):
        outs = _run_device(gt_boxes, num_gt_boxes, im_info, n_images, n_out, height, width,
                           _feat_stride, anchor_scales, dataset, rng, dev)
    if as_np:
        return tuple(o.cpu().numpy() for o in outs)
    return outs


def _run_device(gt_boxes, num_gt_boxes, im_info, n_images, n_out, height, width, _feat_stride,
                anchor_scales, dataset, rng, dev):
    if n_images > 0:
        labels, argmax, counts, gt, anchors = anchor_labels(
            gt_boxes, num_gt_boxes, im_info, n_images, height, width, _feat_stride, anchor_scales,
            dataset, dev)
        if cfg.SAMPLING_RNG == "reference":
            labels = _subsample_reference(labels, npr if rng is None else rng)
        else:
            _device_calls[0] += 1
            seed = (int(cfg.DEVICE_RNG_SEED) * 0x9E3779B1 + _device_calls[0]) & 0xFFFFFFFFFFFFFFFF
            _lib.check(_lib.lib().wssdl_anchor_subsample_device(
                _lib.ptr(labels), n_images, labels.shape[1], int(cfg.TRAIN.RPN_BATCHSIZE),
                float(cfg.TRAIN.RPN_FG_FRACTION), seed, _lib.ptr(counts), _lib.stream()),
                "wssdl_anchor_subsample_device")
    else:
        labels = argmax = gt = None
        anchors = generate_anchors(scales=np.array(anchor_scales))
    return anchor_targets(labels, argmax, gt, anchors, n_images, n_out, height, width,
                          _feat_stride, dev)


def anchor_target_layer(rpn_cls_score, gt_boxes, num_gt_boxes, im_info, data,
                        _feat_stride=[16, ], anchor_scales=[4, 8, 16, 32], dataset='SNUBH',
                        rng=None):
    """Assign anchors to ground-truth targets for every image of the batch
    (alternating mode, anchor_target_layer_tf_bus.py:19-303).  `data` is unused, as
    in the reference; `rpn_cls_score` only supplies the (N, H, W) shape."""
    n, _, _ = _shape_hw(rpn_cls_score)
    return _run(rpn_cls_score, gt_boxes, num_gt_boxes, im_info, n, n, _feat_stride, anchor_scales,
                dataset, rng)


def anchor_target_layer_ws(rpn_cls_score, gt_boxes, num_gt_boxes, im_info, data,
                           _feat_stride=[16, ], anchor_scales=[4, 8, 16, 32]):
    """Weakly supervised batch: every label -1, all targets / weights zero (:306-325)."""
    n, _, _ = _shape_hw(rpn_cls_score)
    return _run(rpn_cls_score, None, None, im_info, 0, n, _feat_stride, anchor_scales, None, None)


def anchor_target_layer_joint(rpn_cls_score, gt_boxes, num_gt_boxes, im_info, data, is_training,
                              _feat_stride=[16, ], anchor_scales=[4, 8, 16, 32], dataset='SNUBH',
                              rng=None):
    """Combined mini-batch (:328-628): the first cfg.TRAIN.IMS_PER_BATCH images are
    supervised; when training, cfg.TRAIN.WS_IMS_PER_BATCH all-ignore images follow."""
    n_s = int(cfg.TRAIN.IMS_PER_BATCH)
    n_out = n_s + (int(cfg.TRAIN.WS_IMS_PER_BATCH) if is_training else 0)
    return _run(rpn_cls_score, gt_boxes, num_gt_boxes, im_info, n_s, n_out, _feat_stride,
                anchor_scales, dataset, rng)
