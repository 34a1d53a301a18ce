"""proposal_target_layer / proposal_target_layer_joint on the GPU.

Reference: code/lib/rpn_msr/proposal_target_layer_tf_bus.py:15-97, :99-184,
_sample_rois :228-280, _sample_rois_ws :282-295.  Same call signatures; returns
``(rois [.,5], labels [.,1], bbox_targets [.,4K], bbox_inside_weights, bbox_outside_weights)``.

Device work: f64 IoU of every candidate RoI (proposals + appended gt boxes) against
its image's positive gt boxes with max / arg-max (``wssdl_roi_gt_assign``), then
gather + labels + bbox_transform + class expansion for the sampled rows
(``wssdl_roi_targets``).  The fg/bg sampling between the two consumes
``numpy.random`` exactly like the reference (same seed and call order => same rows);
with ``cfg.SAMPLING_RNG = 'device'`` it is ``wssdl_roi_sample_device`` instead (same
distribution, counter-based device RNG) and the layer needs no host copy at all: its output
has the fixed shape S * 128 rows; an image that runs short of candidates (the reference then
returns fewer rows) leaves padding rows (-1,0,0,0,0) with label -1 and zero weights.
"""
import numpy as np
import numpy.random as npr
import torch

from .. import _lib
from ..fast_rcnn.config import cfg

DEBUG = False
_device_calls = [0]


def tag_counts(rois, counts):
    """Remember the per-image row counts of a proposal blob (rows are grouped by image,
    ascending) so that the layers below need not read the batch column back."""
    try:
        rois._wssdl_counts = tuple(int(c) for c in counts)
    except Exception:
        pass
    return rois


def _counts_hint(rois, n_images):
    c = getattr(rois, "_wssdl_counts", None)
    if c is None or len(c) < n_images or sum(c) != rois.shape[0]:
        return None
    return c


def _host_gt(gt_boxes, num_gt_boxes):
    g = gt_boxes.detach().cpu().numpy() if isinstance(gt_boxes, torch.Tensor) else np.asarray(gt_boxes)
    n = num_gt_boxes.detach().cpu().numpy() if isinstance(num_gt_boxes, torch.Tensor) else np.asarray(num_gt_boxes)
    return g.astype(np.float32, copy=False), n.astype(np.int64, copy=False)


def _num_pos(gt_host, ng_host, n_images):
    # b_pos / num_pos, :40-42: count of rows with class != 0 (positives come first)
    return np.array([int(np.sum(gt_host[i, :ng_host[i], 4] != 0)) for i in range(n_images)],
                    dtype=np.int32)


def _empty_outputs(dev, num_classes):
    z = lambda w: torch.zeros((0, w), dtype=torch.float32, device=dev)
    return [z(5), z(1), z(4 * num_classes), z(4 * num_classes), z(4 * num_classes)]


def _normalize_arg():
    """cfg.TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED (proposal_target_layer_tf_bus.py:221-224): the 8 host doubles
    wssdl_roi_targets takes (means then stds), or None when the switch is off (the reference's default)."""
    if not cfg.TRAIN.get("BBOX_NORMALIZE_TARGETS_PRECOMPUTED", False):
        return None
    v = np.concatenate([np.asarray(cfg.TRAIN.BBOX_NORMALIZE_MEANS, dtype=np.float64).reshape(4),
                        np.asarray(cfg.TRAIN.BBOX_NORMALIZE_STDS, dtype=np.float64).reshape(4)])
    return np.ascontiguousarray(v)


def _supervised(rois, gt_dev, gt_host, ng_host, images, append_gt, num_classes, rng):
    """Sampled rows for the supervised `images` (in that order), all on the GPU."""
    with _lib.timed("proposal_target_layer", dict(R=int(rois.shape[0]), images=len(images))):
        return _supervised_impl(rois, gt_dev, gt_host, ng_host, images, append_gt, num_classes, rng)


def _supervised_device(rois, gt_dev, ng_dev, images, append_gt, num_classes):
    """cfg.SAMPLING_RNG == 'device': everything stays on the GPU."""
    dev = rois.device
    n_img, max_gt = gt_dev.shape[0], gt_dev.shape[1]
    L = _lib.lib()
    S = len(images)
    rpi = int(cfg.TRAIN.BATCH_SIZE) // 1
    fg_rpi = int(np.round(cfg.TRAIN.FG_FRACTION * rpi))
    iw = np.ascontiguousarray(cfg.TRAIN.BBOX_INSIDE_WEIGHTS, dtype=np.float32)
    norm = _normalize_arg()
    with torch.cuda.device(dev):
        images_dev = _images_tensor(images, dev)
        R = int(rois.shape[0])
        # one C call: candidates (every gt slot of the supervised images is appended, :44-50; slots
        # past the image's positives carry batch index -1 and can never be drawn) -> assignment ->
        # fg / bg draw -> rows and targets.  Fixed shape, no read-back: an image that runs short of
        # candidates leaves rows (-1,0,0,0,0) with label -1 and zero weights, which RoI pooling, the
        # losses and the MIL selection all ignore
        _device_calls[0] += 1
        seed = (int(cfg.DEVICE_RNG_SEED) * 0x9E3779B1 + 0x51ED27 * _device_calls[0]) & 0xFFFFFFFFFFFFFFFF
        n_keep = S * rpi
        nws = L.wssdl_proposal_target_device_workspace_bytes(R, n_img, max_gt, S, rpi, int(bool(append_gt)))
        ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
        out_rois = torch.empty((n_keep, 5), dtype=torch.float32, device=dev)
        labels = torch.empty((n_keep, 1), dtype=torch.float32, device=dev)
        tgs = torch.empty((3, n_keep, 4 * num_classes), dtype=torch.float32, device=dev)
        tg, inw, outw = tgs[0], tgs[1], tgs[2]
        # (the event pair brackets the launches only: with the allocations inside it, it timed the host)
        with _lib.timed("proposal_target_layer", dict(R=R, images=S)):
            _lib.check(L.wssdl_proposal_target_device(
                _lib.ptr(rois), R, _lib.ptr(gt_dev), max_gt, _lib.ptr(ng_dev), n_img, _lib.ptr(images_dev), S,
                int(bool(append_gt)), rpi, fg_rpi, float(cfg.TRAIN.FG_THRESH), float(cfg.TRAIN.BG_THRESH_HI),
                float(cfg.TRAIN.BG_THRESH_LO), seed, int(num_classes), _lib.host_ptr(iw),
                _lib.host_ptr(norm) if norm is not None else None, _lib.ptr(out_rois),
                _lib.ptr(labels), _lib.ptr(tg), _lib.ptr(inw), _lib.ptr(outw), _lib.ptr(ws), nws, _lib.stream()),
                "wssdl_proposal_target_device")
    return [out_rois, labels, tg, inw, outw]


_images_cache = {}


def _images_tensor(images, dev):
    key = (tuple(images), str(dev))
    t = _images_cache.get(key)
    if t is None:
        t = torch.as_tensor(list(images), dtype=torch.int32, device=dev)
        _images_cache[key] = t
    return t


def _use_device_rng(rng):
    return cfg.SAMPLING_RNG == "device" and rng is None


def _supervised_impl(rois, gt_dev, gt_host, ng_host, images, append_gt, num_classes, rng):
    dev = rois.device
    n_img = gt_dev.shape[0]
    num_pos = _num_pos(gt_host, ng_host, n_img)
    R = rois.shape[0]
    extra = []
    if append_gt:                                                  # :44-50
        for i in images:
            p = int(num_pos[i])
            if p:
                e = np.empty((p, 5), dtype=np.float32)
                e[:, 0] = i
                e[:, 1:] = gt_host[i, :p, :4]
                extra.append(e)
    if extra:
        cand = torch.cat([rois, torch.from_numpy(np.concatenate(extra)).to(dev)], dim=0).contiguous()
    else:
        cand = rois
    Rc = cand.shape[0]
    L = _lib.lib()
    with torch.cuda.device(dev):
        num_pos_dev = torch.from_numpy(num_pos).to(dev)
        max_ov = torch.empty((Rc,), dtype=torch.float64, device=dev)
        assign = torch.empty((Rc,), dtype=torch.int32, device=dev)
        _lib.check(L.wssdl_roi_gt_assign(_lib.ptr(cand), Rc, _lib.ptr(gt_dev), gt_dev.shape[1],
                                         _lib.ptr(num_pos_dev), n_img, _lib.ptr(max_ov),
                                         _lib.ptr(assign), _lib.stream()), "wssdl_roi_gt_assign")
        batch_host = cand[:, 0].cpu().numpy()
        ov_host = max_ov.cpu().numpy()
    rois_per_image = int(cfg.TRAIN.BATCH_SIZE) // 1                # :56-57 (py2 int division)
    fg_rois_per_image = int(np.round(cfg.TRAIN.FG_FRACTION * rois_per_image))
    keep_all, fg_all = [], []
    for i in images:
        if num_pos[i] == 0:
            raise ValueError("image %d has no positive gt box (the reference's argmax over an "
                             "empty axis raises here too)" % i)
        # candidates of image i in the reference's order: its proposals, then its gt rows
        idx = np.where(batch_host == i)[0]
        mo = ov_host[idx]
        fg_inds = np.where(mo >= cfg.TRAIN.FG_THRESH)[0]            # :241
        n_fg = min(fg_rois_per_image, fg_inds.size)
        if fg_inds.size > 0:
            fg_inds = rng.choice(fg_inds, size=n_fg, replace=False)
        bg_inds = np.where((mo < cfg.TRAIN.BG_THRESH_HI) & (mo >= cfg.TRAIN.BG_THRESH_LO))[0]
        n_bg = min(rois_per_image - n_fg, bg_inds.size)
        if bg_inds.size > 0:
            bg_inds = rng.choice(bg_inds, size=n_bg, replace=False)
        keep = np.append(fg_inds, bg_inds).astype(np.int64)
        keep_all.append(idx[keep])
        f = np.zeros(keep.size, dtype=np.uint8)
        f[:n_fg] = 1                                               # labels[fg_rois_per_this_image:] = 0
        fg_all.append(f)
    keep_np = np.concatenate(keep_all).astype(np.int32) if keep_all else np.zeros(0, np.int32)
    fg_np = np.concatenate(fg_all) if fg_all else np.zeros(0, np.uint8)
    n_keep = keep_np.size
    iw = np.ascontiguousarray(cfg.TRAIN.BBOX_INSIDE_WEIGHTS, dtype=np.float32)
    norm = _normalize_arg()
    with torch.cuda.device(dev):
        keep_dev = torch.from_numpy(keep_np).to(dev)
        fg_dev = torch.from_numpy(fg_np).to(dev)
        out_rois = torch.empty((n_keep, 5), dtype=torch.float32, device=dev)
        labels = torch.empty((n_keep, 1), dtype=torch.float32, device=dev)
        tg = torch.empty((n_keep, 4 * num_classes), dtype=torch.float32, device=dev)
        inw = torch.empty_like(tg)
        outw = torch.empty_like(tg)
        _lib.check(L.wssdl_roi_targets(
            _lib.ptr(cand), _lib.ptr(keep_dev), _lib.ptr(fg_dev), n_keep, _lib.ptr(assign),
            _lib.ptr(gt_dev), gt_dev.shape[1], int(num_classes), _lib.host_ptr(iw),
            _lib.host_ptr(norm) if norm is not None else None,
            _lib.ptr(out_rois), _lib.ptr(labels), _lib.ptr(tg), _lib.ptr(inw), _lib.ptr(outw),
            _lib.stream()), "wssdl_roi_targets")
    return [out_rois, labels, tg, inw, outw], batch_host[:R]


def _rois_of_images_hinted(rois, counts, images):
    """Same selection when the rows are known to be grouped by image (counts hint)."""
    off = np.concatenate([[0], np.cumsum(counts)])
    images = list(images)
    if images and images == list(range(images[0], images[-1] + 1)):
        return rois[int(off[images[0]]):int(off[images[-1] + 1])]
    parts = [rois[int(off[i]):int(off[i + 1])] for i in images]
    return torch.cat(parts, dim=0) if parts else rois[:0]


def _rois_of_images(rois, batch_host, images):
    """rpn_rois[rpn_rois[:,0]==i] for each i in order, concatenated (on the GPU)."""
    idx = [np.where(batch_host == i)[0] for i in images]
    idx = np.concatenate(idx) if idx else np.zeros(0, np.int64)
    if idx.size == rois.shape[0] and np.array_equal(idx, np.arange(idx.size)):
        return rois
    return rois.index_select(0, torch.from_numpy(idx).to(rois.device))


def _finish(outs, as_np):
    if as_np:
        return tuple(o.cpu().numpy() for o in outs)
    return tuple(outs)


def _weak_rois(rpn_rois, rois, images):
    """Rows of the weak `images`, in order (hint from the proposal layer, else read-back)."""
    images = list(images)
    pitch = getattr(rpn_rois, "_wssdl_pitch", None)
    if pitch is not None and images:
        # padded blob (cfg.PADDED_ROIS): image i owns rows [i*pitch, (i+1)*pitch), the unused ones
        # carry batch index -1 -- a fixed-shape slice, no read-back
        if images == list(range(images[0], images[-1] + 1)):
            return rois[images[0] * pitch:(images[-1] + 1) * pitch]
        return torch.cat([rois[i * pitch:(i + 1) * pitch] for i in images], dim=0)
    hint = _counts_hint(rpn_rois, images[-1] + 1) if (images and isinstance(rpn_rois, torch.Tensor)) else None
    if hint is not None:
        return _rois_of_images_hinted(rois, hint, images)
    return _rois_of_images(rois, rois[:, 0].cpu().numpy(), images)


def proposal_target_layer(rpn_rois, gt_boxes, num_gt_boxes, _num_classes, is_training, is_ws,
                          rng=None):
    """Alternating mode (:15-97).  Weak batches (is_training and is_ws) return every
    RoI with zero labels / targets (:282-295)."""
    as_np = _lib.wants_numpy(rpn_rois)
    device_rng = _use_device_rng(rng)
    rng = npr if rng is None else rng
    rois = _lib.to_device(rpn_rois, torch.float32)
    dev = rois.device
    n_img = int(gt_boxes.shape[0])
    K = int(_num_classes)
    if is_training and is_ws:
        r = _weak_rois(rpn_rois, rois, range(n_img))
        n = r.shape[0]
        z = lambda w: torch.zeros((n, w), dtype=torch.float32, device=dev)
        return _finish([r, z(1), z(4 * K), z(4 * K), z(4 * K)], as_np)
    gt_dev = _lib.to_device(gt_boxes, torch.float32, dev)
    append_gt = bool(is_training) and not bool(is_ws)
    if device_rng:
        outs = _supervised_device(rois, gt_dev, _lib.to_device(num_gt_boxes, torch.int32, dev),
                                  list(range(n_img)), append_gt, K)
        return _finish(outs, as_np)
    gt_host, ng_host = _host_gt(gt_boxes, num_gt_boxes)
    outs, _ = _supervised(rois, gt_dev, gt_host, ng_host, list(range(n_img)), append_gt, K, rng)
    return _finish(outs, as_np)


def proposal_target_layer_joint(rpn_rois, gt_boxes, num_gt_boxes, _num_classes, is_training,
                                rng=None):
    """Combined mode (:99-184): sampled rows for the cfg.TRAIN.IMS_PER_BATCH supervised
    images; when training, the *rois* of the cfg.TRAIN.WS_IMS_PER_BATCH weak images
    are appended to the rois output only (:162-182)."""
    as_np = _lib.wants_numpy(rpn_rois)
    device_rng = _use_device_rng(rng)
    rng = npr if rng is None else rng
    rois = _lib.to_device(rpn_rois, torch.float32)
    dev = rois.device
    gt_dev = _lib.to_device(gt_boxes, torch.float32, dev)
    n_s = int(cfg.TRAIN.IMS_PER_BATCH)
    ws_images = range(n_s, n_s + int(cfg.TRAIN.WS_IMS_PER_BATCH))
    if device_rng:
        outs = _supervised_device(rois, gt_dev, _lib.to_device(num_gt_boxes, torch.int32, dev),
                                  list(range(n_s)), bool(is_training), int(_num_classes))
        ws = _weak_rois(rpn_rois, rois, ws_images) if is_training else None
    else:
        gt_host, ng_host = _host_gt(gt_boxes, num_gt_boxes)
        outs, batch_host = _supervised(rois, gt_dev, gt_host, ng_host, list(range(n_s)),
                                       bool(is_training), int(_num_classes), rng)
        ws = _rois_of_images(rois, batch_host, ws_images) if is_training else None
    if is_training:
        outs[0] = torch.cat([outs[0], ws.reshape(-1, 5)], dim=0)
    return _finish(outs, as_np)
