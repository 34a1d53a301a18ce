"""Blob helper functions on the GPU (the pinnable half of the reference's host image path).

Reference: code/lib/utils/blob.py:19-32 (im_list_to_blob), :34-79 (prep_im_for_blob incl. the
cropping of weak images :43-48), roi_data_layer/minibatch_bus.py:259-318 (_get_image_blob,
_get_image_blob_joint: grey plane stacked x3, flip, supervised images first then weak ones),
datasets/imdb.py:106-121 (box mirroring).  Same function names and argument meaning; images live
on the device.

``skimage.transform.resize`` (blob.py:74-77) and ``skimage.transform.rotate`` (:39-41) run on the device
too (round 4): ``wssdl_image_resize`` / ``wssdl_image_warp`` follow the published algorithm of the release
the reference's README pins (scikit-image 0.14.2: order 1, mode 'constant', cval 0 -- the borders of an
up-scaled image fade towards 0 --, clip to the input's range, no anti-aliasing).  The library is absent
here, so this step is checked against a restatement of that algorithm only ("parity unpinned",
oracle/np_oracle.py); everything around it is pinned by fixtures from the reference's own blob.py.
prep_im_for_blob still takes the resize as a callable (``resize=``) for a caller that wants another filter.
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..fast_rcnn.config import cfg

PIXEL_MEANS = np.array([[[68.274, 68.274, 68.274]]])     # fast_rcnn/config.py:284
PIXEL_STDS = np.array([[[52.802, 52.802, 52.802]]])      # fast_rcnn/config.py:287


def _ws(dev):
    n = _lib.lib().wssdl_image_prep_workspace_bytes()
    return torch.empty((n,), dtype=torch.uint8, device=dev), n


def prep_im_pre_resize(gray, flipped=False, brightness_delta=None, contrast_factor=None,
                       pixel_means=PIXEL_MEANS, crop=None):
    """gray [h,w] u8 (GPU tensor or numpy) -> f32 [h',w',3] GPU tensor: the array the reference hands
    to skimage.transform.resize (blob.py:34-60 after minibatch_bus.py:269-272).  `brightness_delta`
    / `contrast_factor`: the values of the reference's two np.random.uniform draws, None = off.
    `crop` = (offsets_u, offsets_d, offsets_l, offsets_r) of a weak image (blob.py:43-47): the
    reference slices the ALREADY FLIPPED image [u:-d, l:-r]; here that is a strided view of the plane
    (columns [r, w-l) of the unflipped plane when it is flipped) -- no copy, the kernel reads it with
    its row stride and mirrors inside the view."""
    g = _lib.to_device(gray, torch.uint8) if not (isinstance(gray, torch.Tensor) and gray.is_cuda
                                                   and gray.dtype == torch.uint8) else gray
    if g.dim() != 2:
        raise ValueError("expected one grey plane [h, w]")
    if crop is not None:
        u, d, l, r = (int(v) for v in crop)
        if d < 1 or r < 1 or u < 0 or l < 0 or u + d >= g.shape[0] or l + r >= g.shape[1]:
            raise ValueError("crop offsets out of range")
        H0, W0 = g.shape
        g = g[u:H0 - d, r:W0 - l] if flipped else g[u:H0 - d, l:W0 - r]
    if g.stride(1) != 1:
        g = g.contiguous()
    h, w = g.shape
    out = torch.empty((h, w, 3), dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        ws, n = _ws(g.device)
        _lib.check(_lib.lib().wssdl_image_prep(
            _lib.ptr(g), h, w, g.stride(0), int(bool(flipped)),
            int(brightness_delta is not None), float(brightness_delta or 0.0),
            int(contrast_factor is not None), float(contrast_factor if contrast_factor is not None else 1.0),
            float(np.asarray(pixel_means).reshape(-1)[0]), _lib.ptr(out), _lib.ptr(ws), n, _lib.stream()),
            "wssdl_image_prep")
    return out


def skimage_resize_matrix(in_shape, out_shape):
    """The inverse map of skimage.transform.resize (0.14.2, _warps.py): output (col, row, 1) -> input
    (col_scale * (col + 0.5) - 0.5, row_scale * (row + 0.5) - 0.5); a translation to the centre for 1 x 1."""
    rows, cols = int(out_shape[0]), int(out_shape[1])
    h, w = int(in_shape[0]), int(in_shape[1])
    if rows == 1 and cols == 1:
        return np.array([[1.0, 0.0, w / 2.0 - 0.5], [0.0, 1.0, h / 2.0 - 0.5], [0.0, 0.0, 1.0]])
    rs, cs = float(h) / rows, float(w) / cols
    return np.array([[cs, 0.0, cs * 0.5 - 0.5], [0.0, rs, rs * 0.5 - 0.5], [0.0, 0.0, 1.0]])


def skimage_rotate_matrix(in_shape, angle_deg):
    """skimage.transform.rotate (resize=False): T(centre) . R(angle) . T(-centre), centre = (cols, rows)/2 - 0.5."""
    rows, cols = int(in_shape[0]), int(in_shape[1])
    centre = np.array((cols, rows)) / 2. - 0.5
    a = np.deg2rad(angle_deg)
    t1 = np.array([[1.0, 0.0, centre[0]], [0.0, 1.0, centre[1]], [0.0, 0.0, 1.0]])
    t2 = np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])
    t3 = np.array([[1.0, 0.0, -centre[0]], [0.0, 1.0, -centre[1]], [0.0, 0.0, 1.0]])
    return t1.dot(t2.dot(t3))


def skimage_warp(im, matrix, shape, mode="constant", cval=0.0, clip=True):
    """skimage.transform.warp(im, matrix, output_shape=shape, order=1, mode, cval, clip) on the device:
    im [h,w] or [h,w,C] f32 / f64 (GPU tensor or numpy) -> f64 GPU tensor [rows, cols(, C)]."""
    t = _lib.to_device(im, im.dtype if isinstance(im, torch.Tensor) else
                       (torch.float64 if np.asarray(im).dtype == np.float64 else torch.float32))
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    squeeze = t.dim() == 2
    t = (t.unsqueeze(2) if squeeze else t).contiguous()
    h, w, C = (int(v) for v in t.shape)
    rows, cols = int(shape[0]), int(shape[1])
    M = (ctypes.c_double * 9)(*[float(v) for v in np.asarray(matrix, dtype=np.float64).reshape(-1)])
    out = torch.empty((rows, cols, C), dtype=torch.float64, device=t.device)
    L = _lib.lib()
    with torch.cuda.device(t.device):
        n = L.wssdl_image_warp_workspace_bytes()
        ws = torch.empty((n,), dtype=torch.uint8, device=t.device)
        _lib.check(L.wssdl_image_warp(_lib.ptr(t), int(t.dtype == torch.float64), h, w, C, M, rows, cols,
                                      {"constant": 0, "edge": 1}[mode], float(cval), int(bool(clip)), _lib.ptr(out),
                                      _lib.ptr(ws), n, _lib.stream()), "wssdl_image_warp")
    return out[:, :, 0] if squeeze else out


def skimage_resize(im, shape):
    """skimage.transform.resize(im, shape) with the 0.14.2 defaults (blob.py:74-77), on the device."""
    t = _lib.to_device(im, im.dtype if isinstance(im, torch.Tensor) else
                       (torch.float64 if np.asarray(im).dtype == np.float64 else torch.float32))
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    squeeze = t.dim() == 2
    t = (t.unsqueeze(2) if squeeze else t).contiguous()
    h, w, C = (int(v) for v in t.shape)
    rows, cols = int(shape[0]), int(shape[1])
    out = torch.empty((rows, cols, C), dtype=torch.float64, device=t.device)
    L = _lib.lib()
    with torch.cuda.device(t.device):
        n = L.wssdl_image_warp_workspace_bytes()
        ws = torch.empty((n,), dtype=torch.uint8, device=t.device)
        _lib.check(L.wssdl_image_resize(_lib.ptr(t), int(t.dtype == torch.float64), h, w, C, rows, cols,
                                        _lib.ptr(out), _lib.ptr(ws), n, _lib.stream()), "wssdl_image_resize")
    return out[:, :, 0] if squeeze else out


def skimage_rotate(im, angle_deg, cval=0.0):
    """skimage.transform.rotate(im, angle, cval=cval) (blob.py:39-41: order 1, mode 'constant', clip), on the device."""
    shape = tuple(int(v) for v in im.shape)
    return skimage_warp(im, skimage_rotate_matrix(shape, angle_deg), shape[:2], cval=cval)


def prep_im_pre_resize_rotated(gray, flipped, angle_deg, brightness_delta=None, contrast_factor=None,
                               pixel_means=PIXEL_MEANS, crop=None):
    """blob.py:36-60 for a weak image with cfg.TRAIN.USE_ROTATION: /255 in f32, rotate by `angle_deg` with
    cval = pixel_mean / 255 (float64 from here on), crop [u:-d, l:-r], brightness, contrast, mean -- returns
    the f64 [h',w',3] array the reference hands to skimage.transform.resize."""
    mean = float(np.asarray(pixel_means).reshape(-1)[0])
    # u8 -> f32 / 255, flipped, x3, no augmentation and no mean (pixel_mean 0): blob.py:36
    im = prep_im_pre_resize(gray, flipped, None, None, np.zeros((1, 1, 3)))
    rot = skimage_rotate(im, angle_deg, cval=mean / 255.)
    H0, W0 = int(rot.shape[0]), int(rot.shape[1])
    view = rot
    if crop is not None:
        u, d, l, r = (int(v) for v in crop)
        if d < 1 or r < 1 or u < 0 or l < 0 or u + d >= H0 or l + r >= W0:
            raise ValueError("crop offsets out of range")
        view = rot[u:H0 - d, l:W0 - r]
    h, w = int(view.shape[0]), int(view.shape[1])
    out = torch.empty((h, w, 3), dtype=torch.float64, device=rot.device)
    with torch.cuda.device(rot.device):
        ws, n = _ws(rot.device)
        _lib.check(_lib.lib().wssdl_image_adjust_f64(
            _lib.ptr(view), h, w, 3, int(view.stride(0)),
            int(brightness_delta is not None), float(brightness_delta or 0.0),
            int(contrast_factor is not None), float(contrast_factor if contrast_factor is not None else 1.0),
            mean, _lib.ptr(out), _lib.ptr(ws), n, _lib.stream()), "wssdl_image_adjust_f64")
    return out


def torch_bilinear_resize(im, shape):
    """torch's bilinear filter (NOT skimage's: no fading borders, no clip); kept for callers that ask for it."""
    x = im.permute(2, 0, 1).unsqueeze(0).to(torch.float64)
    y = torch.nn.functional.interpolate(x, size=tuple(int(s) for s in shape), mode="bilinear", align_corners=False)
    return y.squeeze(0).permute(1, 2, 0).contiguous()


def prep_im_for_blob(gray, net_name, pixel_means, pixel_stds, target_size, max_size, is_training,
                     is_ws=False, flipped=False, rng=None, resize=None):
    """blob.py:34-79 for one grey plane.  Returns (im [h',w',3] on the GPU, im_scale).  The draws come
    from `rng` (default: numpy's global legacy stream, like the reference) in the reference's order:
    for a weak image (is_ws) the rotation angle (np.random.uniform, :39-41), the four crop offsets
    (np.random.random_integers, :43-46), then -- when training -- brightness (:50) and contrast (:55).
    A rotated image is float64 from skimage.transform.rotate on, like in the reference (the f32 plane goes
    through wssdl_image_warp, the crop is a view, the remaining steps run in wssdl_image_adjust_f64)."""
    rng = np.random if rng is None else rng
    crop = delta = factor = angle = None
    if is_ws:
        if cfg.TRAIN.USE_ROTATION:
            a = cfg.TRAIN.ROTATION_MAX_ANGLE
            angle = rng.uniform(-a, a)
        if cfg.TRAIN.USE_CROPPING:
            h0, w0 = (int(v) for v in gray.shape)
            m = cfg.TRAIN.CROPPING_MAX_MARGIN
            # random_integers(lo, hi) with a float hi == randint(lo, int(hi) + 1) of the legacy stream
            crop = (int(rng.randint(0, int(m * h0) + 1)), int(rng.randint(1, int(m * h0) + 1)),
                    int(rng.randint(0, int(m * w0) + 1)), int(rng.randint(1, int(m * w0) + 1)))
    if is_training:
        if cfg.TRAIN.USE_BRIGHTNESS_ADJUSTMENT:
            d = cfg.TRAIN.BRIGHTNESS_ADJUSTMENT_MAX_DELTA
            delta = rng.uniform(-d, d)
        if cfg.TRAIN.USE_CONTRAST_ADJUSTMENT:
            factor = rng.uniform(cfg.TRAIN.CONTRAST_ADJUSTMENT_LOWER_FACTOR,
                                 cfg.TRAIN.CONTRAST_ADJUSTMENT_UPPER_FACTOR)
    if angle is None:
        pre = prep_im_pre_resize(gray, flipped, delta, factor, pixel_means, crop=crop)
    else:
        pre = prep_im_pre_resize_rotated(gray, flipped, angle, delta, factor, pixel_means, crop=crop)
    h, w = pre.shape[:2]
    im_scale = float(target_size) / float(min(h, w))
    if np.round(im_scale * max(h, w)) > max_size:
        im_scale = float(max_size) / float(max(h, w))
    shape = (int(np.round(h * im_scale)), int(np.round(w * im_scale)))
    resized = (resize or skimage_resize)(pre, shape)
    std = float(np.asarray(pixel_stds).reshape(-1)[0])
    if net_name[:6] == 'Resnet':
        im = _scale_image(resized, std / 255.0, True)
    else:
        im = _scale_image(resized, 255.0, False)
    return im, im_scale


def _scale_image(resized, scale, divide):
    blob = im_list_to_blob([resized], scale=scale, divide=divide)
    return blob[0]


def im_list_to_blob(ims, scale=1.0, divide=False):
    """blob.py:19-32: images [h_i,w_i,3] (f32 or f64 GPU tensors) -> zero-padded f32 blob
    [n, max_h, max_w, 3].  `scale` / `divide` fold blob.py:75-77 into the copy (x / scale or x * scale)."""
    ims = [_lib.to_device(im, im.dtype if isinstance(im, torch.Tensor) else torch.float64) for im in ims]
    dev = ims[0].device
    Hm, Wm = max(im.shape[0] for im in ims), max(im.shape[1] for im in ims)
    blob = torch.empty((len(ims), Hm, Wm, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        for i, im in enumerate(ims):
            if im.dim() != 3 or im.shape[2] != 3 or im.dtype not in (torch.float32, torch.float64):
                raise ValueError("images must be [h,w,3] f32 / f64")
            _lib.check(_lib.lib().wssdl_image_to_blob(
                _lib.ptr(im), int(im.dtype == torch.float64), im.shape[0], im.shape[1], float(scale), int(divide),
                _lib.ptr(blob), i, len(ims), Hm, Wm, _lib.stream()), "wssdl_image_to_blob")
    return blob


def imread(path):
    """The grey plane of an image file as a [h,w] uint8 array -- what ``skimage.io.imread`` hands the reference for its
    single-channel B-mode images (roi_data_layer/minibatch_bus.py:269,294,304).  Decoding a file format is byte
    parsing with no arithmetic in it and stays on the HOST (PIL here; scikit-image is not installed); everything from
    the decoded plane on runs on the device.  Colour / 16-bit files are refused rather than converted silently."""
    from PIL import Image
    with Image.open(path) as im:
        if im.mode == "P":
            # skimage.io.imread expands a palette: a COLOUR palette gives an RGB array (not this pipeline's input),
            # an identity grey palette gives the same plane 'L' would -- only that one is accepted
            pal = im.getpalette() or []
            grey = len(pal) >= 768 and all(pal[3 * k] == pal[3 * k + 1] == pal[3 * k + 2] == k for k in range(256))
            if not grey:
                raise ValueError("%s: palette image whose palette is not the identity grey ramp -- skimage.io.imread "
                                 "would hand the reference an RGB array; convert the file first" % path)
        elif im.mode != "L":
            # ('1' included: skimage.io.imread returns a bool array for it, not a 0/255 plane)
            raise ValueError("%s: mode %s -- the reference's pipeline stacks ONE 8-bit grey plane three times "
                             "(minibatch_bus.py:270); convert the file first" % (path, im.mode))
        plane = np.asarray(im.convert("L"), dtype=np.uint8)
    if plane.ndim != 2:
        raise ValueError("%s: not a single plane" % path)
    return np.ascontiguousarray(plane)


def roidb_planes(roidb):
    """[{'image': path, 'flipped': bool}, ...] (the reference's roidb entries) -> (planes, flips)."""
    return [imread(e["image"]) for e in roidb], [bool(e.get("flipped", False)) for e in roidb]


def _get_image_blob(planes, flipped, net_name, scale_inds, is_training, is_ws, rng=None, resize=None):
    """roi_data_layer/minibatch_bus.py:259-283: one list of images, all supervised or all weak.  `planes` = decoded
    grey planes with `flipped` = their flags, or the reference's own first argument -- a roidb list of
    {'image': path, 'flipped': bool} entries -- with flipped=None."""
    if flipped is None:
        planes, flipped = roidb_planes(planes)
    ims, scales = [], []
    for i, (g, f) in enumerate(zip(planes, flipped)):
        im, s = prep_im_for_blob(g, net_name, PIXEL_MEANS, PIXEL_STDS, cfg.TRAIN.SCALES[int(scale_inds[i])],
                                 cfg.TRAIN.MAX_SIZE, is_training, is_ws=bool(is_ws), flipped=f, rng=rng, resize=resize)
        ims.append(im)
        scales.append(s)
    return im_list_to_blob(ims), scales


def _get_image_blob_joint(planes_s, flipped_s, planes_ws, flipped_ws, net_name, scale_inds, is_training,
                          rng=None, resize=None):
    """roi_data_layer/minibatch_bus.py:285-318: the combined mini-batch -- the supervised images first
    (is_ws=False), then the weak ones (is_ws=True: cropped), `scale_inds` indexed in that order, one RNG
    stream through all of them; zero-padded blob [n_s + n_ws, max_h, max_w, 3] and the im_scales.  With
    flipped_s = flipped_ws = None the first and third arguments are the reference's roidb_s / roidb_ws lists."""
    if flipped_s is None:
        planes_s, flipped_s = roidb_planes(planes_s)
    if flipped_ws is None:
        planes_ws, flipped_ws = roidb_planes(planes_ws)
    ims, scales = [], []
    k = 0
    for planes, flips, ws in ((planes_s, flipped_s, False), (planes_ws, flipped_ws, True)):
        for g, f in zip(planes, flips):
            im, s = prep_im_for_blob(g, net_name, PIXEL_MEANS, PIXEL_STDS, cfg.TRAIN.SCALES[int(scale_inds[k])],
                                     cfg.TRAIN.MAX_SIZE, is_training, is_ws=ws, flipped=f, rng=rng, resize=resize)
            ims.append(im)
            scales.append(s)
            k += 1
    return im_list_to_blob(ims), scales


def flip_boxes(boxes, width):
    """datasets/imdb.py:106-121 on a GPU tensor [n, >=4] f32 (returns a mirrored copy)."""
    b = _lib.to_device(boxes, torch.float32).clone()
    with torch.cuda.device(b.device):
        _lib.check(_lib.lib().wssdl_flip_boxes(_lib.ptr(b), b.shape[0], b.stride(0), float(width), _lib.stream()),
                   "wssdl_flip_boxes")
    return b
