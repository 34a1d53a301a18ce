#!/usr/bin/env python3
"""Golden vectors for the host image path: tests/golden/image_path.npz.

Runs ONLY in the build container (needs /root/reference).  Imports the reference's own
utils/blob.py (prep_im_for_blob, im_list_to_blob) from a scratch copy (lib2to3; fast_rcnn/config.py
beside it, as in oracle/ref_python_stage.py).  skimage is not installed and its resize is not
pinnable (version unknown), so a module named `skimage` is put in sys.modules whose
`transform.resize` is a RECORDER: it stores the array the reference hands it (= everything
blob.py:34-60 computes: /255, brightness, contrast, mean subtraction) and returns a deterministic
f64 array of the requested shape (nearest-neighbour pick of its input), after which the reference's
own code finishes (blob.py:74-77).  The fixtures therefore pin both halves around the resize with
the reference's arithmetic, and say nothing about the resize itself.

Inputs are synthetic u8 planes with the shapes of the five sample TIFFs of the reference
(SNUBH_BUS/TIFFImages: 291x498, 535x777, 578x738, 594x738, 578x738); tests regenerate them from
the seed, only sub-sampled outputs and f64 checksums are stored."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import ref_python_stage as stage  # noqa: E402

from image_inputs import SHAPES, SUB, nearest, synth_plane  # noqa: E402


def main():
    root = stage.stage()
    import shutil
    import subprocess
    dst = os.path.join(root, "utils", "blob.py")
    shutil.copy(os.path.join(stage.LIB, "utils", "blob.py"), dst)
    subprocess.check_call([sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n", dst],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rec = {}
    sk = types.ModuleType("skimage")
    sk.transform = types.ModuleType("skimage.transform")

    def resize(im, shape, *a, **k):
        rec["in"] = np.array(im, copy=True)
        rec["shape"] = tuple(int(s) for s in shape)
        return nearest(im, rec["shape"])
    sk.transform.resize = resize
    sys.modules["skimage"] = sk
    sys.modules["skimage.transform"] = sk.transform
    import importlib
    blob = importlib.import_module("utils.blob")
    cfg = importlib.import_module("fast_rcnn.config").cfg
    cfg.TRAIN.USE_ROTATION = False           # skimage.transform.rotate: not pinnable either
    cfg.TRAIN.USE_CROPPING = False           # a plain slice; exercised by the caller's view, not here
    out = {}
    finals = {}
    case = 0
    for si, (h, w) in enumerate(SHAPES):
        for flipped in (False, True):
            for net in ("Resnet_train", "VGGnet_train"):
                for train in (True, False):
                    if case % 3 == 2 and not train:      # thin out: keep ~2/3 of the combinations
                        case += 1
                        continue
                    name = "c%02d" % case
                    gray = synth_plane(100 + si, h, w)
                    im = np.dstack((gray, gray, gray))              # minibatch_bus.py:270
                    if flipped:
                        im = im[:, ::-1, :]                         # :271-272
                    np.random.seed(1000 + case)
                    st = np.random.get_state()
                    final, scale = blob.prep_im_for_blob(im, net, cfg.PIXEL_MEANS, cfg.PIXEL_STDS, 600, 1000,
                                                         train, is_ws=False)
                    np.random.set_state(st)                         # the draws the call consumed
                    delta = np.random.uniform(-cfg.TRAIN.BRIGHTNESS_ADJUSTMENT_MAX_DELTA,
                                              cfg.TRAIN.BRIGHTNESS_ADJUSTMENT_MAX_DELTA) if train else np.nan
                    factor = np.random.uniform(cfg.TRAIN.CONTRAST_ADJUSTMENT_LOWER_FACTOR,
                                               cfg.TRAIN.CONTRAST_ADJUSTMENT_UPPER_FACTOR) if train else np.nan
                    pre = rec["in"]
                    assert pre.dtype == np.float32 and pre.shape == (h, w, 3)
                    assert np.array_equal(pre[..., 0], pre[..., 1]) and np.array_equal(pre[..., 0], pre[..., 2])
                    assert final.dtype == np.float64 and final.shape[:2] == rec["shape"]
                    out[name + "/meta"] = np.array([si, h, w, int(flipped), int(net.startswith("Resnet")), int(train),
                                                    100 + si], np.int64)
                    out[name + "/draws"] = np.array([delta, factor], np.float64)
                    out[name + "/scale"] = np.array([scale], np.float64)
                    out[name + "/resize_shape"] = np.array(rec["shape"], np.int64)
                    out[name + "/pre_sub"] = pre[::SUB[0], ::SUB[1], 0].copy()
                    out[name + "/pre_sum"] = np.array([pre.astype(np.float64).sum(), np.abs(pre).astype(np.float64).sum()])
                    out[name + "/final_sub"] = final[::SUB[0], ::SUB[1], 0].astype(np.float64)
                    out[name + "/final_sum"] = np.array([final.sum(), np.abs(final).sum()])
                    if case in (0, 5, 9):
                        finals[case] = final
                    case += 1
    # ---- weak images: cropping (blob.py:43-48; four np.random.random_integers draws, then a slice) ----
    cfg.TRAIN.USE_CROPPING = True
    wcase = 0
    for si, (h, w) in enumerate(SHAPES):
        for flipped in (False, True):
            for train in (True, False):
                name = "w%02d" % wcase
                gray = synth_plane(200 + si, h, w)
                im = np.dstack((gray, gray, gray))
                if flipped:
                    im = im[:, ::-1, :]
                np.random.seed(3000 + wcase)
                st = np.random.get_state()
                final, scale = blob.prep_im_for_blob(im, "Resnet_train", cfg.PIXEL_MEANS, cfg.PIXEL_STDS, 600, 1000,
                                                     train, is_ws=True)
                np.random.set_state(st)                             # replay the draws the call consumed
                m = cfg.TRAIN.CROPPING_MAX_MARGIN
                crop = [np.random.random_integers(0, m * h), np.random.random_integers(1, m * h),
                        np.random.random_integers(0, m * w), np.random.random_integers(1, m * w)]
                delta = np.random.uniform(-cfg.TRAIN.BRIGHTNESS_ADJUSTMENT_MAX_DELTA,
                                          cfg.TRAIN.BRIGHTNESS_ADJUSTMENT_MAX_DELTA) if train else np.nan
                factor = np.random.uniform(cfg.TRAIN.CONTRAST_ADJUSTMENT_LOWER_FACTOR,
                                           cfg.TRAIN.CONTRAST_ADJUSTMENT_UPPER_FACTOR) if train else np.nan
                pre = rec["in"]
                assert pre.shape == (h - crop[0] - crop[1], w - crop[2] - crop[3], 3), (pre.shape, crop)
                out[name + "/meta"] = np.array([si, h, w, int(flipped), 1, int(train), 200 + si], np.int64)
                out[name + "/crop"] = np.array(crop, np.int64)
                out[name + "/draws"] = np.array([delta, factor], np.float64)
                out[name + "/scale"] = np.array([scale], np.float64)
                out[name + "/resize_shape"] = np.array(rec["shape"], np.int64)
                out[name + "/pre_sub"] = pre[::SUB[0], ::SUB[1], 0].copy()
                out[name + "/pre_sum"] = np.array([pre.astype(np.float64).sum(), np.abs(pre).astype(np.float64).sum()])
                out[name + "/final_sum"] = np.array([final.sum(), np.abs(final).sum()])
                wcase += 1
    out["n_crop_cases"] = np.array([wcase], np.int64)

    # ---- batch assembly of the combined mode: roi_data_layer/minibatch_bus.py:285-318
    # (_get_image_blob_joint: supervised images first with is_ws=False, then the weak ones with
    # is_ws=True, one global RNG stream through all of them, zero-padded blob).  skimage.io.imread is a
    # data source here: it hands back the synthetic plane registered under the "file name".
    mb_dst = os.path.join(root, "roi_data_layer", "minibatch_bus.py")
    os.makedirs(os.path.dirname(mb_dst), exist_ok=True)
    open(os.path.join(root, "roi_data_layer", "__init__.py"), "w").close()
    shutil.copy(os.path.join(stage.LIB, "roi_data_layer", "minibatch_bus.py"), mb_dst)
    subprocess.check_call([sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n", mb_dst],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    planes = {}
    sk.io = types.ModuleType("skimage.io")
    sk.io.imread = lambda path: planes[path]
    sys.modules["skimage.io"] = sk.io
    mb = importlib.import_module("roi_data_layer.minibatch_bus")
    calls = []
    orig_resize = sk.transform.resize

    def resize_log(im, shape, *a, **k):
        r = orig_resize(im, shape, *a, **k)
        calls.append((np.array(im, copy=True), tuple(int(v) for v in shape)))
        return r
    sk.transform.resize = resize_log
    blob.skimage.transform.resize = resize_log
    spec_s = [(300, 0, False), (301, 2, True)]                  # (plane seed, SHAPES index, flipped)
    spec_ws = [(302, 1, True), (303, 3, False), (304, 4, False)]
    roidb_s, roidb_ws = [], []
    for lst, spec in ((roidb_s, spec_s), (roidb_ws, spec_ws)):
        for seed, si, flipped in spec:
            key = "plane%d" % seed
            planes[key] = synth_plane(seed, *SHAPES[si])
            lst.append(dict(image=key, flipped=flipped))
    np.random.seed(4242)
    jb, jscales = mb._get_image_blob_joint(roidb_s, roidb_ws, "Resnet_train", np.zeros(5, np.int64), True)
    assert jb.dtype == np.float32 and jb.shape[0] == 5 and len(calls) == 5
    out["joint/spec"] = np.array([[seed, si, int(f), ws] for ws, spec in ((0, spec_s), (1, spec_ws))
                                  for seed, si, f in spec], np.int64)
    out["joint/seed"] = np.array([4242], np.int64)
    out["joint/scales"] = np.array(jscales, np.float64)
    out["joint/shape"] = np.array(jb.shape, np.int64)
    out["joint/sub"] = jb[:, ::SUB[0], ::SUB[1], 0].copy()
    out["joint/sum"] = np.array([jb.astype(np.float64).sum(), np.abs(jb).astype(np.float64).sum()])
    out["joint/pre_shapes"] = np.array([c[0].shape[:2] for c in calls], np.int64)
    out["joint/pre_sums"] = np.array([[c[0].astype(np.float64).sum(), np.abs(c[0]).astype(np.float64).sum()] for c in calls])
    out["joint/resize_shapes"] = np.array([c[1] for c in calls], np.int64)
    sk.transform.resize = orig_resize
    blob.skimage.transform.resize = orig_resize
    cfg.TRAIN.USE_CROPPING = False

    # im_list_to_blob on three images of different shapes (blob.py:19-32)
    ims = [finals[k] for k in sorted(finals)]
    b = blob.im_list_to_blob(ims)
    assert b.dtype == np.float32
    out["blob/cases"] = np.array(sorted(finals), np.int64)
    out["blob/shape"] = np.array(b.shape, np.int64)
    out["blob/sub"] = b[:, ::SUB[0], ::SUB[1], :].copy()
    out["blob/sum"] = np.array([b.astype(np.float64).sum(), np.abs(b).astype(np.float64).sum()])
    out["n_cases"] = np.array([case], np.int64)
    path = os.path.join(ROOT, "tests", "golden", "image_path.npz")
    np.savez_compressed(path, **out)
    print(path, case, "cases,", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
