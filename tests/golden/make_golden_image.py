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
