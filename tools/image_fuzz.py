#!/usr/bin/env python3
"""Random image-path cases, device kernels against the oracle's restatement (bit for bit; the restatement of skimage 0.14.2's
resize / rotate / warp itself is 'parity unpinned', tests/test_oracle_resize.py): random input and output sizes (1 ... 700), f32 /
f64 inputs, one or three channels, rotation angles and fill values, projective matrices in both boundary modes, and the whole
prep_im_for_blob with the reference's default augmentation against oracle.prep_im_for_blob on the same RNG stream.
    python3 tools/image_fuzz.py [--cases 80] [--seed 0]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import np_oracle as O  # noqa: E402
from wssdl_bus_amd.utils import blob as B  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=80)
ap.add_argument("--seed", type=int, default=0)
args = ap.parse_args()
rs = np.random.RandomState(args.seed)
bad = 0
for k in range(args.cases):
    h, w = int(rs.randint(1, 400)), int(rs.randint(1, 400))
    oh, ow = int(rs.randint(1, 700)), int(rs.randint(1, 700))
    dt = np.float32 if k % 2 else np.float64
    a = (rs.rand(h, w, 3) * rs.uniform(0.1, 300) - rs.uniform(0, 50)).astype(dt) if k % 3 else (rs.rand(h, w) - 0.4).astype(dt)
    ok_r = np.array_equal(B.skimage_resize(torch.from_numpy(a).cuda(), (oh, ow)).cpu().numpy(), O.skimage_resize(a, (oh, ow)))
    angle, cval = float(rs.uniform(-180, 180)), float(rs.uniform(-1, 2))
    img = a if a.ndim == 3 else np.dstack((a, a, a))
    ok_t = np.array_equal(B.skimage_rotate(torch.from_numpy(img).cuda(), angle, cval=cval).cpu().numpy(),
                          O.skimage_rotate(img, angle, cval=cval))
    M = O.skimage_rotate_matrix(img.shape, float(rs.uniform(-30, 30)))
    M[2] = [rs.uniform(-3e-4, 3e-4), rs.uniform(-3e-4, 3e-4), 1.0]
    mode, clip = ("constant", "edge")[k % 2], bool(k % 3)
    ok_w = np.array_equal(B.skimage_warp(img, M, (oh, ow), mode=mode, cval=cval, clip=clip).cpu().numpy(),
                          O.skimage_warp(img, M, (oh, ow), mode=mode, cval=cval, clip=clip))
    # the whole preparation of one training image (rotation draw, crop draws for a weak image, brightness / contrast, resize)
    gray = rs.randint(0, 256, size=(int(rs.randint(60, 500)), int(rs.randint(60, 700)))).astype(np.uint8)
    is_ws, flipped, train = bool(k % 2), bool(k % 4 < 2), bool(k % 5)
    got, s1 = B.prep_im_for_blob(gray, "Resnet_train", B.PIXEL_MEANS, B.PIXEL_STDS, 600, 1000, train, is_ws=is_ws, flipped=flipped,
                                 rng=np.random.RandomState(k))
    want, s2 = O.prep_im_for_blob(gray, flipped, "Resnet_train", 600, 1000, train, is_ws, np.random.RandomState(k), O.skimage_resize,
                                    dict(USE_ROTATION=True))[:2]
    got = got.cpu().numpy()
    # (brightness / contrast means: f64 sums here, NumPy's pairwise f32 sums there, ~1 ulp of [0, 1] before the division by std / 255)
    ok_p = s1 == s2 and got.shape == want.shape and np.abs(got - want).max() <= 2e-6
    if not (ok_r and ok_t and ok_w and ok_p):
        bad += 1
        print("MISMATCH case %d %dx%d -> %dx%d %s: resize %s rotate %s warp %s prep %s" % (k, h, w, oh, ow, dt.__name__, ok_r, ok_t, ok_w, ok_p), flush=True)
    if (k + 1) % 20 == 0:
        print("case %d ok so far (%d mismatches)" % (k + 1, bad), flush=True)
print("cases %d mismatches %d" % (args.cases, bad))
sys.exit(1 if bad else 0)
