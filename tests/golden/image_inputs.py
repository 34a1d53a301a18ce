"""Seeded synthetic inputs shared by tests/golden/make_golden_image.py and the image-path tests
(the fixtures store outputs only)."""
import numpy as np

SHAPES = [(291, 498), (535, 777), (578, 738), (594, 738), (578, 738)]     # the five sample TIFFs
SUB = (12, 12)               # stored sub-sampling of the [h, w] planes


def synth_plane(seed, h, w):
    """Speckle-like u8 plane."""
    rs = np.random.RandomState(seed)
    low = rs.uniform(0.2, 1.0, size=(h // 32 + 2, w // 32 + 2))
    mask = np.kron(low, np.ones((32, 32)))[:h, :w]
    return np.clip(rs.rayleigh(40.0, size=(h, w)) * mask, 0, 255).astype(np.uint8)


def nearest(im, shape):
    """The stand-in that marks where skimage.transform.resize sits (NOT a restatement of it)."""
    rr = np.minimum((np.arange(shape[0]) * im.shape[0] / float(shape[0])).astype(int), im.shape[0] - 1)
    cc = np.minimum((np.arange(shape[1]) * im.shape[1] / float(shape[1])).astype(int), im.shape[1] - 1)
    return im[rr][:, cc].astype(np.float64)
