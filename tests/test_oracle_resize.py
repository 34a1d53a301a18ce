"""f4: the oracle's restatement of skimage.transform.resize / rotate (scikit-image 0.14.2 as pinned by the
reference's README; the library is absent -> PARITY UNPINNED) against hand-computed known answers.

Conventions under test (oracle/np_oracle.py cites the published sources): pixel centres at +0.5
(out (row, col) samples in (h/rows (row+.5) - .5, w/cols (col+.5) - .5)), order-1 interpolation with
floor / ceil neighbours, mode 'constant' (cval 0 outside the image: up-scaled borders fade), the result
clipped to the input's [min, max], no anti-aliasing when shrinking."""
import numpy as np

from oracle import np_oracle as O


def test_identity_is_exact():
    rs = np.random.RandomState(0)
    a = (rs.rand(7, 5, 3) - 0.3).astype(np.float32)
    out = O.skimage_resize(a, (7, 5))
    assert out.dtype == np.float64 and np.array_equal(out, a.astype(np.float64))
    assert np.array_equal(O.skimage_rotate(a, 0.0, cval=0.25), a.astype(np.float64))


def test_two_times_up_fades_towards_zero_at_the_border():
    a, b = -0.4, 0.8
    out = O.skimage_resize(np.array([[a, b]]), (2, 4))
    # columns sample at -0.25, 0.25, 0.75, 1.25; rows at -0.25 and 0.25 (one neighbour outside: weight 0.25 of 0)
    cols = np.array([0.75 * a, 0.75 * a + 0.25 * b, 0.25 * a + 0.75 * b, 0.75 * b])
    want = np.stack([0.75 * cols, 0.75 * cols])
    assert np.allclose(out, want, rtol=0, atol=1e-16)
    # all-positive input: the faded border is clipped back up to the input's minimum
    out = O.skimage_resize(np.array([[0.4, 0.8]]), (2, 4))
    assert out.min() == 0.4 and out.max() <= 0.8


def test_two_times_down_averages_pairs_without_antialiasing():
    p = np.array([[0.1, 0.3, -0.2, 0.6]])
    out = O.skimage_resize(p, (1, 2))
    assert np.allclose(out, [[0.2, 0.2]], rtol=0, atol=1e-16)
    # 4 -> 1: samples at 1.5 only (the outer pixels are never read: no anti-aliasing)
    assert np.allclose(O.skimage_resize(p, (1, 1)), [[0.05]], rtol=0, atol=1e-16)


def test_non_integer_scale():
    p = np.array([[0.2, -0.1, 0.7]])
    out = O.skimage_resize(p, (1, 2))             # samples at 0.25 and 1.75
    assert np.allclose(out, [[0.75 * 0.2 + 0.25 * -0.1, 0.25 * -0.1 + 0.75 * 0.7]], rtol=0, atol=1e-16)


def test_one_pixel_rows_and_columns():
    a = np.arange(12, dtype=np.float64).reshape(4, 3) / 10 - 0.5
    out = O.skimage_resize(a, (1, 3))             # row sample at 1.5: mean of rows 1 and 2, columns untouched
    assert np.allclose(out, 0.5 * (a[1] + a[2])[None, :], rtol=0, atol=1e-16)
    b = a[:3]                                     # odd height: the middle row itself
    assert np.array_equal(O.skimage_resize(b, (1, 3)), b[1:2])
    out = O.skimage_resize(a, (4, 1))             # column sample at 1.0
    assert np.array_equal(out, a[:, 1:2])


def test_channels_are_interpolated_independently_and_clip_uses_the_whole_array():
    rs = np.random.RandomState(1)
    a = (rs.rand(9, 11, 3) - 0.5).astype(np.float32)
    out = O.skimage_resize(a, (20, 17))
    for ch in range(3):
        # a single channel is clipped to ITS OWN range when resized alone: compare inside the joint range only
        alone = O.skimage_warp(a[:, :, ch], O.skimage_resize_matrix(a.shape, (20, 17)), (20, 17), clip=False)
        assert np.array_equal(np.clip(alone, a.min(), a.max()), out[:, :, ch])
    assert out.min() >= a.min() and out.max() <= a.max()


def test_rotate_quarter_turn_and_cval():
    a = np.arange(25, dtype=np.float64).reshape(5, 5) / 25
    out = O.skimage_rotate(a, 90)
    assert np.allclose(out, np.rot90(a), rtol=0, atol=1e-14)         # counter-clockwise, centre (2, 2)
    # corners rotated in from outside take cval; cval outside the input's range survives the clip
    out = O.skimage_rotate(np.full((6, 6), 0.5), 45, cval=0.9)
    assert out[0, 0] == 0.9 and out[5, 5] == 0.9 and abs(out[3, 3] - 0.5) < 1e-15 and out.max() == 0.9
    # cval inside the range: plain clip
    out = O.skimage_rotate(np.linspace(0, 1, 36).reshape(6, 6), 45, cval=0.25)
    assert out[0, 0] == 0.25


def test_rotated_prep_path_is_float64_and_draws_the_angle_first():
    """blob.py:39-60 with USE_ROTATION: one uniform draw before the crop's four; f64 from rotate() on."""
    rs = np.random.RandomState(4)
    gray = rs.randint(0, 256, size=(60, 90)).astype(np.uint8)
    rng = np.random.RandomState(11)
    im, scale, pre = O.prep_im_for_blob(gray, False, "Resnet_train", 600, 1000, True, True, rng, O.skimage_resize,
                                        c=dict(USE_ROTATION=True))
    ref = np.random.RandomState(11)
    angle = ref.uniform(-5, 5)
    crop = O.draw_crop(ref, 60, 90, 0.05)
    delta, factor = ref.uniform(-0.2, 0.2), ref.uniform(0.2, 1.8)
    want = O.prep_im_pre_resize(gray, False, delta, factor, crop=crop, rotate_angle=angle)
    assert pre.dtype == np.float64 and np.array_equal(pre, want)
    assert im.shape[0] == 600 or im.shape[1] == 1000


def test_against_an_independent_bilinear_with_constant_padding():
    """A second implementation of "order-1 interpolation, constant padding that takes part in the interpolation":
    scipy.ndimage.map_coordinates(order=1, mode='grid-constant') evaluated at the coordinates the restated resize /
    rotate maps produce.  (scipy's plain 'constant' mode does NOT interpolate beyond the edge -- skimage's own
    _warp_fast does, which is why its up-scaled borders fade; 'grid-constant' is the matching scipy mode.)  Agreement
    to 1e-12 before the clip; the clip is the oracle's own."""
    from scipy import ndimage
    rs = np.random.RandomState(6)
    img = rs.rand(23, 31) - 0.4
    for shape in ((23, 31), (50, 61), (11, 9), (1, 31), (40, 1)):
        M = O.skimage_resize_matrix(img.shape, shape)
        x = np.arange(shape[1])[None, :] * np.ones((shape[0], 1))
        y = np.arange(shape[0])[:, None] * np.ones((1, shape[1]))
        cols = M[0, 0] * x + M[0, 1] * y + M[0, 2]
        rows = M[1, 0] * x + M[1, 1] * y + M[1, 2]
        ref = ndimage.map_coordinates(img, [rows, cols], order=1, mode="grid-constant", cval=0.0)
        got = O.skimage_warp(img, M, shape, clip=False)
        assert np.abs(got - ref).max() <= 1e-12, shape
        assert np.array_equal(O.skimage_resize(img, shape), np.clip(got, img.min(), img.max()))
    for angle, cval in ((7.5, 0.3), (-33.0, -0.2)):
        M = O.skimage_rotate_matrix(img.shape, angle)
        x = np.arange(31)[None, :] * np.ones((23, 1))
        y = np.arange(23)[:, None] * np.ones((1, 31))
        cols = M[0, 0] * x + M[0, 1] * y + M[0, 2]
        rows = M[1, 0] * x + M[1, 1] * y + M[1, 2]
        ref = ndimage.map_coordinates(img, [rows, cols], order=1, mode="grid-constant", cval=cval)
        got = O.skimage_warp(img, M, img.shape, cval=cval, clip=False)
        assert np.abs(got - ref).max() <= 1e-12, angle
