"""utils.blob.imread: the host half of the image path (the reference decodes with skimage.io.imread,
roi_data_layer/minibatch_bus.py:269,294,304; here PIL).  No GPU needed."""
import numpy as np
import pytest


def test_imread_returns_the_grey_plane_of_tiff_png_bmp(tmp_path):
    from PIL import Image
    from wssdl_bus_amd.utils import blob as B
    p = np.random.RandomState(0).randint(0, 256, (37, 53)).astype(np.uint8)
    for ext in ("tif", "png", "bmp"):
        f = str(tmp_path / ("a." + ext))
        Image.fromarray(p).save(f)
        got = B.imread(f)
        assert got.dtype == np.uint8 and got.flags["C_CONTIGUOUS"] and np.array_equal(got, p), ext
    planes, flips = B.roidb_planes([{"image": str(tmp_path / "a.tif"), "flipped": True}, {"image": str(tmp_path / "a.png")}])
    assert flips == [True, False] and all(np.array_equal(q, p) for q in planes)


def test_imread_refuses_colour_and_16_bit_files(tmp_path):
    from PIL import Image
    from wssdl_bus_amd.utils import blob as B
    rgb = str(tmp_path / "rgb.png")
    Image.fromarray(np.zeros((4, 4, 3), np.uint8)).save(rgb)
    deep = str(tmp_path / "deep.tif")
    Image.fromarray((np.arange(16, dtype=np.uint16) * 4000).reshape(4, 4)).save(deep)
    for f in (rgb, deep):
        with pytest.raises(ValueError):
            B.imread(f)


def test_imread_palette_and_bilevel_files(tmp_path):
    """skimage.io.imread expands a colour palette to RGB and returns bool for 1-bit files: neither is the plane the
    reference's pipeline stacks, so both are refused; an identity grey palette decodes to the plane 'L' would give."""
    from PIL import Image
    from wssdl_bus_amd.utils import blob as B
    p = np.random.RandomState(1).randint(0, 256, (9, 11)).astype(np.uint8)
    grey = Image.fromarray(p).convert("P")
    grey.putpalette([v for k in range(256) for v in (k, k, k)])
    f = str(tmp_path / "grey_palette.png")
    grey.save(f)
    assert np.array_equal(B.imread(f), p)
    colour = Image.fromarray(p).convert("P")
    colour.putpalette([v for k in range(256) for v in (k, 255 - k, (3 * k) % 256)])
    f = str(tmp_path / "colour_palette.png")
    colour.save(f)
    with pytest.raises(ValueError, match="palette"):
        B.imread(f)
    f = str(tmp_path / "bilevel.png")
    Image.fromarray(p > 127).save(f)
    with pytest.raises(ValueError, match="mode 1"):
        B.imread(f)
