"""The pointwise colourspaces of ConvertRGBToGeneric / ConvertGenericToRGB
(MagickCore/colorspace.c:411-595, :122-305) and ModulateImage's colour models
(enhance.c:3826-3890) against the compiled reference.  Q16 results are bit-identical (a last-bit
difference of a device pow / atan2 / cbrt flips a Quantum rounding only on an exact tie); float
Quantum results may differ by one float ULP for the same reason."""
import numpy as np
import pytest

from conftest import make_pixels, to_device, assert_parity

pytestmark = pytest.mark.gpu

Q16, HDRI = np.uint16, np.float32

SPACES = ["CMY", "HCL", "HCLp", "HSB", "HSI", "HSL", "HSV", "HWB", "LCH", "LCHab", "LCHuv", "LMS", "Luv",
          "xyY", "YCbCr", "YDbDr", "YIQ", "YPbPr", "YUV", "Jzazbz", "DisplayP3", "Adobe98", "ProPhoto",
          "OkLab", "OkLCH", "CAT02LMS"]


def special_pixels(px):
    """Rows of the cases the hue models branch on: grays, black, white, primaries, equal pairs."""
    top = 65535 if px.dtype == np.uint16 else 65535.0
    cases = [(0, 0, 0), (top, top, top), (top, 0, 0), (0, top, 0), (0, 0, top), (top, top, 0), (0, top, top),
             (top, 0, top), (1234, 1234, 1234), (40000, 40000, 100), (100, 40000, 40000), (40000, 100, 40000),
             (1, 0, 0), (0, 1, 0), (65534, 65535, 65533)]
    for i, c in enumerate(cases):
        px[0, i % px.shape[1], :3] = c
    return px


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("space", SPACES)
def test_srgb_to_colorspace(im, refmod, space, dtype):
    px = special_pixels(make_pixels(41, 53, 4, dtype, seed=len(space) * 7 + 1))
    dev = im.Image(to_device(px), colorspace="sRGB")
    im.transform_image_colorspace(dev, space)
    want = refmod.RefImage(px, "sRGB").colorspace(space).numpy()
    assert_parity(dev.numpy(), want, True, "sRGB -> %s" % space, max_ulp=1)
    assert np.array_equal(dev.numpy()[:, :, 3], px[:, :, 3])          # alpha untouched


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("space", SPACES)
def test_colorspace_to_srgb(im, refmod, space, dtype):
    px = special_pixels(make_pixels(37, 45, 3, dtype, seed=len(space) * 11 + 3))
    dev = im.Image(to_device(px), colorspace=space)
    im.transform_image_colorspace(dev, "sRGB")
    want = refmod.RefImage(px, space).colorspace("sRGB").numpy()
    assert_parity(dev.numpy(), want, True, "%s -> sRGB" % space, max_ulp=1)


# the table-driven half of sRGBTransformImage / TransformsRGBImage (colorspace.c:1226-1420, :2560-2790)
TABLE_SPACES = ["OHTA", "Rec601YCbCr", "Rec709YCbCr", "YCC", "scRGB", "Log"]


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("space", TABLE_SPACES)
def test_srgb_to_table_driven_colorspace(im, refmod, space, dtype):
    """Every table entry is one rounded product of the map index: formed in place, in the table's
    own operations — bit-identical on both Quantum types (scRGB: linear RGB's loop, whose float
    results may differ by an ULP of the device pow)."""
    px = special_pixels(make_pixels(45, 67, 4, dtype, seed=len(space) * 5 + 2))
    if dtype == HDRI:
        px[3, :8, 0] = [-5.0, 70000.0, 0.25, 0.5, 65534.5, 65535.0, 1.5, 2.5]      # clamps and .5 indices
    dev = im.Image(to_device(px), colorspace="sRGB")
    im.transform_image_colorspace(dev, space)
    want = refmod.RefImage(px, "sRGB").colorspace(space).numpy()
    if space == "scRGB":
        assert_parity(dev.numpy(), want, True, "sRGB -> scRGB", max_ulp=1)
    elif space == "Log" and dtype == HDRI:
        # the table index is the decoded float sample rounded: where the device pow's last bit moves it
        # across n + 1/2 the result is the neighbouring table entry (a handful of samples per million)
        same = dev.numpy().view(np.uint32) == want.view(np.uint32)
        assert same.mean() > 0.9999 and np.abs(dev.numpy() - want).max() < 2.0, "sRGB -> Log (float)"
    else:
        bits = np.uint32 if dtype == HDRI else np.uint16
        assert np.array_equal(dev.numpy().view(bits), want.view(bits)), "sRGB -> %s" % space
    assert np.array_equal(dev.numpy()[:, :, 3], px[:, :, 3])


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("space", ["OHTA", "Rec601YCbCr", "Rec709YCbCr", "scRGB", "Log"])
def test_table_driven_colorspace_to_srgb(im, refmod, space, dtype):
    px = special_pixels(make_pixels(39, 51, 3, dtype, seed=len(space) * 3 + 4))
    dev = im.Image(to_device(px), colorspace=space)
    im.transform_image_colorspace(dev, "sRGB")
    want = refmod.RefImage(px, space).colorspace("sRGB").numpy()
    if space in ("scRGB", "Log"):
        assert_parity(dev.numpy(), want, True, "%s -> sRGB" % space, max_ulp=1)
    else:
        bits = np.uint32 if dtype == HDRI else np.uint16
        assert np.array_equal(dev.numpy().view(bits), want.view(bits)), "%s -> sRGB" % space


def test_ycc_to_srgb_is_declined(im):
    """YCC's way back goes through the 1389-entry film curve YCCMap (colorspace.c:2789-2797): not
    accelerated, and the image is left as it was."""
    px = make_pixels(8, 9, 3, Q16, seed=1)
    dev = im.Image(to_device(px), colorspace="YCC")
    with pytest.raises(Exception):
        im.transform_image_colorspace(dev, "sRGB")
    assert np.array_equal(dev.numpy(), px)


@pytest.mark.parametrize("pair", [("HSL", "Lab"), ("YUV", "RGB"), ("OkLab", "LCHuv"), ("XYZ", "HWB"),
                                  ("OHTA", "Rec709YCbCr"), ("Lab", "YCC")])
def test_colorspace_to_colorspace_goes_through_srgb(im, refmod, pair):
    """TransformImageColorspace X -> Y is X -> sRGB -> Y (colorspace.c:1751-1783), each step
    rounded to Quantum."""
    px = make_pixels(30, 40, 4, Q16, seed=9)
    dev = im.Image(to_device(px), colorspace=pair[0])
    im.transform_image_colorspace(dev, pair[1])
    want = refmod.RefImage(px, pair[0]).colorspace(pair[1]).numpy()
    assert_parity(dev.numpy(), want, True, "%s -> %s" % pair)


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("model", ["HCL", "HCLp", "HSB", "HSI", "HSL", "HSV", "HWB", "LCH", "LCHab", "LCHuv"])
@pytest.mark.parametrize("percent", [(120.0, 80.0, 130.0), (90.0, 150.0, 100.0)])
def test_modulate_colour_models(im, refmod, model, percent, dtype):
    """Percentages that are multiples of ten put many results EXACTLY on a rounding tie (a channel
    is a multiple of 0.1 levels, e.g. 40960.5): there the last bit of the reference's own libm
    decides the level, so Q16 allows one level on those samples and nowhere else; NaN results of
    the float build (HCLp of pure white: 0/0 in the reference too) must be NaN on both sides."""
    px = special_pixels(make_pixels(33, 47, 4, dtype, seed=21))
    dev = im.Image(to_device(px))
    im.modulate_image(dev, percent[0], percent[1], percent[2], model)
    got = dev.numpy()
    want = refmod.RefImage(px).modulate(percent[0], percent[1], percent[2], model).numpy()
    if dtype == Q16:
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        assert d.max() <= 1 and (d != 0).mean() < 0.005, "modulate %s %s: max %d, %d differ" % (
            model, percent, d.max(), int((d != 0).sum()))
        return
    both_nan = np.isnan(got) & np.isnan(want)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert_parity(np.where(both_nan, np.float32(0), got), np.where(both_nan, np.float32(0), want), True,
                  "modulate %s %s" % (model, percent), max_ulp=1)
