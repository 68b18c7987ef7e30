"""GPU parity tests: the HIP path (through the C ABI of libmagickhip.so) against
the compiled reference CPU implementation (oracle/_ref) on the same seeded
inputs.  Bar: bit-exact for Q16 and for float Quantum in EXACT precision
(1 float ULP where device libm `pow` is involved); +-1 level in FAST precision."""
import numpy as np
import pytest

from conftest import make_pixels, to_device, assert_parity

pytestmark = pytest.mark.gpu

Q16, HDRI = np.uint16, np.float32


def run_pair(im, refmod, pixels, colorspace="sRGB", **image_kw):
    dev = im.Image(to_device(pixels), colorspace=colorspace, **image_kw)
    ref = refmod.RefImage(pixels, colorspace)
    return dev, ref


# ------------------------------------------------------------------ BlurImage
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("shape,sigma", [((61, 97, 4), 2.0), ((200, 300, 4), 10.0),
                                         ((33, 40, 3), 1.5), ((50, 70, 1), 3.0),
                                         ((45, 64, 2), 2.5), ((1, 1, 4), 2.0),
                                         ((1, 130, 4), 2.0), ((130, 1, 4), 2.0),
                                         ((20, 20, 4), 10.0)])
def test_blur_exact(im, refmod, dtype, shape, sigma):
    px = make_pixels(*shape, dtype)
    dev, ref = run_pair(im, refmod, px)
    got = im.blur_image(dev, 0.0, sigma).numpy()
    want = ref.blur(0.0, sigma).numpy()
    assert_parity(got, want, True, "blur %s sigma=%g" % (shape, sigma))


@pytest.mark.parametrize("pattern", ["columns", "rows", "checker", "alpha"])
@pytest.mark.parametrize("sigma", [3.0, 10.0])
def test_blur_exact_on_rounding_ties(im, refmod, pattern, sigma):
    """EXACT forms its sums with fused multiply-adds and recomputes, in the reference's own
    operation order, every result that lies within 1e-6 of a rounding tie (device_common.hpp,
    Tie64).  Images that alternate between v and v+1 put the blurred value on a tie (the even and
    the odd taps of a Gaussian each sum to 1/2 to within 1e-16) almost everywhere: the reference's
    level there is decided by the last bits of ITS summation, and the result must still be
    bit-identical."""
    rows, cols = 96, 130
    y, x = np.mgrid[0:rows, 0:cols]
    base = {"columns": x & 1, "rows": y & 1, "checker": (x + y) & 1, "alpha": x & 1}[pattern]
    px = np.empty((rows, cols, 4), dtype=np.uint16)
    for c, level in enumerate((1000, 32767, 65534)):
        px[:, :, c] = level + base
    px[:, :, 3] = 65535 if pattern != "alpha" else 40000 + (y & 1)
    dev, ref = run_pair(im, refmod, px)
    assert_parity(im.blur_image(dev, 0.0, sigma).numpy(), ref.blur(0.0, sigma).numpy(), True,
                  "blur on ties, %s sigma %g" % (pattern, sigma))


@pytest.mark.parametrize("pattern", ["float_ties", "mixed_sign", "wide_range", "small_alpha",
                                     "negative_alpha", "powers_of_two", "integers"])
@pytest.mark.parametrize("sigma", [3.0, 10.0])
def test_blur_hdri_on_float_rounding_ties(im, refmod, pattern, sigma):
    """float Quantum, long kernels: fused fp64 sums plus the float-rounding tie check of
    Accum::finish() (convolve.hip).  A frame that alternates between a float and its upper
    neighbour blurs to the midpoint of the two (to 1e-16), where the float the reference stores
    is decided by the last bits of its own summation order; mixed signs, 60 decades of range,
    alpha sums around PerceptibleReciprocal's clamp and negative alpha stress the error bound the
    check relies on.  All bit-identical."""
    rows, cols = 90, 150
    rng = np.random.default_rng(77)
    y, x = np.mgrid[0:rows, 0:cols]
    px = (rng.random((rows, cols, 4)) * 65535.0).astype(np.float32)
    if pattern == "float_ties":
        for c, level in enumerate((1000.25, 32767.5, 3.0e-3)):
            low = np.float32(level)
            px[:, :, c] = np.where(((x + y) & 1) == 1, np.nextafter(low, np.float32(np.inf)), low)
        px[:, :, 3] = 65535.0
    elif pattern == "mixed_sign":
        px[:, :, :3] -= 32768.0
        px[:, :, 3] = np.where((x & 3) == 0, 0.0, px[:, :, 3])
    elif pattern == "wide_range":
        px[:, :, :3] = (10.0 ** rng.uniform(-30, 30, (rows, cols, 3))).astype(np.float32)
        px[:, :, 3] = 65535.0
    elif pattern == "small_alpha":
        px[:, :, 3] = (10.0 ** rng.uniform(-12, 0, (rows, cols))).astype(np.float32)
        px[:, : cols // 3, 3] = 0.0
    elif pattern == "negative_alpha":
        px[:, :, 3] -= 20000.0
    elif pattern == "powers_of_two":
        px[:, :, :3] = np.where(((x & 1) == 1)[:, :, None], 4096.0, np.nextafter(np.float32(4096.0), np.float32(0)))
        px[:, :, 3] = 65535.0
    else:
        px = np.floor(px)
    dev, ref = run_pair(im, refmod, px)
    got = im.blur_image(dev, 0.0, sigma).numpy()
    want = ref.blur(0.0, sigma).numpy()
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), "HDRI blur, %s sigma %g: %d of %d samples differ, first at %s" % (
        pattern, sigma, int((~same).sum()), same.size, np.argwhere(~same)[:3].tolist())


@pytest.mark.parametrize("sigma", [0.303, 2.0, 10.05])
def test_blur_fast_structured_ties_bound(im, refmod, sigma):
    """FAST is +-1 BY CONSTRUCTION: the row pass is the exact-integer one
    (convolve_fused_exact.hip), so the Quantum-rounded intermediate is the reference's own, and
    the f16 column pass is within +-1 of the reference's column pass on that input.  A
    checkerboard of two levels (alpha included) and a frame of 0..3-level alpha put row-pass
    values on exact rounding ties over whole windows — the inputs on which round 2's f16 row pass
    reached 2."""
    rows, cols = 133, 310
    rng = np.random.default_rng(31)
    checker = np.empty((rows, cols, 4), dtype=np.uint16)
    checker[:] = (np.add.outer(np.arange(rows), np.arange(cols)) % 2 * 40000 + 100)[:, :, None]
    sparse = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    sparse[:, :, 3] = rng.integers(0, 4, (rows, cols), dtype=np.uint16)
    pair = np.empty((rows, cols, 4), dtype=np.uint16)          # two colours averaged under zero centre alpha
    pair[:, :, :3] = np.where((np.arange(cols) % 2 == 0)[None, :, None], 20001, 40000)
    pair[:, :, 3] = np.where(np.arange(cols) % 4 == 1, 0, 65535)[None, :]
    im.set_precision(im.PRECISION_FAST)
    try:
        for name, px in (("checkerboard", checker), ("0..3-level alpha", sparse), ("colour pair", pair)):
            dev, ref = run_pair(im, refmod, px)
            d = np.abs(im.blur_image(dev, 0.0, sigma).numpy().astype(np.int64) -
                       ref.blur(0.0, sigma).numpy().astype(np.int64))
            assert d.max() <= 1, "%s sigma %g: max %d" % (name, sigma, d.max())
    finally:
        im.set_precision(im.PRECISION_EXACT)


@pytest.mark.parametrize("channels", [4, 3])
@pytest.mark.parametrize("sigma", [1.0, 4.0, 10.0])
def test_blur_exact_integer_kernel_counts_its_recomputations(im, refmod, channels, sigma):
    """EXACT BlurImage through the exact-integer kernel: bit-identical on a frame with every alpha
    structure (opaque, random, 0..3 levels, fully transparent regions, a hard alpha edge), and the
    number of pixels that needed the reference's own operation order stays small except where the
    alpha sum is tiny (the error bound grows with 1/alpha)."""
    rows, cols = 300, 520
    rng = np.random.default_rng(int(sigma * 7) + channels)
    px = rng.integers(0, 65536, (rows, cols, channels), dtype=np.uint16)
    if channels == 4:
        px[:, 100:200, 3] = 65535
        px[:, 200:300, 3] = rng.integers(0, 4, (rows, 100), dtype=np.uint16)
        px[:, 300:400, 3] = 0
        px[100:, 400:, 3] = 65535
        px[:100, 400:, 3] = 0
    dev, ref = run_pair(im, refmod, px)
    lib = im.load()
    lib.MhExactBlurRecomputed(1)
    got = im.blur_image(dev, 0.0, sigma).numpy()
    recomputed = lib.MhExactBlurRecomputed(0)
    assert_parity(got, ref.blur(0.0, sigma).numpy(), True, "exact-integer blur, %d channels sigma %g" % (channels, sigma))
    assert recomputed < rows * cols // 4, recomputed
    if channels == 3:
        assert recomputed < 64, recomputed


def test_blur_radius_argument(im, refmod):
    px = make_pixels(64, 80, 4, Q16)
    dev, ref = run_pair(im, refmod, px)
    assert_parity(im.blur_image(dev, 5.0, 2.0).numpy(), ref.blur(5.0, 2.0).numpy(), True, "blur 5x2")


def adversarial_blur_frames(rows, cols):
    """Frames on which a blur that rounds an APPROXIMATE intermediate leaves the +-1 contract: both passes on
    rounding ties (levels v + (x&1) + (y&1): the even and the odd taps of a Gaussian each sum to 1/2), the
    frames of test_blur_fast_structured_ties_bound, tiny / binary alpha, and a sprite (opaque rectangles on a
    transparent ground: whole windows whose only opaque samples lie under the kernel's outermost taps)."""
    rng = np.random.default_rng(31)
    y, x = np.mgrid[0:rows, 0:cols]
    ties = np.empty((rows, cols, 4), dtype=np.uint16)
    for c, level in enumerate((1000, 32767, 65533, 40000)):
        ties[:, :, c] = level + (x & 1) + (y & 1)
    opaque_ties = ties.copy()
    opaque_ties[:, :, 3] = 65535
    checker = np.empty((rows, cols, 4), dtype=np.uint16)
    checker[:] = ((x + y) % 2 * 40000 + 100)[:, :, None]
    sparse = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    sparse[:, :, 3] = rng.integers(0, 4, (rows, cols), dtype=np.uint16)
    pair = np.empty((rows, cols, 4), dtype=np.uint16)
    pair[:, :, :3] = np.where((np.arange(cols) % 2 == 0)[None, :, None], 20001, 40000)
    pair[:, :, 3] = np.where(np.arange(cols) % 4 == 1, 0, 65535)[None, :]
    binary = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    binary[:, :, 3] = np.where(rng.random((rows, cols)) < 0.5, 0, 65535)
    sprite = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    sprite[:, :, 3] = 0
    for _ in range(7):
        y0, x0 = int(rng.integers(0, rows - 2)), int(rng.integers(0, cols - 2))
        sprite[y0: y0 + int(rng.integers(1, 25)), x0: x0 + int(rng.integers(1, 25)), 3] = 65535
    return (("ties in both passes", ties), ("opaque ties in both passes", opaque_ties), ("checkerboard", checker),
            ("0..3-level alpha", sparse), ("colour pair", pair), ("binary alpha", binary), ("sprite", sprite))


def layout_pair(im, refmod, px, layout, call):
    """(device image, reference result of `call`) for RGBA (alpha-weighted), four plain channels or RGB."""
    if layout == "rgb":
        px = np.ascontiguousarray(px[:, :, :3])
        return im.Image(to_device(px)), call(refmod.RefImage(px)).numpy()
    if layout == "plain4":
        want = np.concatenate([call(refmod.RefImage(px[:, :, c].copy())).numpy().reshape(px.shape[0], px.shape[1], 1)
                               for c in range(4)], axis=2)
        return im.Image(to_device(px), has_alpha=False), want
    return im.Image(to_device(px)), call(refmod.RefImage(px)).numpy()


@pytest.mark.parametrize("layout", ["rgba", "plain4", "rgb"])
@pytest.mark.parametrize("radius,sigma", [(30.0, 2.0), (40.0, 3.0), (25.0, 2.0), (7.0, 0.856), (12.0, 2.0), (3.0, 0.47),
                                          (25.0, 5.055)])
def test_blur_fast_kernels_with_tiny_outer_taps(im, refmod, radius, sigma, layout):
    """The DEFAULT mode on BlurImage / UnsharpMaskImage with a radius far beyond the sigma (-blur 30x2): the outer
    taps are exact zeros (morphology.c:2494-2495) or tiny (1e-8) beside the centre.  Where such a tap is the only
    one that meets an opaque sample (a sprite on a transparent ground) it IS the result — sum(k*alpha*p) /
    sum(k*alpha), morphology.c:2746-2776 — and round 5's library answered tens of thousands of levels off
    (round 2's all-f16 kernel, reachable through an untested fall-through; the f16 taps' fixed factor of 256
    left a tap of 4e-7 with nine bits).  Now: zero taps are dropped (they add +0.0), f16 operands are scaled to
    the kernel's largest tap, and an alpha-weighted frame under taps below 2^-19 of the largest takes the exact
    kernels — BlurImage within one level (bit-identical on those), UnsharpMaskImage within the contract, on
    the adversarial frames."""
    import bench
    # (smallest tap 4e-7 = 2^-17.6 of the largest: the f16 kernels keep it — the two matrix-core passes here, the
    # alpha certificate of the one-launch form wants a tap of 1e-6)
    resolved = (radius, sigma) == (25.0, 5.055)
    im.set_precision(im.PRECISION_FAST)
    try:
        for name, px in adversarial_blur_frames(133, 310):
            holder = {}
            image, want = layout_pair(im, refmod, px, layout, lambda r: r.blur(radius, sigma))
            launched = set(bench.kernel_profile(im, lambda: holder.update(o=im.blur_image(image, radius, sigma)), 1))
            what = "fast blur %gx%g %s, %s (%s)" % (radius, sigma, layout, name, " ".join(sorted(launched)))
            if layout == "rgba" and not resolved:
                assert "blur_fused_hybrid" not in launched, launched
            assert_parity(holder["o"].numpy().reshape(want.shape), want, layout == "rgba" and not resolved, what)
            gain, threshold = 1.0, 0.02
            image, want = layout_pair(im, refmod, px, layout, lambda r: r.unsharp(radius, sigma, gain, threshold))
            blurred = layout_pair(im, refmod, px, layout, lambda r: r.blur(radius, sigma))[1].astype(np.int64)
            launched = set(bench.kernel_profile(
                im, lambda: holder.update(o=im.unsharp_mask_image(image, radius, sigma, gain, threshold)), 1))
            got = holder["o"].numpy().reshape(want.shape)
            what = "fast unsharp %gx%g %s, %s (%s)" % (radius, sigma, layout, name, " ".join(sorted(launched)))
            if launched == {"unsharp_fused_exact"} or (layout == "rgba" and not resolved):
                assert_parity(got, want, True, what)                  # on the reference's own blur: bit-identical
            else:
                # plain channels on the f16 column pass: a blurred sample one level off moves the sharpened one
                # by 1 + gain levels, and flips the threshold test where |2(p - b)| sits on it (effect.c:4364-4369)
                source = (px[:, :, :3] if layout == "rgb" else px).astype(np.int64)
                d = np.abs(got.astype(np.int64) - want.astype(np.int64))
                d[np.abs(2 * np.abs(source - blurred.reshape(source.shape)) - 65535.0 * threshold) <= 2.0] = 0
                assert d.max() <= 2, "%s: max %d" % (what, d.max())
    finally:
        im.set_precision(im.PRECISION_EXACT)


@pytest.mark.parametrize("layout", ["rgba", "plain4", "rgb"])
@pytest.mark.parametrize("radius,sigma", [(0.0, 11.0), (0.0, 13.4), (50.0, 20.0)])
def test_blur_fast_outside_the_one_launch_kernels(im, refmod, radius, sigma, layout):
    """FAST BlurImage with more than 81 taps takes MorphologyApply's two passes on the f16 matrix cores
    (convolve_mfma.hip); still within one level on the adversarial frames."""
    import bench
    im.set_precision(im.PRECISION_FAST)
    try:
        for name, px in adversarial_blur_frames(133, 310):
            holder = {}
            image, want = layout_pair(im, refmod, px, layout, lambda r: r.blur(radius, sigma))
            launched = set(bench.kernel_profile(im, lambda: holder.update(o=im.blur_image(image, radius, sigma)), 1))
            assert not any(k.startswith("blur_fused") for k in launched), launched
            assert_parity(holder["o"].numpy().reshape(want.shape), want, False,
                          "fast blur %gx%g %s, %s" % (radius, sigma, layout, name))
    finally:
        im.set_precision(im.PRECISION_EXACT)


@pytest.mark.parametrize("kind", ["opaque", "smooth"])
def test_blur_other_distributions(im, refmod, kind):
    px = make_pixels(120, 150, 4, Q16, kind=kind)
    dev, ref = run_pair(im, refmod, px)
    assert_parity(im.blur_image(dev, 0.0, 4.0).numpy(), ref.blur(0.0, 4.0).numpy(), True, kind)


def test_blur_fast_precision_within_one_level(im, refmod):
    px = make_pixels(240, 320, 4, Q16)
    dev, ref = run_pair(im, refmod, px)
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.blur_image(dev, 0.0, 10.0).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    exact_fraction = assert_parity(got, ref.blur(0.0, 10.0).numpy(), False, "fast blur")
    assert exact_fraction > 0.95


def test_blur_fast_tap_tables_are_kept_by_content(im, refmod):
    """The FAST blur keeps its tap tables on the device, looked up by content (runtime.cpp,
    shared_table: 32 entries): more distinct kernels than entries, then the first ones again
    (evicted and rebuilt), every result against the reference."""
    px = make_pixels(48, 96, 4, Q16)
    dev, ref = run_pair(im, refmod, px)
    sigmas = [0.5 + 0.11 * i for i in range(40)] + [0.5, 0.61, 4.79]
    im.set_precision(im.PRECISION_FAST)
    try:
        for sigma in sigmas:
            got = im.blur_image(dev, 0.0, sigma).numpy()
            assert_parity(got, ref.blur(0.0, sigma).numpy(), False, "fast blur sigma %g" % sigma)
    finally:
        im.set_precision(im.PRECISION_EXACT)


@pytest.mark.parametrize("shape", [(5, 7), (1, 40), (40, 1), (33, 70), (64, 16), (17, 129), (130, 31)])
@pytest.mark.parametrize("sigma", [0.8, 2.0, 6.5])
def test_blur_fast_odd_shapes(im, refmod, shape, sigma):
    """FAST RGBA blur (matrix-core passes) on images smaller than one strip / one step, with
    partial strips and steps, and kernels from 7 to 53 taps: within +-1 level everywhere,
    edges included."""
    px = make_pixels(shape[0], shape[1], 4, Q16, seed=shape[0] * 131 + shape[1])
    dev, ref = run_pair(im, refmod, px)
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.blur_image(dev, 0.0, sigma).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(got, ref.blur(0.0, sigma).numpy(), False, "fast blur %s sigma %g" % (shape, sigma))


@pytest.mark.parametrize("shape", [(5, 7), (1, 40), (40, 1), (33, 70), (64, 16), (17, 129), (130, 31),
                                   (70, 66), (67, 132)])
@pytest.mark.parametrize("sigma", [0.8, 2.0, 6.5])
def test_blur_fast_rgb_without_alpha(im, refmod, shape, sigma):
    """FAST blur of a 3-channel (6-byte pixel) image: the matrix-core passes in their plain
    mode, with row pitches that are and are not multiples of 8 bytes."""
    px = make_pixels(shape[0], shape[1], 3, Q16, seed=shape[0] * 17 + shape[1])
    dev, ref = run_pair(im, refmod, px)
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.blur_image(dev, 0.0, sigma).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(got, ref.blur(0.0, sigma).numpy(), False, "fast RGB blur %s sigma %g" % (shape, sigma))


@pytest.mark.parametrize("shape", [(33, 70), (17, 129), (130, 31)])
@pytest.mark.parametrize("sigma", [1.5, 6.5])
def test_blur_fast_four_plain_channels(im, refmod, shape, sigma):
    """Four channels without an alpha trait (CMYK's layout): every channel is convolved on its
    own, so the result must match the reference's RGB blur of the first three channels and its
    gray blur of the fourth."""
    px = make_pixels(shape[0], shape[1], 4, Q16, seed=shape[1])
    px[::7, ::5, 3] = 0
    want = np.concatenate([refmod.RefImage(px[:, :, :3].copy()).blur(0.0, sigma).numpy(),
                           refmod.RefImage(px[:, :, 3].copy()).blur(0.0, sigma).numpy().reshape(shape + (1,))],
                          axis=2)
    dev = im.Image(to_device(px), has_alpha=False)
    assert_parity(im.blur_image(dev, 0.0, sigma).numpy(), want, True, "plain 4-channel blur")
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.blur_image(dev, 0.0, sigma).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(got, want, False, "fast plain 4-channel blur")


def test_convolve_fast_signed_separable_kernel_plain_channels(im):
    """A separable kernel with negative taps through the plain matrix-core mode: FAST within
    +-1 level of EXACT, clamping at both ends of the range included."""
    import torch
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randint(-32768, 32768, (301, 260, 3), generator=g, device="cuda", dtype=torch.int16)
    img = im.Image(a.view(torch.uint16))
    kernel = "9x1: -0.05,-0.1,0.15,0.25,0.5,0.25,0.15,-0.1,-0.05"
    exact = im.convolve_image(img, kernel).pixels.view(torch.int16).to(torch.int32) & 0xffff
    im.set_precision(im.PRECISION_FAST)
    try:
        fast = im.convolve_image(img, kernel).pixels.view(torch.int16).to(torch.int32) & 0xffff
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert int((fast - exact).abs().max()) <= 1


@pytest.mark.parametrize("shape", [(150, 331), (70, 64), (33, 65), (1, 200), (200, 1), (17, 2)])
@pytest.mark.parametrize("sigma", [0.6, 2.5, 10.0])
def test_blur_and_unsharp_fast_rgb_single_launch(im, refmod, shape, sigma, options):
    """RGB (6-byte pixels, no alpha) through the single-launch fused kernels as four plain channels
    whose fourth is zero (MFMA_PLAIN3: only the pixel loads and stores differ); strips and segments
    ragged at both edges."""
    import bench
    px = make_pixels(shape[0], shape[1], 3, Q16, seed=shape[0] + shape[1])
    dev, ref = run_pair(im, refmod, px)
    holder = {}
    im.set_precision(im.PRECISION_FAST)
    try:
        launched = set(bench.kernel_profile(im, lambda: holder.update(b=im.blur_image(dev, 0.0, sigma)), 1))
        if shape[1] >= 2 and shape[0] >= 2:
            assert launched == {"blur_fused_hybrid"}, launched
        launched = set(bench.kernel_profile(
            im, lambda: holder.update(u=im.unsharp_mask_image(dev, 0.0, sigma, 1.5, 0.01)), 1))
        if shape[1] >= 2 and shape[0] >= 2:
            assert launched == {"unsharp_fused_exact"}, launched
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(holder["b"].numpy(), ref.blur(0.0, sigma).numpy(), False, "fast RGB blur %s" % (shape,))
    # FAST UnsharpMaskImage in one launch runs on the reference's own blur (both passes exact): bit-identical
    if shape[1] >= 2 and shape[0] >= 2:
        assert_parity(holder["u"].numpy(), ref.unsharp(0.0, sigma, 1.5, 0.01).numpy(), True, "fast RGB unsharp %s" % (shape,))


@pytest.mark.parametrize("shape", [(150, 331), (67, 64), (130, 2), (257, 33), (516, 70)])
@pytest.mark.parametrize("sigma", [0.6, 2.0, 5.0, 10.0])
def test_gray_blur_and_unsharp_as_four_row_bands(im, refmod, shape, sigma, options):
    """A one-channel Q16 frame through the one-launch kernels as four row bands = four channels of a frame a
    quarter as tall (operators.cpp fused_blur_gray_bands, pointwise.hip gray_bands_pack_kernel): heights that
    are not multiples of four, bands barely taller than the rows added to them (the route declines below that:
    the frame's own two passes), every kind of frame edge.  FAST BlurImage within one level, EXACT BlurImage and
    UnsharpMaskImage in both modes bit-identical (morphology.c:2654-2979, effect.c:4364-4369)."""
    import bench
    options.set("MAGICKHIP_GRAY_BANDS_MIN_PIXELS", "0")
    px = make_pixels(shape[0], shape[1], 1, Q16, seed=shape[0] + shape[1] + int(sigma * 10))
    # rows that tell the bands apart, and the frame's first and last row unlike their neighbours
    px[0, :, 0] = 65535
    px[-1, :, 0] = 0
    dev, ref = run_pair(im, refmod, px)
    taps = im.optimal_kernel_width_1d(0.0, sigma)
    banded = (shape[0] + 3) // 4 >= 2 * (taps // 2) and taps <= 81
    holder = {}
    for precision, exact in ((im.PRECISION_FAST, False), (im.PRECISION_EXACT, True)):
        im.set_precision(precision)
        try:
            launched = set(bench.kernel_profile(im, lambda: holder.update(b=im.blur_image(dev, 0.0, sigma)), 1))
            if banded:
                assert launched == {"gray_bands_pack", "blur_fused_exact" if exact else "blur_fused_hybrid",
                                    "gray_bands_unpack"}, launched
            else:
                assert "gray_bands_pack" not in launched, launched
            launched = set(bench.kernel_profile(
                im, lambda: holder.update(u=im.unsharp_mask_image(dev, 0.0, sigma, 1.5, 0.01)), 1))
            if banded:
                assert launched == {"gray_bands_pack", "unsharp_fused_exact", "gray_bands_unpack"}, launched
        finally:
            im.set_precision(im.PRECISION_EXACT)
        assert_parity(holder["b"].numpy(), ref.blur(0.0, sigma).numpy(), exact, "gray blur %s, precision %d" % (shape, precision))
        if banded:
            assert_parity(holder["u"].numpy(), ref.unsharp(0.0, sigma, 1.5, 0.01).numpy(), True,
                          "gray unsharp %s, precision %d" % (shape, precision))


@pytest.mark.parametrize("channels", [4, 3])
@pytest.mark.parametrize("sigma", [3.2, 5.0, 8.0, 11.0, 14.0])
def test_blur_fast_every_ring_size(im, refmod, channels, sigma):
    """One sigma per ring geometry of the matrix-core kernel not covered above (tap counts 27,
    41, 65, 87 and 111: 5 to 9 sixteen-sample chunks, each with its own LDS layout)."""
    px = make_pixels(150, 140, channels, Q16, seed=int(sigma * 10))
    dev, ref = run_pair(im, refmod, px)
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.blur_image(dev, 0.0, sigma).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(got, ref.blur(0.0, sigma).numpy(), False, "fast blur sigma %g, %d channels" % (sigma, channels))


@pytest.mark.parametrize("frame", ["random", "alternating columns", "alternating rows", "flat with spikes"])
@pytest.mark.parametrize("layout", ["rgb", "plain4", "gray"])
@pytest.mark.parametrize("sigma", [10.5, 12.5, 14.0])
def test_blur_fast_beyond_81_taps_on_frames_without_alpha(im, refmod, sigma, layout, frame, options):
    """FAST BlurImage beyond the one-launch kernel's reach (83 to 113 taps) on frames without alpha weighting: both
    passes on the f16 matrix cores with undivided f32 sums between them (operators.cpp fused_blur; gray frames as
    four row bands).  The reference rounds the intermediate to a level, this route does not: frames whose row sums
    sit on x.5 (columns alternating v, v+1: the two halves of the taps are 0.5 each to 1e-9) put that difference at
    its largest.  Within one level; MAGICKHIP_NO_LONG_BLUR_SUMS: the general route (fp64 row pass) beside it."""
    channels = {"rgb": 3, "plain4": 4, "gray": 1}[layout]
    rows, cols = (300, 131) if layout == "gray" else (150, 140)
    options.set("MAGICKHIP_GRAY_BANDS_MIN_PIXELS", "0")
    rng = np.random.default_rng(int(sigma * 10) + channels)
    if frame == "random":
        px = rng.integers(0, 65536, (rows, cols, channels), dtype=np.uint16)
    elif frame == "alternating columns":
        base = rng.integers(0, 65535, (1, 1, channels))
        px = (base + (np.arange(cols) % 2)[None, :, None]).astype(np.uint16) * np.ones((rows, 1, 1), dtype=np.uint16)
    elif frame == "alternating rows":
        base = rng.integers(0, 65535, (1, 1, channels))
        px = (base + (np.arange(rows) % 2)[:, None, None]).astype(np.uint16) * np.ones((1, cols, 1), dtype=np.uint16)
    else:
        px = np.full((rows, cols, channels), 31000, dtype=np.uint16)
        px[rng.random((rows, cols)) < 0.01] = 65535
        px[rng.random((rows, cols)) < 0.01] = 0
    px = np.ascontiguousarray(px)
    if channels == 4:
        dev = im.Image(to_device(px), has_alpha=False)
        want = np.concatenate([refmod.RefImage(px[:, :, c].copy()).blur(0.0, sigma).numpy().reshape(rows, cols, 1)
                               for c in range(4)], axis=2)
    else:
        dev, ref = run_pair(im, refmod, px)
        want = ref.blur(0.0, sigma).numpy().reshape(px.shape)
    assert 81 < im.optimal_kernel_width_1d(0.0, sigma) <= 113
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.blur_image(dev, 0.0, sigma).numpy().reshape(px.shape)
        options.set("MAGICKHIP_NO_LONG_BLUR_SUMS", "1")
        general = im.blur_image(dev, 0.0, sigma).numpy().reshape(px.shape)
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(got, want, False, "fast blur 0x%g, %s, %s" % (sigma, layout, frame))
    assert_parity(general, want, False, "fast blur 0x%g, %s, %s (general route)" % (sigma, layout, frame))


@pytest.mark.parametrize("channels", [1, 2, 3, 4])
@pytest.mark.parametrize("radius,sigma", [(0.0, 1.5), (0.0, 4.0), (2.0, 3.0)])
def test_gaussian_blur_fast_is_separated(im, refmod, channels, radius, sigma):
    """FAST GaussianBlurImage: the 2-D Gaussian kernel is an outer product, so the library runs
    it as a row and a column pass over float sums (one division at the end, as the reference's
    2-D loop) instead of width x height taps per pixel: within +-1 level of the reference's
    2-D result, on random data, edges included."""
    px = make_pixels(83, 96, channels, Q16, seed=channels * 7 + int(sigma * 10))
    dev, ref = run_pair(im, refmod, px)
    want = ref.gaussian_blur(radius, sigma).numpy()
    import bench
    holder = {}
    im.set_precision(im.PRECISION_FAST)
    try:
        launched = set(bench.kernel_profile(im, lambda: holder.update(out=im.gaussian_blur_image(dev, radius, sigma)), 1))
    finally:
        im.set_precision(im.PRECISION_EXACT)
    got = holder["out"].numpy()
    cells = im.kernel_to_numpy("Gaussian:%gx%g" % (radius, sigma))[0]
    width = cells.shape[0]
    # (alpha-weighted frames keep kernels with a cell below 2^-13 of the largest off the f16 2-D kernel: a sprite's
    # result can be that cell alone, convolve2d_mfma.hip)
    resolved = channels == 3 or float(cells.min()) >= float(cells.max()) * 2.0 ** -13
    if channels >= 3 and width <= 13 and resolved:   # small kernels: the w x h sum in one launch is cheaper than two passes
        assert launched == {"conv2d_mfma"}, launched
    elif channels >= 3:                 # both passes on the matrix cores, float sums in between
        assert launched == {"conv_row", "conv_column"}, launched
    else:
        assert "separable_finish" in launched and "premultiply" in launched, launched
    assert_parity(got, want, False, "fast gaussian %gx%g c%d" % (radius, sigma, channels))


@pytest.mark.parametrize("channels,alpha", [(4, True), (4, False), (3, False), (1, False), (2, True)])
@pytest.mark.parametrize("kernel", ["3x3: 1,2,1 2,4,2 1,2,1", "5x5: 1,4,6,4,1 4,16,24,16,4 6,24,36,24,6 4,16,24,16,4 1,4,6,4,1",
                                    "Gaussian:3x1.2", "Gaussian:6x2", "Gaussian:7x2", "Square:1", "Square:2",
                                    "3x5: 1,2,1 2,4,2 3,6,3 2,4,2 1,2,1"])
def test_convolve_fast_small_outer_product_kernels(im, refmod, kernel, channels, alpha, options):
    """FAST Convolve with a small kernel that is an outer product: below 5 x 5 cells the generic kernel (bit-identical),
    up to 13 x 13 the w x h sum in one launch on the matrix cores (convolve2d_mfma.hip), beyond that the two separated
    passes — each faster than the next on its range (operators.cpp separable_convolve; 4096^2 RGBA 5 x 5: 0.10 ms
    against 0.21).  Within one level of the reference on every route, and the separated passes' result beside it."""
    import bench
    px = make_pixels(97, 131, channels, Q16, seed=len(kernel) + channels)
    dev, ref = run_pair(im, refmod, px, has_alpha=alpha) if channels in (2, 4) else run_pair(im, refmod, px)
    if channels == 4 and not alpha:
        want = np.concatenate([refmod.RefImage(px[:, :, c].copy()).set_artifact("convolve:scale", "!")
                               .morphology("Convolve", 1, kernel).numpy().reshape(97, 131, 1) for c in range(4)], axis=2)
    else:
        want = ref.set_artifact("convolve:scale", "!").morphology("Convolve", 1, kernel).numpy().reshape(px.shape)
    shape = im.kernel_to_numpy(kernel)[0].shape
    holder = {}
    im.set_precision(im.PRECISION_FAST)
    try:
        launched = set(bench.kernel_profile(
            im, lambda: holder.update(out=im.morphology_image(dev, "Convolve", 1, kernel, scale=(1.0, 1))), 1))
        options.set("MAGICKHIP_SEPARABLE_SMALL", "1")
        separated = im.morphology_image(dev, "Convolve", 1, kernel, scale=(1.0, 1)).numpy().reshape(px.shape)
    finally:
        im.set_precision(im.PRECISION_EXACT)
    if shape[0] * shape[1] < 25:
        assert not ({"conv_row", "conv_column", "conv2d_mfma"} & launched), launched
    elif max(shape) <= 13 and channels >= 3:
        assert launched == {"conv2d_mfma"}, launched
    assert_parity(holder["out"].numpy().reshape(px.shape), want, False, "fast %s c%d alpha=%s via %s" % (kernel[:20], channels, alpha, sorted(launched)))
    assert_parity(separated, want, False, "fast %s c%d (separated passes)" % (kernel[:20], channels))


def test_gaussian_blur_fast_opaque_and_plain_four_channels(im, refmod):
    """Fully opaque alpha (the largest sums the float intermediate carries) and four channels
    without an alpha trait through the matrix-core sums passes."""
    px = make_pixels(120, 150, 4, Q16, seed=12)
    px[:, :, 3] = 65535
    px[:40, :, :3] = 65535
    dev, ref = run_pair(im, refmod, px)
    want = ref.gaussian_blur(0.0, 6.0).numpy()
    plain_want = np.concatenate([refmod.RefImage(px[:, :, :3].copy()).gaussian_blur(0.0, 6.0).numpy(),
                                 refmod.RefImage(px[:, :, 3].copy()).gaussian_blur(0.0, 6.0).numpy().reshape(120, 150, 1)],
                                axis=2)
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.gaussian_blur_image(dev, 0.0, 6.0).numpy()
        plain = im.gaussian_blur_image(im.Image(to_device(px), has_alpha=False), 0.0, 6.0).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(got, want, False, "fast gaussian, opaque")
    assert_parity(plain, plain_want, False, "fast gaussian, four plain channels")


@pytest.mark.parametrize("case", ["transparent_band", "tiny_alpha", "checker"])
def test_gaussian_blur_fast_alpha_cases(im, refmod, case):
    rng = np.random.default_rng(5)
    rows, cols = 90, 110
    px = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    if case == "transparent_band":
        px[20:60, :, 3] = 0
    elif case == "tiny_alpha":
        px[:, :, 3] = rng.integers(0, 4, (rows, cols))
    else:
        y, x = np.mgrid[0:rows, 0:cols]
        px[:, :, 3] = ((x + y) & 1) * 65535
    dev, ref = run_pair(im, refmod, px)
    want = ref.gaussian_blur(0.0, 2.5).numpy()
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.gaussian_blur_image(dev, 0.0, 2.5).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(got, want, False, "fast gaussian, %s" % case)


@pytest.mark.parametrize("kernel", ["Gaussian:0x0.992", "Gaussian:0x1.375",
                                    "5x5: 1e-7,0.04,0.04,0.04,2e-7 0.04,0.05,0.04,0.04,0.04 0.04,0.04,0.08,0.04,0.04 "
                                    "0.04,0.04,0.04,0.03,0.04 3e-7,0.04,0.04,0.04,1e-7"])
def test_convolve_2d_fast_tiny_cells_on_a_sprite_frame(im, refmod, kernel):
    """An alpha-weighted result is a quotient of sums: on a sprite (opaque rectangles on a transparent ground) a pixel
    beside a rectangle's corner sees ONE opaque sample under the kernel's corner cell, and that cell — 1e-7 of the
    largest — is the whole result (morphology.c:2968-2977).  The f16 2-D kernel's terms do not carry it (ten levels off,
    found by the randomised run when small Gaussians were first sent there): such kernels keep the separated / fp64
    routes on alpha-weighted frames.  FAST within one level."""
    import bench
    rng = np.random.default_rng(12)
    px = rng.integers(0, 65536, (120, 150, 4), dtype=np.uint16)
    px[:, :, 3] = 0
    for y, x, h, w in ((10, 10, 20, 24), (50, 60, 1, 1), (70, 20, 13, 2), (90, 100, 25, 40), (31, 35, 3, 3)):
        px[y: y + h, x: x + w, 3] = 65535
    dev, ref = run_pair(im, refmod, px)
    want = ref.morphology("Convolve", 1, kernel).numpy()
    holder = {}
    im.set_precision(im.PRECISION_FAST)
    try:
        launched = set(bench.kernel_profile(im, lambda: holder.update(out=im.morphology_image(dev, "Convolve", 1, kernel)), 1))
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert "conv2d_mfma" not in launched, launched
    assert_parity(holder["out"].numpy(), want, False, "fast %s on a sprite frame via %s" % (kernel[:18], sorted(launched)))


@pytest.mark.parametrize("channels", [3, 4, 2])
def test_convolve_fast_outer_product_kernel_with_offset_origin(im, refmod, channels, options):
    """A hand-written outer-product kernel (5 x 5, origin off centre, unnormalised) is separated
    like the Gaussian: the origin of each axis and the reversed walk must carry over — and, this small, takes the
    w x h sum in one launch first (three and four channels; separable_convolve): both within one level."""
    import bench
    column, row = (0.1, 0.2, 0.3, 0.25, 0.15), (0.1, 0.2, 0.3, 0.2, 0.1)
    kernel = "5x5+1+3: " + " ".join(",".join("%g" % (c * r) for r in row) for c in column)
    px = make_pixels(77, 93, channels, Q16, seed=channels)
    dev, ref = run_pair(im, refmod, px)
    want = ref.convolve(kernel).numpy()
    holder = {}
    im.set_precision(im.PRECISION_FAST)
    try:
        launched = set(bench.kernel_profile(im, lambda: holder.update(out=im.convolve_image(dev, kernel)), 1))
        options.set("MAGICKHIP_SEPARABLE_SMALL", "1")
        separated = set(bench.kernel_profile(im, lambda: holder.update(sep=im.convolve_image(dev, kernel)), 1))
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert "morph2d" not in launched and "morph2d" not in separated, (launched, separated)
    assert launched == ({"conv2d_mfma"} if channels >= 3 else separated), launched
    assert {"conv_row", "conv_column"} <= separated, separated
    assert_parity(holder["out"].numpy(), want, False, "outer-product kernel, %d channels" % channels)
    assert_parity(holder["sep"].numpy(), want, False, "outer-product kernel, %d channels, separated" % channels)


def test_convolve_fast_signed_outer_product_kernel(im, refmod):
    """Sobel is an outer product with cells of both signs: plain channels (RGB) are separated
    (+-1), alpha-weighted channels (RGBA) stay on the fp64 kernels in FAST mode (identical),
    because sum(k*alpha) can vanish there."""
    rgb = make_pixels(70, 81, 3, Q16, seed=31)
    rgba = make_pixels(70, 81, 4, Q16, seed=32)
    im.set_precision(im.PRECISION_FAST)
    try:
        got3 = im.convolve_image(im.Image(to_device(rgb)), "Sobel").numpy()
        got4 = im.convolve_image(im.Image(to_device(rgba)), "Sobel").numpy()
        row4 = im.convolve_image(im.Image(to_device(rgba)), "3x1: -1,0,1").numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(got3, refmod.RefImage(rgb).convolve("Sobel").numpy(), False, "fast Sobel, RGB")
    assert_parity(got4, refmod.RefImage(rgba).convolve("Sobel").numpy(), True, "fast Sobel, RGBA")
    assert_parity(row4, refmod.RefImage(rgba).convolve("3x1: -1,0,1").numpy(), True, "fast signed row kernel, RGBA")


def test_convolve_fast_non_separable_kernel_stays_exact(im, refmod):
    """A kernel that is not an outer product and is too small (or has taps of both signs under
    alpha weighting) for the matrix-core 2-D kernel takes the generic fp64 kernel in FAST mode
    too: bit-identical.  Disk:2.5 (5 x 5) is the smallest the matrix-core kernel takes: +-1."""
    px = make_pixels(60, 71, 4, Q16, seed=2)
    dev, ref = run_pair(im, refmod, px)
    im.set_precision(im.PRECISION_FAST)
    try:
        for kernel in ("Disk:2.5", "3x3: 0,1,0 1,-3,1 0,1,1", "Gaussian:0x1.2",
                       "5x5: 1,1,1,1,1 1,1,1,1,1 1,1,-9,1,1 1,1,1,1,1 1,1,1,1,1"):
            got = im.convolve_image(dev, kernel).numpy()
            exact = kernel not in ("Gaussian:0x1.2", "Disk:2.5")
            assert_parity(got, ref.convolve(kernel).numpy(), exact, "fast convolve %s" % kernel)
    finally:
        im.set_precision(im.PRECISION_EXACT)


@pytest.mark.parametrize("walk", [False, True])
@pytest.mark.parametrize("alpha", [True, False])
@pytest.mark.parametrize("kernel", ["Disk:15", "Disk:7.3", "Octagon:5", "Diamond:4", "Plus:3",
                                    "7x5: 1,2,3,4,3,2,1 2,4,6,8,6,4,2 3,6,9,13,9,6,2 2,4,6,8,6,4,2 1,2,3,4,3,2,1",
                                    "6x6+1+4: 1,0,2,nan,1,3 0,1,1,2,nan,1 2,2,0,1,1,1 nan,1,3,1,0,2 1,1,1,1,2,0 3,0,1,2,1,1",
                                    "Ring:10,14"])
def test_convolve_2d_fast_on_matrix_cores(im, refmod, kernel, alpha, walk, options):
    """FAST ConvolveMorphology with a non-separable kernel of 5 x 5 cells or more (RGBA with
    alpha-weighted colour, or four plain channels): the w x h sum as h banded products on the
    matrix cores (convolve2d_mfma.hip) — flat disks, weighted and asymmetric user kernels, NaN
    cells, origins off centre, frames ragged against the 64-column strips and 32-row steps; the
    kernel normalised as `-define convolve:scale='!'` does; `walk`: one workgroup per strip walks
    all four steps through its ring of rows (a small frame is otherwise cut into single steps).
    Within one level of the reference, and of the generic kernel it replaces."""
    import bench
    options.set("MAGICKHIP_NO_EXACT_2D", "1")     # (integer cells otherwise take convolve2d_exact.hip)
    if walk:
        options.set("MAGICKHIP_CONV2D_CUTS", "1")
    px = make_pixels(107 if walk else 75, 150, 4, Q16, seed=len(kernel))
    if alpha:
        px[10:30, 20:60, 3] = np.random.default_rng(3).integers(0, 4, (20, 40))       # tiny alpha
        px[40:50, 100:140, 3] = 0                                                         # transparent
    dev = im.Image(to_device(px), has_alpha=alpha)
    ref = refmod.RefImage(px) if alpha else None
    holder = {}
    im.set_precision(im.PRECISION_FAST)
    try:
        launched = set(bench.kernel_profile(
            im, lambda: holder.update(out=im.morphology_image(dev, "Convolve", 1, kernel, scale=(1.0, 1))), 1))
        options.set("MAGICKHIP_NO_MFMA_2D", "1")
        generic = im.morphology_image(dev, "Convolve", 1, kernel, scale=(1.0, 1)).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert launched == {"conv2d_mfma"}, launched
    got = holder["out"].numpy()
    if alpha:
        want = ref.set_artifact("convolve:scale", "!").morphology("Convolve", 1, kernel).numpy()
    else:
        want = np.concatenate([refmod.RefImage(px[:, :, c].copy()).set_artifact("convolve:scale", "!")
                               .morphology("Convolve", 1, kernel).numpy().reshape(px.shape[0], 150, 1) for c in range(4)], axis=2)
    assert_parity(got, want, False, "2-D convolve %s alpha=%s" % (kernel, alpha))
    assert_parity(generic, want, False, "generic 2-D convolve %s" % kernel)


@pytest.mark.parametrize("kernel", ["Disk:15", "Octagon:5", "Ring:10,14",
                                    "6x6+1+4: 1,0,2,nan,1,3 0,1,1,2,nan,1 2,2,0,1,1,1 nan,1,3,1,0,2 1,1,1,1,2,0 3,0,1,2,1,1"])
def test_convolve_2d_fast_on_matrix_cores_rgb(im, refmod, kernel, options):
    """The same kernel on an RGB frame (6-byte pixels): three plain channels, the matrix tile's
    fourth entry zero (convolve2d_mfma.hip, MFMA_PLAIN3); frames ragged against the tiles.
    Within one level of the reference."""
    import bench
    options.set("MAGICKHIP_NO_EXACT_2D", "1")
    px = make_pixels(75, 150, 3, Q16, seed=len(kernel) + 3)
    dev, ref = run_pair(im, refmod, px)
    holder = {}
    im.set_precision(im.PRECISION_FAST)
    try:
        launched = set(bench.kernel_profile(
            im, lambda: holder.update(out=im.morphology_image(dev, "Convolve", 1, kernel, scale=(1.0, 1))), 1))
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert launched == {"conv2d_mfma"}, launched
    want = ref.set_artifact("convolve:scale", "!").morphology("Convolve", 1, kernel).numpy()
    assert_parity(holder["out"].numpy(), want, False, "2-D convolve %s RGB" % kernel)


@pytest.mark.parametrize("shape", [(64, 80), (33, 71), (2, 2), (70, 2), (1, 40), (129, 17)])
@pytest.mark.parametrize("gain,threshold", [(1.0, 0.02), (2.5, 0.0), (0.6, 0.2), (1.3, 1.0 / 65535.0)])
@pytest.mark.parametrize("single_launch", [True, False])
def test_unsharp_mask_fast_fused(im, refmod, shape, gain, threshold, single_launch, options):
    """FAST UnsharpMaskImage on RGBA Q16: the column pass applies the threshold/gain epilogue
    while it copies its results out (no blurred frame in memory).  In the one launch that does both
    passes the blur is the reference's own (exact integer sums in both passes, round 4: a blurred
    sample one level off would move the result by `gain` levels and flip the threshold test next to
    it), so the result is BIT-IDENTICAL — no tolerance, no exempted region.
    With the one launch switched off (separate fp64 row pass + column pass) the old bound holds: a
    blurred sample that differs by one level moves the result by at most 1+gain levels, and can
    flip the threshold test only when 2|p-b| sits on the threshold itself."""
    import bench
    if not single_launch:
        options.set("MAGICKHIP_NO_FUSED_BLUR", "1")
    px = make_pixels(shape[0], shape[1], 4, Q16, seed=shape[0] + 3 * shape[1])
    dev, ref = run_pair(im, refmod, px)
    want = ref.unsharp(0.0, 2.0, gain, threshold).numpy().astype(np.int64)
    blurred = ref.blur(0.0, 2.0).numpy().astype(np.int64)
    holder = {}
    im.set_precision(im.PRECISION_FAST)
    try:
        launched = set(bench.kernel_profile(
            im, lambda: holder.update(out=im.unsharp_mask_image(dev, 0.0, 2.0, gain, threshold)), 1))
    finally:
        im.set_precision(im.PRECISION_EXACT)
    if shape[1] >= 2:
        assert launched == ({"unsharp_fused_exact"} if single_launch else {"conv_row", "conv_column"}), launched
    got = holder["out"].numpy().astype(np.int64)
    diff = np.abs(got - want)
    if single_launch and shape[1] >= 2:
        assert int(diff.max()) == 0, "one-launch FAST UnsharpMask must be bit-identical (max %d)" % int(diff.max())
    limit = int(np.ceil(1.0 + gain))
    level = 65535.0 * threshold
    on_the_edge = np.abs(2 * np.abs(px.astype(np.int64) - blurred) - level) <= 2.0
    assert int(diff[~on_the_edge].max(initial=0)) <= limit, (int(diff[~on_the_edge].max()), limit)
    if diff.size >= 2000:
        assert float((diff == 0).mean()) > 0.97


def test_unsharp_mask_fast_four_plain_channels_and_fallbacks(im, refmod):
    """Four channels without alpha take the fused pass too; RGB (three channels) and the
    MAGICKHIP_NO_FUSED_UNSHARP switch take the three-kernel form: same contract."""
    import os
    px = make_pixels(90, 75, 4, Q16, seed=8)
    want4 = np.concatenate([refmod.RefImage(px[:, :, :3].copy()).unsharp(0.0, 3.0, 1.5, 0.01).numpy(),
                            refmod.RefImage(px[:, :, 3].copy()).unsharp(0.0, 3.0, 1.5, 0.01).numpy().reshape(90, 75, 1)],
                           axis=2).astype(np.int64)
    rgb = px[:, :, :3].copy()
    want3 = refmod.RefImage(rgb).unsharp(0.0, 3.0, 1.5, 0.01).numpy().astype(np.int64)
    want_blend = refmod.RefImage(px).unsharp(0.0, 3.0, 1.5, 0.01).numpy().astype(np.int64)
    im.set_precision(im.PRECISION_FAST)
    try:
        got4 = im.unsharp_mask_image(im.Image(to_device(px), has_alpha=False), 0.0, 3.0, 1.5, 0.01).numpy()
        got3 = im.unsharp_mask_image(im.Image(to_device(rgb)), 0.0, 3.0, 1.5, 0.01).numpy()
        with im.option("MAGICKHIP_NO_FUSED_UNSHARP"):
            unfused = im.unsharp_mask_image(im.Image(to_device(px)), 0.0, 3.0, 1.5, 0.01).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    for name, got, want in (("plain4", got4, want4), ("rgb", got3, want3), ("unfused", unfused, want_blend)):
        diff = np.abs(got.astype(np.int64) - want)
        assert float((diff <= 3).mean()) > 0.999, name
        assert float((diff == 0).mean()) > 0.9, name


def test_blur_fast_matrix_and_vector_paths_agree_within_one_level(im, refmod):
    """MAGICKHIP_NO_MFMA=1 selects the f32 vector kernels: both FAST implementations honour the
    same +-1 contract against the reference (they need not agree with each other exactly)."""
    import os
    px = make_pixels(97, 203, 4, Q16, seed=9)
    dev, ref = run_pair(im, refmod, px)
    want = ref.blur(0.0, 4.0).numpy()
    im.set_precision(im.PRECISION_FAST)
    try:
        matrix = im.blur_image(dev, 0.0, 4.0).numpy()
        with im.option("MAGICKHIP_NO_MFMA"):
            vector = im.blur_image(dev, 0.0, 4.0).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(matrix, want, False, "matrix-core FAST")
    assert_parity(vector, want, False, "vector FAST")


def test_blur_channel_mask_copies_unselected_channels(im, refmod):
    px = make_pixels(40, 52, 4, Q16)
    ref = refmod.RefImage(px).set_channel_mask("RG")
    want = ref.blur(0.0, 2.0).numpy()
    dev = im.Image(to_device(px), copy_channels=(2, 3))
    assert_parity(im.blur_image(dev, 0.0, 2.0).numpy(), want, True, "blur -channel RG")


def test_blur_host_memory_path(im, refmod):
    """MH_MEMORY_HOST: the library stages the pixel-cache block itself."""
    px = make_pixels(90, 111, 4, Q16)
    want = refmod.RefImage(px).blur(0.0, 3.0).numpy()
    got = im.blur_image(im.Image(px.copy()), 0.0, 3.0).numpy()
    assert_parity(got, want, True, "host-memory blur")


# ------------------------------------------------- ConvolveImage / Morphology
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("kernel", ["Gaussian:0x1.5", "3x3: 1,2,1 2,4,2 1,2,1",
                                    "Sobel", "Laplacian:1", "5x1: 1,2,3,2,1", "1x5: 1,2,3,2,1",
                                    "3x3+0+0: 1,-,1 -,1,- 1,nan,1", "1x3: 1,-,2",
                                    "DoG:0x2,1", "Binomial:2"])
def test_convolve(im, refmod, dtype, kernel):
    px = make_pixels(57, 83, 4, dtype)
    dev, ref = run_pair(im, refmod, px)
    assert_parity(im.convolve_image(dev, kernel).numpy(), ref.convolve(kernel).numpy(), True,
                  "convolve " + kernel)


@pytest.mark.parametrize("channels", [1, 2, 3])
def test_convolve_channel_layouts(im, refmod, channels):
    px = make_pixels(40, 50, channels, Q16)
    dev, ref = run_pair(im, refmod, px)
    k = "Gaussian:0x1"
    assert_parity(im.convolve_image(dev, k).numpy(), ref.convolve(k).numpy(), True, "convolve C=%d" % channels)


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("method,kernel,iterations", [
    ("Dilate", "Disk:5", 1), ("Erode", "Disk:5", 1), ("Dilate", "Disk:15", 1),
    ("Erode", "Octagon:3", 2), ("Dilate", "Rectangle:5x3+1+1", 1), ("Erode", "Plus:2", 3),
    ("Open", "Disk:2.5", 1), ("Close", "Diamond:2", 1), ("Smooth", "Square:1", 1),
    ("ErodeIntensity", "Disk:3", 1), ("DilateIntensity", "Disk:3", 1),
    ("Correlate", "3x3: 1,2,3 4,5,6 7,8,9", 1), ("Convolve", "3x3: 1,2,3 4,5,6 7,8,9", 1),
    ("Dilate", "Ring:2,4", 1), ("IterativeDistance", "Chebyshev:1,100", 2),
])
def test_morphology(im, refmod, dtype, method, kernel, iterations):
    px = make_pixels(71, 90, 4, dtype)
    dev, ref = run_pair(im, refmod, px)
    got = im.morphology_image(dev, method, iterations, kernel).numpy()
    want = ref.morphology(method, iterations, kernel).numpy()
    assert_parity(got, want, True, "%s %s x%d" % (method, kernel, iterations))


@pytest.mark.parametrize("channels", [4, 2])
@pytest.mark.parametrize("method,kernel", [
    ("Dilate", "Disk:15"), ("Erode", "Disk:15"), ("Dilate", "Disk:7.3"), ("Erode", "Octagon:6"),
    ("Dilate", "Diamond:9"), ("Erode", "Square:4"), ("Dilate", "Rectangle:9x5+2+1"), ("Dilate", "Plus:11"),
    ("Erode", "Rectangle:1x9"), ("Dilate", "Rectangle:13x1"), ("Dilate", "Disk:31"),
])
def test_symmetric_convex_kernels_over_several_tiles(im, refmod, method, kernel, channels, options):
    """Erode / Dilate with a kernel that is a union of centred rectangles (morph_rects_kernel:
    column windows from the staged tile, row windows across lanes) on a frame that spans several
    workgroup tiles in both directions, ragged at the right and bottom edges; bit-identical to
    the reference and to the plane-per-width kernel it replaces."""
    import bench
    px = make_pixels(131, 277, channels, Q16, seed=len(kernel) + channels)
    dev, ref = run_pair(im, refmod, px)
    holder = {}
    launched = set(bench.kernel_profile(
        im, lambda: holder.update(out=im.morphology_image(dev, method, 1, kernel)), 1))
    assert launched == {"morph_rects"}, launched
    want = ref.morphology(method, 1, kernel).numpy()
    assert_parity(holder["out"].numpy(), want, True, "%s %s c%d" % (method, kernel, channels))
    options.set("MAGICKHIP_NO_RECTS", "1")
    assert_parity(im.morphology_image(dev, method, 1, kernel).numpy(), want, True, "%s %s (planes)" % (method, kernel))


@pytest.mark.parametrize("rows", [83, 258])
@pytest.mark.parametrize("what", ["gaussian 0x1.5", "gaussian 0x4", "gaussian 2x3", "sharpen 0x2", "Disk:5", "Octagon:3",
                                  "3x3: 1,2,3 4,5,6 7,8,9", "5x3+0+2: 1,2,3,4,5 0,1,0,1,0 -1,2,-1,2,-1", "LoG:0x2",
                                  "Gaussian:0x2.5", "7x7+5+1: " + ",".join(str((i * 7) % 11 - 3) for i in range(49))])
def test_gray_convolve_2d_as_four_row_bands(im, refmod, what, rows, options):
    """2-D Convolve (GaussianBlurImage, SharpenImage, ConvolveImage, -morphology Convolve) of a one-channel Q16
    frame as four row bands = four plain channels of a frame a quarter as tall (operators.cpp primitive()), through
    whichever wide-pixel form takes the kernel: separated passes, integer cells on the matrix cores, fused fp64
    sums.  Kernels with their origin off the middle row, cells of both signs.  EXACT bit-identical, FAST within one
    level (morphology.c:2892-2979); the result of the frame's own form besides."""
    import bench
    options.set("MAGICKHIP_GRAY_BANDS_MIN_PIXELS", "0")
    px = make_pixels(rows, 96, 1, Q16, seed=len(what) + rows)
    px[0] = 65535
    px[-1] = 0
    dev, ref = run_pair(im, refmod, px)
    if what.startswith("gaussian"):
        radius, sigma = (float(v) for v in what.split()[1].split("x"))
        call, want = (lambda: im.gaussian_blur_image(dev, radius, sigma)), ref.gaussian_blur(radius, sigma).numpy()
        reach = (im.optimal_kernel_width_2d(radius, sigma) - 1) // 2 if hasattr(im, "optimal_kernel_width_2d") else None
    elif what.startswith("sharpen"):
        call, want = (lambda: im.sharpen_image(dev, 0.0, 2.0)), ref.sharpen(0.0, 2.0).numpy()
        reach = None
    else:
        call, want = (lambda: im.morphology_image(dev, "Convolve", 1, what)), ref.morphology("Convolve", 1, what).numpy()
        reach = None
    holder = {}
    for precision, exact in ((im.PRECISION_EXACT, True), (im.PRECISION_FAST, False)):
        im.set_precision(precision)
        try:
            launched = set(bench.kernel_profile(im, lambda: holder.update(out=call()), 1))
        finally:
            im.set_precision(im.PRECISION_EXACT)
        if rows == 258:
            assert {"gray_bands_pack", "gray_bands_unpack"} <= launched, launched
        assert_parity(holder["out"].numpy(), want, exact, "gray %s, precision %d, %d rows: %s" % (what, precision, rows, sorted(launched)))
    options.set("MAGICKHIP_NO_GRAY_BANDS", "1")
    assert_parity(call().numpy(), want, True, "gray %s (the frame's own form)" % what)


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("method,kernel,iterations", [
    ("Dilate", "Disk:15", 1), ("Erode", "Disk:15", 1), ("Dilate", "Disk:7.3", 1), ("Erode", "Octagon:6", 2),
    ("Dilate", "Diamond:9", 1), ("Erode", "Square:4", 1), ("Dilate", "Rectangle:9x5+2+1", 1), ("Dilate", "Plus:11", 1),
    ("Erode", "Rectangle:1x9", 1), ("Dilate", "Square:1", 3), ("Open", "Disk:5", 1), ("Close", "Octagon:3", 1),
    ("Smooth", "Square:2", 1), ("EdgeIn", "Diamond:2", 1), ("TopHat", "Disk:4", 1), ("Dilate", "Square:2", -1),
])
def test_rgb_erode_dilate_with_a_fourth_empty_channel(im, refmod, dtype, method, kernel, iterations, options):
    """Erode / Dilate (and the compound methods built on them) on a three-channel frame — RGB without alpha, what
    most photographs are, 6- or 12-byte pixels: padded to four channels for the union-of-rectangles kernel
    (morphology.hip try_rects_rgb_padded; the fourth channel is empty and dropped again, the change count that
    ends an unbounded iteration is taken on the way, morphology.c:3195-3199).  Bit-identical to the reference
    and to the form it replaces; values below zero and beyond the Quantum range on float frames."""
    import bench
    options.set("MAGICKHIP_RGB_PAD_MIN_PIXELS", "0")
    options.set("MAGICKHIP_RGB_PAD_FLOAT_ALWAYS", "1")   # (float frames take the form from Disk:8 on by themselves)
    if dtype is Q16:
        px = make_pixels(131, 277, 3, Q16, seed=len(kernel))
    else:
        px = (np.random.default_rng(len(kernel)).random((131, 277, 3)) * 90000.0 - 12000.0).astype(np.float32)
    if iterations < 0:
        px[:] = 0
        px[60, 100] = (40000, 3, 65535)
        px[3, 270, 1] = 1234
    dev, ref = run_pair(im, refmod, px)
    holder = {}
    launched = set(bench.kernel_profile(
        im, lambda: holder.update(out=im.morphology_image(dev, method, iterations, kernel)), 1))
    assert {"rgb_pad", "morph_rects", "rgb_unpad"} <= launched and "morph_convex" not in launched, launched
    want = ref.morphology(method, iterations, kernel).numpy()
    assert_parity(holder["out"].numpy(), want, True, "RGB %s %s x%d" % (method, kernel, iterations))
    options.set("MAGICKHIP_NO_RGB_PAD", "1")
    assert_parity(im.morphology_image(dev, method, iterations, kernel).numpy(), want, True,
                  "RGB %s %s (the frame's own form)" % (method, kernel))


def test_rgb_padded_form_routing(im, options):
    """With nothing set: Q16 RGB takes the padded form from 64 Kpixel on; float RGB only where morph_convex's tile
    does not fit (Disk:8 and wider) — its small kernels are faster on their own."""
    import bench
    import torch
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randint(-32768, 32768, (300, 300, 3), generator=g, device="cuda", dtype=torch.int16).view(torch.uint16)
    f = torch.rand((300, 300, 3), generator=g, device="cuda", dtype=torch.float32) * 65535.0
    small = q[:100, :100].contiguous()
    run = lambda px, kernel: set(bench.kernel_profile(im, lambda: im.morphology_image(im.Image(px), "Dilate", 1, kernel), 1))
    assert "rgb_pad" in run(q, "Disk:5") and "rgb_pad" in run(q, "Disk:15")
    assert "rgb_pad" not in run(small, "Disk:5")
    assert "rgb_pad" not in run(f, "Disk:5") and "rgb_pad" not in run(f, "Disk:7")
    assert "rgb_pad" in run(f, "Disk:8") and "rgb_pad" in run(f, "Disk:15")


@pytest.mark.parametrize("dtype", [Q16, HDRI])
def test_rgb_padded_erode_with_a_channel_mask(im, refmod, dtype, options):
    """... with channels that keep their source value (-channel RB): they come back bit for bit and are not
    counted as changed."""
    options.set("MAGICKHIP_RGB_PAD_MIN_PIXELS", "0")
    options.set("MAGICKHIP_RGB_PAD_FLOAT_ALWAYS", "1")
    px = make_pixels(90, 140, 3, dtype, seed=9)
    dev = im.Image(to_device(px), copy_channels=(1,))
    ref = refmod.RefImage(px).set_channel_mask("RB")
    for method, kernel, iterations in (("Erode", "Disk:6", 1), ("Dilate", "Octagon:2", -1)):
        assert_parity(im.morphology_image(dev, method, iterations, kernel).numpy(),
                      ref.morphology(method, iterations, kernel).numpy(), True, "%s %s -channel RB" % (method, kernel))


@pytest.mark.parametrize("rows", [90, 131, 258])
@pytest.mark.parametrize("method,kernel,iterations", [
    ("Dilate", "Disk:15", 1), ("Erode", "Disk:15", 1), ("Dilate", "Disk:7.3", 1), ("Erode", "Octagon:6", 2),
    ("Dilate", "Diamond:9", 1), ("Erode", "Square:4", 1), ("Dilate", "Rectangle:9x5+2+1", 1), ("Erode", "Rectangle:9x5+2+3", 1),
    ("Dilate", "Plus:11", 1), ("Erode", "Rectangle:1x9", 1), ("Dilate", "Square:1", 3),
    ("Open", "Disk:5", 1), ("Close", "Octagon:3", 1), ("Smooth", "Square:2", 1), ("Edge", "Diamond:2", 1),
    ("TopHat", "Disk:4", 1), ("Dilate", "Square:2", -1),
])
def test_gray_erode_dilate_as_four_row_bands(im, refmod, method, kernel, iterations, rows, options):
    """Erode / Dilate (and the compound methods built on them) on a one-channel Q16 frame — the masks these
    operators are mostly run on: the frame's rows as four bands = the four channels of a frame a quarter as tall,
    through the union-of-rectangles kernel (morphology.hip try_rects_gray_bands).  Heights that are and are not
    multiples of four, kernels whose origin is not their middle row, bands barely taller than the kernel's reach
    (the form declines below that), an unbounded iteration that ends on the change count taken while unpacking
    (morphology.c:3199, :3634-4077).  Bit-identical to the reference and to the form it replaces."""
    import bench
    options.set("MAGICKHIP_GRAY_BANDS_MIN_PIXELS", "0")
    px = make_pixels(rows, 277, 1, Q16, seed=len(kernel) + rows)
    px[0] = 65535
    px[-1] = 0
    if iterations < 0:
        px[:] = 0
        px[rows // 2, 100] = 40000
        px[3, 270] = 65535
    dev, ref = run_pair(im, refmod, px)
    holder = {}
    launched = set(bench.kernel_profile(
        im, lambda: holder.update(out=im.morphology_image(dev, method, iterations, kernel)), 1))
    want = ref.morphology(method, iterations, kernel).numpy()
    assert_parity(holder["out"].numpy(), want, True, "gray %s %s x%d, %d rows" % (method, kernel, iterations, rows))
    reach = {"Disk:15": 15, "Disk:7.3": 7, "Octagon:6": 6, "Diamond:9": 9, "Square:4": 4, "Rectangle:9x5+2+1": 3,
             "Rectangle:9x5+2+3": 3, "Plus:11": 11, "Rectangle:1x9": 4, "Square:1": 1, "Disk:5": 5, "Octagon:3": 3,
             "Square:2": 2, "Diamond:2": 2, "Disk:4": 4}[kernel]
    if (rows + 3) // 4 >= 2 * reach:
        assert {"gray_bands_pack", "morph_rects", "gray_bands_unpack"} <= launched, launched
        assert "morph_convex" not in launched, launched
    else:
        assert "gray_bands_pack" not in launched, launched
    options.set("MAGICKHIP_NO_GRAY_BANDS", "1")
    assert_parity(im.morphology_image(dev, method, iterations, kernel).numpy(), want, True,
                  "gray %s %s (the frame's own form)" % (method, kernel))


@pytest.mark.parametrize("channels", [4, 2, 1])
@pytest.mark.parametrize("method,kernel", [
    ("Dilate", "Disk:15"), ("Erode", "Disk:15"), ("Dilate", "Disk:7.3"), ("Erode", "Octagon:6"),
    ("Dilate", "Diamond:9"), ("Erode", "Square:4"), ("Dilate", "Rectangle:9x5+2+1"), ("Dilate", "Plus:11"),
    ("Erode", "Rectangle:1x9"), ("Dilate", "Rectangle:13x1"),
])
def test_symmetric_convex_kernels_float_quantum(im, refmod, method, kernel, channels, options):
    """The union-of-rectangles kernel on float Quantum (the reference's default build is HDRI; the
    generic 2-D kernel took 102 ms for Dilate Disk:15 on 16384^2): one float channel per 32-bit
    word, v_max_f32 / v_min_f32, RGBA as one column per lane.  Values beyond the Quantum range
    and below zero (Dilate starts from 0, Erode from the pixel itself: morphology.c:2895-2912),
    frames ragged against the tiles.  Bit-identical to the reference and to the generic kernel."""
    import bench
    rng = np.random.default_rng(len(kernel) + channels)
    px = (rng.random((131, 277, channels)) * 90000.0 - 12000.0).astype(np.float32)
    dev, ref = run_pair(im, refmod, px)
    holder = {}
    launched = set(bench.kernel_profile(
        im, lambda: holder.update(out=im.morphology_image(dev, method, 1, kernel)), 1))
    assert launched == {"morph_rects"}, launched
    want = ref.morphology(method, 1, kernel).numpy()
    got = holder["out"].numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "%s %s c%d: %d samples differ" % (
        method, kernel, channels, int((got.view(np.uint32) != want.view(np.uint32)).sum()))
    options.set("MAGICKHIP_NO_FLOAT_RECTS", "1")
    generic = im.morphology_image(dev, method, 1, kernel).numpy()
    assert np.array_equal(generic.view(np.uint32), want.view(np.uint32))


def test_float_rects_channel_mask_change_count_and_nan(im, refmod):
    """Float Quantum through morph_rects_kernel's general epilogue: channels without the update
    trait, the `changed` count that ends an unbounded iteration (|difference| >= MagickEpsilon,
    morphology.c:3195), NaN samples (a NaN neighbour is skipped, a NaN centre survives an Erode)."""
    rng = np.random.default_rng(5)
    px = (rng.random((90, 140, 4)) * 65535.0).astype(np.float32)
    dev = im.Image(to_device(px), copy_channels=(1, 3))
    ref = refmod.RefImage(px).set_channel_mask("RB")
    assert_parity(im.morphology_image(dev, "Erode", 1, "Disk:6").numpy(), ref.morphology("Erode", 1, "Disk:6").numpy(),
                  True, "float Erode Disk:6 -channel RB")
    sparse = np.zeros((60, 70, 4), dtype=np.float32)
    sparse[30, 35] = 1234.5
    dev, ref = run_pair(im, refmod, sparse)
    assert_parity(im.morphology_image(dev, "Dilate", -1, "Square:2").numpy(),
                  ref.morphology("Dilate", -1, "Square:2").numpy(), True, "float Dilate Square:2 until stable")
    holes = px.copy()
    holes[rng.random(holes.shape) < 0.01] = np.nan
    dev, ref = run_pair(im, refmod, holes)
    for method in ("Dilate", "Erode"):
        got = im.morphology_image(dev, method, 1, "Disk:4").numpy()
        want = ref.morphology(method, 1, "Disk:4").numpy()
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "%s with NaN samples: %d differ" % (method, int((~same).sum()))


@pytest.mark.parametrize("channels,cuts", [(4, 1), (4, 2), (2, 1), (4, None)])
@pytest.mark.parametrize("method,kernel", [
    ("Dilate", "Disk:15"), ("Erode", "Disk:15"), ("Erode", "Octagon:6"), ("Dilate", "Square:3"),
    ("Dilate", "Rectangle:9x5+2+1"), ("Erode", "Diamond:11"), ("Dilate", "Rectangle:1x9"), ("Erode", "Rectangle:13x1"),
])
def test_symmetric_convex_kernels_down_a_strip(im, refmod, method, kernel, channels, cuts, options):
    """The same union-of-rectangles evaluation as a walk down 256-column strips (morph_strips_kernel,
    opt-in with MAGICKHIP_STRIPS=1: four columns per lane, a ring of rows in LDS that
    global_load_lds_dwordx4 refills while the tile is evaluated) on a frame of three ragged strips
    by twelve ragged steps; walks of twelve, six and one step (MAGICKHIP_STRIP_CUTS).
    Bit-identical to the reference and to the tile kernel."""
    import bench
    options.set("MAGICKHIP_STRIPS", "1")
    if cuts is not None:
        options.set("MAGICKHIP_STRIP_CUTS", str(cuts))
    px = make_pixels(271, 530, channels, Q16, seed=len(kernel) + channels)
    dev, ref = run_pair(im, refmod, px)
    holder = {}
    launched = set(bench.kernel_profile(
        im, lambda: holder.update(out=im.morphology_image(dev, method, 1, kernel)), 1))
    assert launched == {"morph_rects"}, launched
    got = holder["out"].numpy()
    options.set("MAGICKHIP_STRIPS", None)
    tiles = im.morphology_image(dev, method, 1, kernel).numpy()
    assert np.array_equal(got, tiles), "%s %s c%d: strip walk != tile kernel at %s" % (
        method, kernel, channels, np.argwhere(got != tiles)[:4].tolist())
    if cuts in (1, None):
        assert_parity(got, ref.morphology(method, 1, kernel).numpy(), True, "%s %s c%d" % (method, kernel, channels))


def test_strip_walk_channel_mask_and_change_count(im, refmod, options):
    """morph_strips_kernel's general epilogue: channels without the update trait, the `changed`
    count that ends an unbounded iteration, and a kernel whose origin is off centre."""
    options.set("MAGICKHIP_STRIPS", "1")
    options.set("MAGICKHIP_STRIP_CUTS", "2")
    px = make_pixels(200, 470, 4, Q16, seed=78)
    dev = im.Image(to_device(px), copy_channels=(1, 3))
    ref = refmod.RefImage(px).set_channel_mask("RB")
    got = im.morphology_image(dev, "Erode", 1, "Disk:6").numpy()
    assert_parity(got, ref.morphology("Erode", 1, "Disk:6").numpy(), True, "Erode Disk:6 -channel RB")
    sparse = np.zeros((200, 460, 4), dtype=np.uint16)
    sparse[100, 230] = 65535
    sparse[199, 459] = 40000
    dev, ref = run_pair(im, refmod, sparse)
    assert_parity(im.morphology_image(dev, "Dilate", 6, "Square:2").numpy(),
                  ref.morphology("Dilate", 6, "Square:2").numpy(), True, "Dilate Square:2 x6")
    assert_parity(im.morphology_image(dev, "Dilate", 2, "Rectangle:7x5+1+3").numpy(),
                  ref.morphology("Dilate", 2, "Rectangle:7x5+1+3").numpy(), True, "Dilate Rectangle:7x5+1+3 x2")


def test_symmetric_convex_kernel_channel_mask_and_change_count(im, refmod):
    """Channels without the update trait keep the source value; an unbounded iteration count
    stops on the `changed` count of the kernel (morphology.c:3180-3196, :3892-3905)."""
    px = make_pixels(90, 140, 4, Q16, seed=77)
    dev = im.Image(to_device(px), copy_channels=(1, 3))
    ref = refmod.RefImage(px).set_channel_mask("RB")
    got = im.morphology_image(dev, "Dilate", 1, "Disk:6").numpy()
    assert_parity(got, ref.morphology("Dilate", 1, "Disk:6").numpy(), True, "Dilate Disk:6 -channel RB")
    sparse = np.zeros((60, 70, 4), dtype=np.uint16)
    sparse[30, 35] = 65535
    dev, ref = run_pair(im, refmod, sparse)
    assert_parity(im.morphology_image(dev, "Dilate", -1, "Square:2").numpy(),
                  ref.morphology("Dilate", -1, "Square:2").numpy(), True, "Dilate Square:2 until stable")


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("channels", [1, 3, 4])
@pytest.mark.parametrize("method,kernel,iterations", [
    ("EdgeIn", "Disk:2.5", 1), ("EdgeOut", "Octagon:2", 1), ("Edge", "Diamond:2", 1),
    ("TopHat", "Disk:3", 1), ("BottomHat", "Rectangle:5x3+1+1", 1), ("Edge", "Disk:2", 2),
    ("TopHat", "Square:1", 3),
])
def test_compound_morphology_with_difference(im, refmod, dtype, channels, method, kernel, iterations):
    """MorphologyApply compound methods whose last step is CompositeImage(Difference)
    (morphology.c:3986-4013), composited on the device."""
    px = make_pixels(53, 67, channels, dtype)
    dev, ref = run_pair(im, refmod, px)
    got = im.morphology_image(dev, method, iterations, kernel).numpy()
    want = ref.morphology(method, iterations, kernel).numpy()
    assert_parity(got, want, True, "%s %s x%d c%d" % (method, kernel, iterations, channels))


@pytest.mark.parametrize("method,kernel,iterations", [
    ("HitAndMiss", "Corners", 1), ("HitAndMiss", "LineEnds", 1), ("HitAndMiss", "Corners", 2),
    ("Thinning", "Skeleton", 3), ("Thinning", "Skeleton", -1), ("Thicken", "ConvexHull", 2),
    ("Convolve", "Sobel:>", 1), ("Dilate", "3x3: 0,1,0 1,1,1 0,1,0 ; 3x1: 1,1,1", 1),
])
def test_morphology_kernel_lists(im, refmod, method, kernel, iterations):
    """Kernel lists: re-iteration (Thinning/Thicken/Convolve/Dilate) and the Lighten union of
    the HitAndMiss results (morphology.c:4016-4052)."""
    px = make_pixels(48, 64, 3, Q16, kind="binary")
    dev, ref = run_pair(im, refmod, px)
    got = im.morphology_image(dev, method, iterations, kernel).numpy()
    want = ref.morphology(method, iterations, kernel).numpy()
    assert_parity(got, want, True, "%s %s x%d" % (method, kernel, iterations))


@pytest.mark.parametrize("method,kernel,compose", [
    ("Convolve", "Sobel:>", "Lighten"),            # compass list: the strongest response per pixel
    ("Convolve", "3x3: 0,1,0 1,2,1 0,1,0;3x3: 1,0,1 0,2,0 1,0,1", "Difference"),
    ("HitAndMiss", "LineEnds", "None"),            # the union replaced by re-iteration
    ("Dilate", "Plus:1;Square:1", "Lighten"),
    ("Erode", "Diamond:2;Ring:1,2", "None"),
    ("Convolve", "Sobel:>", "Plus"),               # morphology.c:772: the sum of the directional responses
    ("Erode", "Diamond:2;Ring:1,2", "Darken"),     # intersection
    ("Convolve", "3x3: 0,1,0 1,2,1 0,1,0;3x3: 1,0,1 0,2,0 1,0,1", "Multiply"),
    ("Dilate", "Plus:1;Square:1", "Screen"),
    ("Convolve", "Sobel:>", "Exclusion"),
    ("Convolve", "3x3: 0,1,0 1,2,1 0,1,0;3x3: 1,0,1 0,2,0 1,0,1", "MinusSrc"),
    ("Convolve", "3x3: 0,1,0 1,2,1 0,1,0;3x3: 1,0,1 0,2,0 1,0,1", "MinusDst"),
    ("Dilate", "Plus:1;Square:1", "LinearDodge"),
    ("Erode", "Diamond:2;Ring:1,2", "Over"),
    ("Convolve", "Sobel:>", "DstOver"),
])
def test_morphology_compose_override(im, refmod, method, kernel, compose):
    """The user's `-define morphology:compose=` (morphology.c:4206-4215, :3779-3782) for the
    operators the backend composes with; any other operator is declined."""
    px = make_pixels(56, 72, 4, Q16, seed=len(kernel))
    dev, ref = run_pair(im, refmod, px)
    scale = (1.0, 1) if method == "Convolve" else None
    got = im.morphology_image(dev, method, 1, kernel, scale=scale, compose=compose).numpy()
    ref.set_artifact("morphology:compose", compose)
    if scale is not None:
        ref.set_artifact("convolve:scale", "!")
    want = ref.morphology(method, 1, kernel).numpy()
    assert_parity(got, want, True, "%s %s compose %s" % (method, kernel, compose))
    with pytest.raises(im.MagickHipError):
        im.morphology_image(dev, method, 1, kernel, scale=scale, compose="Overlay")


@pytest.mark.parametrize("compose", ["Over", "DstOver", "Screen", "Lighten"])
def test_morphology_compose_with_alpha_outside_the_channel_mask(im, refmod, compose):
    """`-channel RGB -define morphology:compose=...`: the alpha channel carries the Copy trait.
    CompositeOverImage still writes the merged alpha (composite.c:1096-1104); the general
    operators leave a copy channel alone (composite.c:2580-2590)."""
    px = make_pixels(44, 60, 4, Q16, seed=7)
    ref = refmod.RefImage(px).set_channel_mask("RGB")
    ref.set_artifact("morphology:compose", compose)
    want = ref.morphology("Dilate", 1, "Plus:1;Square:1").numpy()
    dev = im.Image(to_device(px), channel_mask=0x7, copy_channels=(3,))
    got = im.morphology_image(dev, "Dilate", 1, "Plus:1;Square:1", compose=compose).numpy()
    assert_parity(got, want, True, "Dilate list compose %s, -channel RGB" % compose)


def test_hit_and_miss_union_with_alpha(im, refmod):
    px = make_pixels(40, 52, 4, Q16)
    dev, ref = run_pair(im, refmod, px)
    got = im.morphology_image(dev, "HitAndMiss", 1, "Corners").numpy()
    want = ref.morphology("HitAndMiss", 1, "Corners").numpy()
    assert_parity(got, want, True, "HitAndMiss Corners RGBA")


@pytest.mark.parametrize("method", ["HitAndMiss", "Thinning", "Thicken"])
def test_hit_and_miss_family(im, refmod, method):
    px = make_pixels(48, 64, 3, Q16, kind="binary")
    dev, ref = run_pair(im, refmod, px)
    kernel = "3x3: 0,1,- 0,1,1 -,1,-"
    got = im.morphology_image(dev, method, 2, kernel).numpy()
    want = ref.morphology(method, 2, kernel).numpy()
    assert_parity(got, want, True, method)


def test_morphology_until_convergence(im, refmod):
    px = make_pixels(40, 40, 1, Q16, kind="binary")
    dev, ref = run_pair(im, refmod, px)
    got = im.morphology_image(dev, "Dilate", -1, "Plus:1").numpy()
    want = ref.morphology("Dilate", -1, "Plus:1").numpy()
    assert_parity(got, want, True, "dilate until no change")


# ----------------------------------------------------------- WaveletDenoiseImage
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("channels", [1, 2, 3, 4])
@pytest.mark.parametrize("args", [(5000.0, 0.0), (9000.0, 0.4), (300.0, 1.0)])
def test_wavelet_denoise(im, refmod, dtype, channels, args):
    px = make_pixels(45, 70, channels, dtype)
    dev, ref = run_pair(im, refmod, px)
    got = im.wavelet_denoise_image(dev, *args).numpy()
    assert_parity(got, ref.wavelet_denoise(*args).numpy(), True, "wavelet denoise %s c%d" % (args, channels))


# ----------------------------------------------------------- DespeckleImage
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("channels", [1, 2, 3, 4])
@pytest.mark.parametrize("kind", ["random", "smooth"])
def test_despeckle(im, refmod, dtype, channels, kind):
    px = make_pixels(58, 77, channels, dtype, kind=kind)
    dev, ref = run_pair(im, refmod, px)
    assert_parity(im.despeckle_image(dev).numpy(), ref.despeckle().numpy(), True, "despeckle %s c%d" % (kind, channels))


# ----------------------------------------------------------- LocalContrastImage
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("channels", [1, 2, 3, 4])
@pytest.mark.parametrize("args", [(60.0, 40.0), (30.0, -25.0), (110.0, 100.0)])
def test_local_contrast(im, refmod, dtype, channels, args):
    px = make_pixels(67, 91, channels, dtype)
    px[5:9, 7:11, :] = 0                                  # zero luma: the gain divides by it
    dev, ref = run_pair(im, refmod, px)
    got = im.local_contrast_image(dev, *args).numpy()
    want = ref.local_contrast(*args).numpy()
    if dtype == HDRI:                                      # NaN where 0/0; compare the rest bit for bit
        assert np.array_equal(np.isnan(got), np.isnan(want))
        got, want = np.nan_to_num(got), np.nan_to_num(want)
    assert_parity(got, want, True, "local contrast %s c%d" % (args, channels))


# ----------------------------------------------------------- RotationalBlurImage
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("channels", [1, 2, 3, 4])
@pytest.mark.parametrize("angle", [12.0, -40.0, 3.0, 200.0])
def test_rotational_blur(im, refmod, dtype, channels, angle):
    px = make_pixels(53, 71, channels, dtype)
    dev, ref = run_pair(im, refmod, px)
    got = im.rotational_blur_image(dev, angle).numpy()
    assert_parity(got, ref.rotational_blur(angle).numpy(), True, "rotational blur %g c%d" % (angle, channels))


# ----------------------------------------------------------- MotionBlurImage
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("channels", [1, 3, 4])
@pytest.mark.parametrize("args", [(0.0, 3.0, 30.0), (0.0, 1.5, -110.0), (4.0, 2.0, 90.0), (0.0, 6.0, 200.0)])
def test_motion_blur(im, refmod, dtype, channels, args):
    px = make_pixels(47, 61, channels, dtype)
    dev, ref = run_pair(im, refmod, px)
    got = im.motion_blur_image(dev, *args).numpy()
    assert_parity(got, ref.motion_blur(*args).numpy(), True, "motion blur %s c%d" % (args, channels))


# ----------------------------------------------------------- the other ConvolveImage callers
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("channels", [3, 4])
def test_gaussian_blur_sharpen_edge_emboss(im, refmod, dtype, channels):
    """GaussianBlurImage, SharpenImage, EdgeImage, EmbossImage (effect.c): host-built kernels,
    one Convolve pass; Emboss also equalizes its result."""
    px = make_pixels(57, 73, channels, dtype, kind="smooth")
    dev, ref = run_pair(im, refmod, px)
    for name, got, want in (
            ("gaussian 0x1.5", im.gaussian_blur_image(dev, 0.0, 1.5), ref.gaussian_blur(0.0, 1.5)),
            ("gaussian 2x3", im.gaussian_blur_image(dev, 2.0, 3.0), ref.gaussian_blur(2.0, 3.0)),
            ("sharpen 0x1", im.sharpen_image(dev, 0.0, 1.0), ref.sharpen(0.0, 1.0)),
            ("sharpen 3x0.8", im.sharpen_image(dev, 3.0, 0.8), ref.sharpen(3.0, 0.8)),
            ("edge 0", im.edge_image(dev, 0.0), ref.edge(0.0)),
            ("edge 2", im.edge_image(dev, 2.0), ref.edge(2.0)),
            ("emboss 0x1", im.emboss_image(dev, 0.0, 1.0), ref.emboss(0.0, 1.0)),
            ("emboss 2x0.7", im.emboss_image(dev, 2.0, 0.7), ref.emboss(2.0, 0.7))):
        assert_parity(got.numpy(), want.numpy(), True, "%s c%d" % (name, channels))


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("channels,alpha", [(4, True), (4, False), (3, False), (2, True), (1, False)])
@pytest.mark.parametrize("kernel", ["Gaussian:0x2", "Gaussian:0x3.7", "Gaussian:4x1.1", "Square:3", "Rectangle:7x5+1+3",
                                    "5x5: 1,2,3,2,1 2,4,6,4,2 3,6,9,6,3 2,4,6,4,2 1,2,3,2,1",
                                    # cells of both signs (a smoothed derivative), and outer products
                                    # with an odd origin cell (SharpenImage / EdgeImage shapes)
                                    "5x5: -1,-2,0,2,1 -4,-8,0,8,4 -6,-12,0,12,6 -4,-8,0,8,4 -1,-2,0,2,1",
                                    "5x5: -1,-2,-3,-2,-1 -2,-4,-6,-4,-2 -3,-6,100,-6,-3 -2,-4,-6,-4,-2 -1,-2,-3,-2,-1",
                                    "5x5: -1,-1,-1,-1,-1 -1,-1,-1,-1,-1 -1,-1,24.5,-1,-1 -1,-1,-1,-1,-1 -1,-1,-1,-1,-1"])
@pytest.mark.parametrize("folded", [True, False])
def test_separable_2d_convolve_exact(im, refmod, dtype, channels, alpha, kernel, folded, options):
    """EXACT 2-D Convolve with a kernel that is an outer product (GaussianBlurImage's kernels,
    boxes, column x row products): two fp64 1-D passes over alpha-premultiplied doubles and a tie
    check, the undecided samples recomputed in the reference's w x h order
    (convolve_separable.hip) — bit-identical on Q16 and on float Quantum, every layout, small and
    zero alpha, frames ragged against the kernels' tiles.  folded: the premultiplication inside
    the row pass and the tie check inside the column pass (two launches + the queue of undecided
    samples) or round 3's four launches."""
    import bench
    options.set("MAGICKHIP_NO_EXACT_2D", "1")     # (Q16 boxes and integer kernels otherwise take convolve2d_exact.hip)
    if not folded:
        options.set("MAGICKHIP_NO_SEPARABLE_FOLD", "1")
    rng = np.random.default_rng(len(kernel) + channels)
    px = make_pixels(83, 141, channels, dtype, seed=len(kernel))
    if alpha:
        px[10:30, 20:60, channels - 1] = rng.integers(0, 4, (20, 40)).astype(px.dtype)
        px[40:50, 100:130, channels - 1] = 0
    dev = im.Image(to_device(px), has_alpha=alpha)
    # normalised as `-define convolve:scale='!'` does — except the zero-sum derivative kernel
    zero_sum = kernel.startswith("5x5: -1,-2,0")
    scale = None if zero_sum else (1.0, 1)

    def reference(pixels):
        r = refmod.RefImage(pixels)
        if not zero_sum:
            r = r.set_artifact("convolve:scale", "!")
        return r.morphology("Convolve", 1, kernel).numpy()
    if alpha or channels in (1, 3):
        want = reference(px)
    else:
        want = np.concatenate([reference(px[:, :, c].copy()).reshape(83, 141, 1) for c in range(channels)], axis=2)
    holder = {}
    launched = set(bench.kernel_profile(
        im, lambda: holder.update(out=im.morphology_image(dev, "Convolve", 1, kernel, scale=scale)), 1))
    # (alpha-weighted frames under cells that cancel keep the generic kernel: sum(k*alpha) is all
    # cancellation and PerceptibleReciprocal's clamp decides every pixel)
    cancelling = zero_sum or "24.5" in kernel            # |sum of cells| <= 5 % of sum |cell|
    assert (("separable_column_finish" if folded else "separable_finish") in launched) == (not (cancelling and alpha)), launched
    if folded and not (cancelling and alpha):
        assert launched == {"separable_row_sums", "separable_column_finish", "separable_settle"}, launched
    got = holder["out"].numpy()
    if dtype == HDRI:
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "%s c%d: %d float samples differ" % (kernel, channels, int((~same).sum()))
    else:
        assert_parity(got, want, True, "separable %s c%d alpha=%s" % (kernel, channels, alpha))


@pytest.mark.parametrize("queue", ["queue", "overflow", "four launches"])
@pytest.mark.parametrize("dtype", [Q16, HDRI])
def test_separable_2d_convolve_exact_on_ties(im, refmod, dtype, queue, options):
    """A checkerboard of two adjacent levels under an even box puts every value exactly on a
    rounding tie (Q16) or on the midpoint of two floats (HDRI): all of them go through the
    reference-order recomputation; GaussianBlurImage of the same frame lands 1e-5 beside the ties.
    Both bit-identical.  queue: the folded column pass hands them all to separable_settle_kernel;
    overflow: a queue of 1000 entries, the rest settled inside the column pass."""
    options.set("MAGICKHIP_NO_EXACT_2D", "1")
    if queue == "overflow":
        options.set("MAGICKHIP_SEPARABLE_QUEUE", "1000")
    elif queue == "four launches":
        options.set("MAGICKHIP_NO_SEPARABLE_FOLD", "1")
    rows, cols = 70, 110
    y, x = np.mgrid[0:rows, 0:cols]
    px = np.empty((rows, cols, 4), dtype=dtype)
    if dtype == HDRI:
        for c, level in enumerate((1000.25, 32767.5, 3.0e-3)):
            low = np.float32(level)
            px[:, :, c] = np.where(((x + y) & 1) == 1, np.nextafter(low, np.float32(np.inf)), low)
    else:
        for c, level in enumerate((1000, 32767, 65534)):
            px[:, :, c] = level + ((x + y) & 1)
    px[:, :, 3] = 65535
    dev, ref = run_pair(im, refmod, px)
    bits = np.uint32 if dtype == HDRI else np.uint16
    # an 8 x 4 box: 16 cells of 1/32 on either colour of the board — the real value IS the tie
    im._lib.load().MhSeparableRecomputed(1)
    got = im.morphology_image(dev, "Convolve", 1, "Rectangle:8x4", scale=(1.0, 1)).numpy()
    recomputed = im._lib.load().MhSeparableRecomputed(0)
    assert recomputed > rows * cols, recomputed
    want = ref.set_artifact("convolve:scale", "!").morphology("Convolve", 1, "Rectangle:8x4").numpy()
    assert np.array_equal(got.view(bits), want.view(bits))
    # a Gaussian's alternating sum is 1e-5, not 0: near the ties, decided without recomputation
    ref = refmod.RefImage(px)
    got = im.gaussian_blur_image(dev, 0.0, 2.0).numpy()
    assert np.array_equal(got.view(bits), ref.gaussian_blur(0.0, 2.0).numpy().view(bits))


INTEGER_KERNELS = ["Disk:15", "Disk:7.3", "Octagon:5", "Diamond:4", "Plus:3", "Ring:10,14", "Square:3",
                   "Rectangle:8x4", "Rectangle:49x5+3+1", "Rectangle:65x3+40+1",
                   "41x5+30+1: " + " ".join(",".join(str((7 * x + 3 * y) % 5) for x in range(41)) for y in range(5)),
                   "7x5: 1,2,3,4,3,2,1 2,4,6,8,6,4,2 3,6,9,13,9,6,2 2,4,6,8,6,4,2 1,2,3,4,3,2,1",
                   "6x6+1+4: 1,0,2,nan,1,3 0,1,1,2,nan,1 2,2,0,1,1,1 nan,1,3,1,0,2 1,1,1,1,2,0 3,0,1,2,1,1",
                   # cells 2s and 3s (the unit is half the smallest cell), 0.25 steps, and both signs
                   "5x5: 2,3,2,3,2 3,2,3,2,3 2,3,2,3,2 3,2,3,2,3 2,3,2,3,2",
                   "5x5: -1,-2,0,2,1 -4,-8,0,8,4 -6,-12,0,12,6 -4,-8,0,8,4 -1,-2,0,2,1",
                   "5x5: -1,-1,-1,-1,-1 -1,-1,-1,-1,-1 -1,-1,24.5,-1,-1 -1,-1,-1,-1,-1 -1,-1,-1,-1,-1"]


@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("layout", ["rgba", "plain4", "rgb"])
@pytest.mark.parametrize("kernel", INTEGER_KERNELS)
def test_convolve_2d_integer_cells_on_matrix_cores(im, refmod, kernel, layout, mode, options):
    """2-D Convolve whose cells are integer multiples of a unit (flat shapes, integer and
    half-integer user kernels, NaN holes, origins off centre, the widest window the band holds):
    exact integer sums on the i8 matrix cores + tie check (convolve2d_exact.hip) — BIT-IDENTICAL in
    both precision modes, on RGBA with alpha-weighted colour (tiny and zero alpha included), four
    plain channels and RGB, frames ragged against the 64-column strips and 32-row steps, one
    workgroup walking its whole strip through the ring of rows."""
    import bench
    options.set("MAGICKHIP_CONV2D_CUTS", "1")
    channels = 3 if layout == "rgb" else 4
    alpha = layout == "rgba"
    rows, cols = 107, 150
    px = make_pixels(rows, cols, channels, Q16, seed=len(kernel) + channels)
    if alpha:
        px[10:30, 20:60, 3] = np.random.default_rng(3).integers(0, 4, (20, 40))       # tiny alpha
        px[40:50, 100:140, 3] = 0                                                         # transparent
        px[60:100, 5:50, 3] = 65535                                                       # opaque
    signed = "-1" in kernel
    zero_sum = kernel.startswith("5x5: -1,-2,0")
    scale = None if zero_sum else (1.0, 1)
    dev = im.Image(to_device(px), has_alpha=alpha)

    def reference(pixels):
        r = refmod.RefImage(pixels)
        if not zero_sum:
            r = r.set_artifact("convolve:scale", "!")
        return r.morphology("Convolve", 1, kernel).numpy()
    if alpha or channels == 3:
        want = reference(px)
    else:
        want = np.concatenate([reference(px[:, :, c].copy()).reshape(rows, cols, 1) for c in range(4)], axis=2)
    holder = {}
    im.set_precision(im.PRECISION_FAST if mode == "fast" else im.PRECISION_EXACT)
    try:
        launched = set(bench.kernel_profile(
            im, lambda: holder.update(out=im.morphology_image(dev, "Convolve", 1, kernel, scale=scale)), 1))
    finally:
        im.set_precision(im.PRECISION_EXACT)
    outer_product = zero_sum or kernel.startswith(("Square", "Rectangle"))
    width = im.kernel_to_numpy(kernel)[0].shape[1]
    # FAST gives narrow kernels to the f16 kernel, whose band is one chunk up to 17 columns: alpha-
    # weighted frames up to there, four plain channels up to 9, RGB never (operators.cpp)
    f16_first = (width <= 17) if alpha else (layout == "plain4" and width <= 9)
    if not (signed and alpha) and (mode == "exact" or not (f16_first or outer_product)):
        # (alpha-weighted sums of signed cells keep the fp64 kernels; FAST separates an outer product first)
        assert launched == {"conv2d_exact"}, launched
    elif mode == "fast" and f16_first and not outer_product and not signed:
        assert launched == {"conv2d_mfma"}, launched
    assert_parity(holder["out"].numpy(), want, mode == "exact" or "conv2d_exact" in launched,
                  "integer 2-D convolve %s %s %s" % (kernel, layout, mode))


@pytest.mark.parametrize("layout", ["rgba", "plain4", "rgb"])
def test_convolve_2d_integer_cells_on_ties(im, refmod, layout):
    """A checkerboard of two adjacent levels under an 8 x 4 box (16 cells on either colour): the
    real value of every sample IS the rounding tie, so all of them go through the reference-order
    recomputation of convolve2d_exact.hip; bit-identical.  A Disk on the same frame has an odd
    cell count: never within 1/(2*count) of a tie, nothing is recomputed."""
    rows, cols = 70, 110
    channels = 3 if layout == "rgb" else 4
    y, x = np.mgrid[0:rows, 0:cols]
    px = np.empty((rows, cols, channels), dtype=Q16)
    for c, level in enumerate((1000, 32767, 65534)):
        px[:, :, c] = level + ((x + y) & 1)
    if channels == 4:
        px[:, :, 3] = 65535 if layout == "rgba" else 7 + 2 * ((x + y) & 1)
    dev = im.Image(to_device(px), has_alpha=layout == "rgba")

    def reference(kernel):
        if layout == "plain4":
            return np.concatenate([refmod.RefImage(px[:, :, c].copy()).set_artifact("convolve:scale", "!")
                                   .morphology("Convolve", 1, kernel).numpy().reshape(rows, cols, 1) for c in range(4)], axis=2)
        return refmod.RefImage(px).set_artifact("convolve:scale", "!").morphology("Convolve", 1, kernel).numpy()
    lib = im._lib.load()
    lib.MhConvolve2DRecomputed(1)
    got = im.morphology_image(dev, "Convolve", 1, "Rectangle:8x4", scale=(1.0, 1)).numpy()
    recomputed = lib.MhConvolve2DRecomputed(1)
    assert recomputed >= rows * cols * 3 // 2, recomputed
    assert np.array_equal(got, reference("Rectangle:8x4"))
    got = im.morphology_image(dev, "Convolve", 1, "Disk:4.3", scale=(1.0, 1)).numpy()
    recomputed = lib.MhConvolve2DRecomputed(0)
    assert recomputed == 0, recomputed
    assert np.array_equal(got, reference("Disk:4.3"))


TIE_KERNELS = ["Disk:7.3", "LoG:0x1.4", "DoG:0,1.2,2.5", "Comet:0x2+30",
               "7x5+2+1: 0.11,0.52,0.73,0.14,0.95,0.36,0.27 0.2,nan,0.6,0.8,0.6,nan,0.2 0.31,0.62,0.93,1.3,0.9,0.6,0.2 "
               "0.2,0.4,0.6,0.8,0.6,0.4,0.2 0.1,0.2,0.3,0.4,0.3,0.2,0.1",
               "5x5: 0.5,-1.25,0,2.5,1 -4,-8.5,0,8,4 -6,-12,0.75,12,7 -4,-8,0,8.25,4 -1,-2,0,2,1"]


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("channels,alpha", [(4, True), (4, False), (3, False), (2, True), (1, False)])
@pytest.mark.parametrize("kernel", TIE_KERNELS)
def test_convolve_2d_fused_fp64_with_tie_check(im, refmod, dtype, channels, alpha, kernel, options):
    """What is left of 2-D Convolve after the outer-product and integer-cell paths — cells of any
    value and sign, NaN holes, off-centre origins, one- and two-channel layouts, float frames with
    fractional levels and values beyond the Quantum range: one fused multiply-add per cell and channel
    over alpha-premultiplied doubles and a tie check (convolve2d_tie.hip), bit-identical on Q16 and
    on float Quantum; tiny and zero alpha included."""
    import bench
    options.set("MAGICKHIP_NO_EXACT_2D", "1")         # (Disk on Q16 would take the integer kernel)
    rng = np.random.default_rng(len(kernel) + channels)
    px = make_pixels(83, 141, channels, dtype, seed=len(kernel) + 5)
    if alpha:
        px[10:30, 20:60, channels - 1] = rng.integers(0, 4, (20, 40)).astype(px.dtype)
        px[40:50, 100:130, channels - 1] = 0
    signed = kernel.startswith(("LoG", "DoG", "5x5"))
    scale = None if signed else (1.0, 1)
    dev = im.Image(to_device(px), has_alpha=alpha)

    def reference(pixels):
        r = refmod.RefImage(pixels)
        if not signed:
            r = r.set_artifact("convolve:scale", "!")
        return r.morphology("Convolve", 1, kernel).numpy()
    if alpha or channels in (1, 3):
        want = reference(px)
    else:
        want = np.concatenate([reference(px[:, :, c].copy()).reshape(83, 141, 1) for c in range(channels)], axis=2)
    holder = {}
    launched = set(bench.kernel_profile(
        im, lambda: holder.update(out=im.morphology_image(dev, "Convolve", 1, kernel, scale=scale)), 1))
    # (alpha-weighted frames under cells that nearly cancel keep the generic kernel; Comet is one row:
    # the 1-D kernels)
    if not kernel.startswith("Comet") and not (alpha and signed):
        assert launched == {"conv2d_tie"}, launched
    got = holder["out"].numpy()
    if dtype == HDRI:
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "%s c%d: %d float samples differ" % (kernel, channels, int((~same).sum()))
    else:
        assert_parity(got, want, True, "fused 2-D convolve %s c%d alpha=%s" % (kernel, channels, alpha))


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("channels,alpha", [(4, True), (4, False), (1, False)])
def test_tall_column_kernel_with_nan_cells(im, refmod, dtype, channels, alpha):
    """A 1 x 31 column kernel with NaN cells is the reference's width == 1 fast path, whose gamma
    carries height / count when cells are missing (morphology.c:2775-2776): 25 cells and more must
    not reach convolve2d_tie.hip, alpha-weighted and plain, Q16 and float."""
    cells = ["nan" if i in (3, 4, 17, 29) else "%.3f" % (0.2 + 0.05 * ((7 * i) % 11)) for i in range(31)]
    kernel = "1x31+0+12: " + ",".join(cells)
    px = make_pixels(77, 91, channels, dtype, seed=77 + channels)
    if alpha:
        px[20:40, 10:50, channels - 1] = 0
    dev = im.Image(to_device(px), has_alpha=alpha)
    ref = refmod.RefImage(px) if (alpha or channels == 1) else None
    if ref is not None:
        want = ref.morphology("Convolve", 1, kernel).numpy()
    else:
        want = np.concatenate([refmod.RefImage(px[:, :, c].copy()).morphology("Convolve", 1, kernel).numpy()
                               .reshape(77, 91, 1) for c in range(channels)], axis=2)
    got = im.morphology_image(dev, "Convolve", 1, kernel).numpy()
    if dtype == HDRI:
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), "1x31 NaN column, c%d: %d float samples differ" % (channels, int((~same).sum()))
    else:
        assert_parity(got, want, True, "1x31 column kernel with NaN cells c%d alpha=%s" % (channels, alpha))


@pytest.mark.parametrize("dtype", [Q16, HDRI])
def test_convolve_2d_fused_fp64_on_ties_and_non_finite_samples(im, refmod, dtype, options):
    """A checkerboard of two adjacent levels (adjacent floats) under a kernel whose two colours of
    cells weigh the same puts every value on a rounding tie: all of them are recomputed in the
    reference's order; a float frame with an infinity and a NaN under a kernel with NaN cells (where
    a zero cannot stand in for "no cell") recomputes the tiles that hold them.  Bit-identical."""
    options.set("MAGICKHIP_NO_EXACT_2D", "1")
    options.set("MAGICKHIP_NO_SEPARABLE_EXACT", "1")
    rows, cols = 70, 110
    y, x = np.mgrid[0:rows, 0:cols]
    px = np.empty((rows, cols, 4), dtype=dtype)
    if dtype == HDRI:
        for c, level in enumerate((1000.25, 32767.5, 3.0e-3)):
            low = np.float32(level)
            px[:, :, c] = np.where(((x + y) & 1) == 1, np.nextafter(low, np.float32(np.inf)), low)
    else:
        for c, level in enumerate((1000, 32767, 65534)):
            px[:, :, c] = level + ((x + y) & 1)
    px[:, :, 3] = 65535
    dev, ref = run_pair(im, refmod, px)
    bits = np.uint32 if dtype == HDRI else np.uint16
    lib = im._lib.load()
    lib.MhConvolve2DTieRecomputed(1)
    got = im.morphology_image(dev, "Convolve", 1, "Rectangle:8x4", scale=(1.0, 1)).numpy()
    recomputed = lib.MhConvolve2DTieRecomputed(0)
    assert recomputed > rows * cols, recomputed
    want = ref.set_artifact("convolve:scale", "!").morphology("Convolve", 1, "Rectangle:8x4").numpy()
    assert np.array_equal(got.view(bits), want.view(bits))
    if dtype == HDRI:
        px = make_pixels(90, 150, 4, HDRI, seed=9)
        px[20, 30, 1] = np.inf
        px[70, 120, 3] = np.nan
        px[50, 10, 0] = -np.inf
        dev, ref = run_pair(im, refmod, px)
        got = im.morphology_image(dev, "Convolve", 1, "Disk:4.3", scale=(1.0, 1)).numpy()
        want = ref.set_artifact("convolve:scale", "!").morphology("Convolve", 1, "Disk:4.3").numpy()
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), int((~same).sum())


# 41 x 5 integer cells that are no outer product: three 32-slot chunks per band
WIDE_INTEGER_KERNEL = "41x5+30+1: " + " ".join(",".join(str((7 * x + 3 * y) % 5) for x in range(41)) for y in range(5))


@pytest.mark.parametrize("layout", ["rgba", "plain4", "rgb"])
@pytest.mark.parametrize("kernel", ["Disk:15", "Octagon:5", "Ring:10,14", WIDE_INTEGER_KERNEL,
                                    "6x6+1+4: 1,0,2,nan,1,3 0,1,1,2,nan,1 2,2,0,1,1,1 nan,1,3,1,0,2 1,1,1,1,2,0 3,0,1,2,1,1",
                                    "5x5: -1,-2,0,2,1 -4,-8,0,8,4 -6,-12,0,12,7 -4,-8,0,8,4 -1,-2,0,2,1"])
def test_convolve_2d_integer_cells_float_quantum(im, refmod, kernel, layout, options):
    """A float-Quantum frame whose samples are integers of 0..65535 (what an 8- or 16-bit file
    decodes to in the reference's default HDRI build) under a kernel with integer-multiple cells:
    the same exact integer sums on the i8 matrix cores, the results rounded to float with the tie
    check at float-rounding midpoints — bit-identical to the reference's w x h walk; the generic
    kernel is launched behind it and leaves at once."""
    import bench
    options.set("MAGICKHIP_CONV2D_CUTS", "1")
    channels = 3 if layout == "rgb" else 4
    alpha = layout == "rgba"
    rows, cols = 107, 150
    q = make_pixels(rows, cols, channels, Q16, seed=len(kernel) + channels)
    if alpha:
        q[10:30, 20:60, 3] = np.random.default_rng(3).integers(0, 4, (20, 40))
        q[40:50, 100:140, 3] = 0
        q[60:100, 5:50, 3] = 65535
    px = q.astype(np.float32)
    signed = "-1" in kernel
    scale = None if signed else (1.0, 1)
    dev = im.Image(to_device(px), has_alpha=alpha)

    def reference(pixels):
        r = refmod.RefImage(pixels)
        if not signed:
            r = r.set_artifact("convolve:scale", "!")
        return r.morphology("Convolve", 1, kernel).numpy()
    if alpha or channels == 3:
        want = reference(px)
    else:
        want = np.concatenate([reference(px[:, :, c].copy()).reshape(rows, cols, 1) for c in range(4)], axis=2)
    holder = {}
    launched = bench.kernel_profile(
        im, lambda: holder.update(out=im.morphology_image(dev, "Convolve", 1, kernel, scale=scale)), 1)
    if not (signed and alpha):
        # (the fused fp64 kernel stands behind it for frames that turn out not to be integers)
        assert set(launched) == {"conv2d_exact", "conv2d_tie"}, launched
    got = holder["out"].numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "%d float samples differ" % int(
        (got.view(np.uint32) != want.view(np.uint32)).sum())


@pytest.mark.parametrize("spoiler", [1000.25, -3.0, 65536.0, 1.0e9, float("nan"), float("inf")])
def test_convolve_2d_integer_cells_float_quantum_falls_back(im, refmod, spoiler):
    """One sample that is not an integer of 0..65535 anywhere in the frame — a fraction, a negative
    value, one beyond the range, NaN, inf — and the generic kernel behind the integer one does the
    whole frame: bit-identical (NaN where the reference has NaN)."""
    rows, cols = 150, 131
    px = make_pixels(rows, cols, 4, Q16, seed=77).astype(np.float32)
    px[140, 120, 1] = spoiler
    dev, ref = run_pair(im, refmod, px)
    got = im.morphology_image(dev, "Convolve", 1, "Disk:6.3", scale=(1.0, 1)).numpy()
    want = ref.set_artifact("convolve:scale", "!").morphology("Convolve", 1, "Disk:6.3").numpy()
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    assert same.all(), int((~same).sum())


@pytest.mark.parametrize("channels", [4, 3])
def test_c5_convolve_disk15_exact_full_rows(im, refmod, channels):
    """ConvolveMorphology Disk:15 (SURVEY 8d's MAC-bound variant of C5) bit-identical on a frame
    wide enough for several strips per XCD and tall enough for the ring to wrap more than once
    (704 x 1100; RGBA with random alpha: 32-row steps; RGB: 64-row steps, a wave on both column
    tiles): the reference takes a few seconds."""
    px = make_pixels(1100, 704, channels, Q16, seed=15)
    dev, ref = run_pair(im, refmod, px)
    got = im.morphology_image(dev, "Convolve", 1, "Disk:15", scale=(1.0, 1)).numpy()
    want = ref.set_artifact("convolve:scale", "!").morphology("Convolve", 1, "Disk:15").numpy()
    assert np.array_equal(got, want)


# ----------------------------------------------------------- ImportImagePixels / ExportImagePixels
IO_TYPES = ["uint8", "uint16", "uint32", "uint64", "float32", "float64"]


@pytest.mark.parametrize("tag", ["q16", "hdri"])
@pytest.mark.parametrize("kind", IO_TYPES)
def test_import_export_pixels_against_reference_vectors(im, vectors, tag, kind):
    """Device ImportImagePixels / ExportImagePixels against the outputs of the reference's own
    pixel.c loops (committed vectors): all storage types, component orders with pads and alpha
    first, gray+alpha, a sub-region; component buffers in host memory."""
    base, gray = vectors[tag + "_io_base"], vectors[tag + "_io_gray_base"]
    for m in ("RGBA", "BGRA", "RGB", "ARGB", "BGRP", "RAB"):
        data = vectors["%s_import|%s|%s|data" % (tag, kind, m)]
        img = im.Image(to_device(base))
        im.import_image_pixels(img, 5, 3, m, data)
        assert_parity(img.numpy(), vectors["%s_import|%s|%s" % (tag, kind, m)], True, "import %s %s" % (kind, m))
    data = vectors["%s_import|%s|IA|data" % (tag, kind)]
    img = im.Image(to_device(gray), colorspace="gray")
    im.import_image_pixels(img, 5, 3, "IA", data)
    assert_parity(img.numpy(), vectors["%s_import|%s|IA" % (tag, kind)], True, "import %s IA" % kind)
    img = im.Image(to_device(base))
    for m in ("RGBA", "BGRA", "RGB", "ARGB", "BGRP", "RGBP", "I", "IA", "RPPA"):
        want = vectors["%s_export|%s|%s" % (tag, kind, m)]
        got = im.export_image_pixels(img, 4, 2, m, np.zeros_like(want))
        assert np.array_equal(got, want), "export %s %s: %d differ" % (kind, m, int((got != want).sum()))
    want = vectors["%s_export_gray|%s|IA" % (tag, kind)]
    got = im.export_image_pixels(im.Image(to_device(gray), colorspace="gray"), 4, 2, "IA", np.zeros_like(want))
    assert np.array_equal(got, want), "export gray %s IA" % kind


@pytest.mark.parametrize("dtype", [Q16, HDRI])
def test_import_export_device_buffers_round_trip(im, refmod, dtype):
    """Component buffers that already live on the device (decoded scanlines in HBM):
    8-bit BGRA in, blur, 8-bit BGRA out, against the same chain through the reference."""
    import torch
    rng = np.random.default_rng(3)
    frame = rng.integers(0, 256, (90, 120, 4), dtype=np.uint8)
    canvas = np.zeros((90, 120, 4), dtype=dtype)
    ref = refmod.RefImage(canvas).import_pixels(0, 0, "BGRA", frame)
    img = im.Image(to_device(canvas))
    im.import_image_pixels(img, 0, 0, "BGRA", torch.from_numpy(frame).cuda())
    assert_parity(img.numpy(), ref.numpy(), True, "import BGRA from a device buffer")
    blurred = im.blur_image(img, 0.0, 1.5)
    out = torch.zeros((90, 120, 4), dtype=torch.uint8, device="cuda")
    im.export_image_pixels(blurred, 0, 0, "BGRA", out)
    want = ref.blur(0.0, 1.5).export_pixels(0, 0, 120, 90, "BGRA", np.uint8)
    assert np.array_equal(out.cpu().numpy(), want)


def test_import_export_reject_what_the_backend_does_not_take(im):
    img = im.Image(to_device(make_pixels(8, 8, 3, Q16)))
    from imagemagick_amd import MagickHipError
    for call in (lambda: im.import_image_pixels(img, 0, 0, "RGBA", np.zeros((8, 8, 4), np.uint8)),     # no alpha channel
                 lambda: im.import_image_pixels(img, 4, 4, "RGB", np.zeros((8, 8, 3), np.uint8)),      # outside
                 lambda: im.import_image_pixels(img, 0, 0, "CMY", np.zeros((8, 8, 3), np.uint8))):     # CMYK
        with pytest.raises(MagickHipError):
            call()


# ----------------------------------------------------------- ContrastImage / ModulateImage
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("channels", [3, 4])
def test_contrast_and_modulate(im, refmod, dtype, channels):
    """ContrastImage (HSB sine push) and ModulateImage (HSL default, HSB): fp64 as the CPU path;
    the one libm call (sin) may differ in the last bit, which Q16 rounding hides and float
    Quantum may show as 1 ULP."""
    px = make_pixels(61, 83, channels, dtype)
    gray = make_pixels(8, 83, channels, dtype)
    gray[:, :, 1] = gray[:, :, 0]
    gray[:, :, 2] = gray[:, :, 0]                      # zero saturation rows
    px = np.concatenate([px, gray], axis=0)
    for name, run_gpu, run_ref in (
            ("contrast +", lambda i: im.contrast_image(i, True), lambda r: r.contrast(True)),
            ("contrast -", lambda i: im.contrast_image(i, False), lambda r: r.contrast(False)),
            ("modulate 110,80,135", lambda i: im.modulate_image(i, 110.0, 80.0, 135.0),
             lambda r: r.modulate(110.0, 80.0, 135.0)),
            ("modulate 60,150,20", lambda i: im.modulate_image(i, 60.0, 150.0, 20.0),
             lambda r: r.modulate(60.0, 150.0, 20.0)),
            ("modulate 100,100,100", lambda i: im.modulate_image(i), lambda r: r.modulate()),
            ("modulate HSB 120,70,160", lambda i: im.modulate_image(i, 120.0, 70.0, 160.0, "HSB"),
             lambda r: r.modulate(120.0, 70.0, 160.0, "HSB"))):
        dev, ref = run_pair(im, refmod, px)
        run_gpu(dev)
        run_ref(ref)
        assert_parity(dev.numpy(), ref.numpy(), True, "%s c%d" % (name, channels), max_ulp=1)


# ----------------------------------------------------------- UnsharpMaskImage
@pytest.mark.parametrize("dtype", [Q16, HDRI])
def test_unsharp_mask(im, refmod, dtype):
    px = make_pixels(66, 77, 4, dtype)
    dev, ref = run_pair(im, refmod, px)
    got = im.unsharp_mask_image(dev, 0.0, 2.0, 1.0, 0.02).numpy()
    want = ref.unsharp(0.0, 2.0, 1.0, 0.02).numpy()
    assert_parity(got, want, True, "unsharp")


@pytest.mark.parametrize("dtype,channels", [(HDRI, 4), (HDRI, 3), (HDRI, 2), (HDRI, 1), (Q16, 2), (Q16, 1)])
@pytest.mark.parametrize("gain,threshold", [(1.0, 0.02), (2.5, 0.0)])
def test_unsharp_mask_epilogue_in_the_column_pass(im, refmod, dtype, channels, gain, threshold):
    """Float Quantum (and the Q16 layouts the matrix kernel does not take, EXACT): the fp64
    column pass applies UnsharpMaskImage's threshold / gain as it stores (convolve.hip,
    unsharp_on_the_way_out) — two launches, no epilogue kernel, bit-identical."""
    import bench
    px = make_pixels(75, 101, channels, dtype, seed=channels)
    dev, ref = run_pair(im, refmod, px)
    holder = {}
    launched = set(bench.kernel_profile(
        im, lambda: holder.update(out=im.unsharp_mask_image(dev, 0.0, 4.0, gain, threshold)), 1))
    assert launched == {"conv_row", "conv_column"}, launched
    assert_parity(holder["out"].numpy(), ref.unsharp(0.0, 4.0, gain, threshold).numpy(), True,
                  "unsharp c%d gain %g" % (channels, gain))


# ----------------------------------------------------------------- ResizeImage
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("shape,target,filt", [
    ((16, 24, 4), (96, 64), "Lanczos"), ((40, 50, 4), (17, 23), "Lanczos"),
    ((31, 45, 4), (90, 31), "Mitchell"), ((31, 45, 3), (20, 70), "Catrom"),
    ((25, 25, 1), (100, 100), "Triangle"), ((64, 48, 2), (48, 64), "Lanczos"),
    ((30, 30, 4), (30, 90), "Box"), ((200, 150, 4), (20, 15), "Lanczos"),
    ((20, 20, 4), (21, 19), "Point"), ((33, 29, 4), (70, 70), "Gaussian"),
    ((33, 29, 4), (70, 70), "Hann"), ((33, 29, 4), (41, 37), "Spline"),
])
def test_resize(im, refmod, dtype, shape, target, filt):
    px = make_pixels(*shape, dtype)
    dev, ref = run_pair(im, refmod, px)
    cols, rows = target
    got = im.resize_image(dev, cols, rows, filt).numpy()
    want = ref.resize(cols, rows, filt).numpy()
    assert_parity(got, want, True, "resize %s -> %s %s" % (shape, target, filt))


def test_resize_contribution_table_cache(im, refmod):
    """Tables are cached by (filter parameters, sizes): the same geometry with another filter,
    the same filter with another geometry, and repeats (cache hits, enough distinct keys to
    evict) must all still match the reference; a long axis takes the threaded builder."""
    px = make_pixels(30, 5000, 4, Q16)
    dev, ref = run_pair(im, refmod, px)
    plan = [("Lanczos", 9000, 20), ("Mitchell", 9000, 20), ("Lanczos", 9000, 20), ("Lanczos", 2500, 45),
            ("Catrom", 9000, 20), ("Triangle", 9000, 20), ("Hermite", 2500, 45), ("Box", 2500, 45),
            ("Gaussian", 2500, 45), ("Spline", 2500, 45), ("Hann", 2500, 45), ("Lanczos", 9000, 20),
            ("Mitchell", 9000, 20)]
    wanted = {}
    for filt, cols, rows in plan:
        if (filt, cols, rows) not in wanted:
            wanted[(filt, cols, rows)] = ref.resize(cols, rows, filt).numpy()
        got = im.resize_image(dev, cols, rows, filt).numpy()
        assert_parity(got, wanted[(filt, cols, rows)], True, "resize %s %dx%d" % (filt, cols, rows))


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("target", [(320, 240), (33, 21)])
def test_resize_fast_precision(im, refmod, dtype, target):
    """FAST resampling stays in fp64 but contracts multiply-adds (Fma64): +-1 Q16 level /
    1 float ULP by contract, and in practice identical (a flip needs an exact value within
    ~1e-11 of a rounding boundary)."""
    px = make_pixels(60, 80, 4, dtype)
    dev, ref = run_pair(im, refmod, px)
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.resize_image(dev, target[0], target[1], "Lanczos").numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    exact = assert_parity(got, ref.resize(target[0], target[1], "Lanczos").numpy(), False,
                          "resize under FAST", max_ulp=1)
    assert exact > 0.999


@pytest.mark.parametrize("precision", ["exact", "fast"])
@pytest.mark.parametrize("target", [(560, 940), (31, 47), (300, 33)])
@pytest.mark.parametrize("alpha", [True, False])
def test_resize_float_frame_with_non_finite_samples(im, refmod, precision, target, alpha):
    """A float frame with +-Inf and NaN samples: the passes pad their tap lists
    with zero weights, and 0*inf = NaN must not reach an output whose window does not hold the
    sample — the reference only ever multiplies the samples of the window (resize.c:3494-3530).
    The vertical pass skips zero weights (a scalar test), the horizontal pass watches the samples
    it stages and takes a tested tap loop for a tile that holds a non-finite one.  Same NaN / Inf
    pattern as the reference, everything else within the mode's contract; enlargement, reduction
    and a mixed resize."""
    rng = np.random.default_rng(target[0] + (3 if alpha else 0))
    px = make_pixels(140, 235, 4, HDRI, seed=target[1])
    for value in (np.inf, -np.inf, np.nan):
        for _ in range(6):
            y, x, c = int(rng.integers(0, 140)), int(rng.integers(0, 235)), int(rng.integers(0, 4))
            px[y, x, c] = value
    px[70, 100:104, :] = np.inf                       # a run of whole pixels
    dev = im.Image(to_device(px), has_alpha=alpha)
    if alpha:
        want = refmod.RefImage(px).resize(target[0], target[1], "Lanczos").numpy()
    else:
        want = np.concatenate([refmod.RefImage(px[:, :, c].copy()).resize(target[0], target[1], "Lanczos").numpy()
                               .reshape(target[1], target[0], 1) for c in range(4)], axis=2)
    im.set_precision(im.PRECISION_FAST if precision == "fast" else im.PRECISION_EXACT)
    try:
        got = im.resize_image(dev, target[0], target[1], "Lanczos").numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert np.array_equal(np.isnan(got), np.isnan(want)), "NaN pattern differs: %d vs %d" % (
        int(np.isnan(got).sum()), int(np.isnan(want).sum()))
    assert np.array_equal(np.isinf(got), np.isinf(want)) and np.array_equal(got[np.isinf(got)], want[np.isinf(want)])
    finite = np.isfinite(want)
    assert finite.sum() > 0.5 * want.size
    assert_parity(np.where(finite, got, 0.0).astype(np.float32), np.where(finite, want, 0.0).astype(np.float32),
                  precision == "exact", "resize of a frame with non-finite samples", max_ulp=0 if precision == "exact" else 1)


# ------------------------------------------------------------------ colourspace
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("src,dst", [("sRGB", "RGB"), ("RGB", "sRGB"), ("sRGB", "Lab"),
                                     ("Lab", "sRGB"), ("sRGB", "XYZ"), ("XYZ", "sRGB"),
                                     ("RGB", "Lab"), ("Lab", "XYZ")])
@pytest.mark.parametrize("channels", [3, 4])
def test_colorspace(im, refmod, dtype, src, dst, channels):
    px = make_pixels(50, 64, channels, dtype)
    dev, ref = run_pair(im, refmod, px, colorspace=src)
    got = im.transform_image_colorspace(dev, dst).numpy()
    want = ref.colorspace(dst).numpy()
    uses_pow = "Lab" in (src, dst)
    assert_parity(got, want, True, "%s->%s" % (src, dst), max_ulp=1 if uses_pow else 0)


@pytest.mark.parametrize("frame", ["random", "constant", "two_levels", "dark", "odd_size"])
@pytest.mark.parametrize("black,white", [(0.02, 0.01), (0.0, 0.0), (0.7, 0.6), (1.5, 0.0), (0.0, 1.5)])
def test_lab_contrast_stretch_three_launches(im, frame, black, white, options):
    """FAST sRGB->Lab + ContrastStretch in one call on RGBA Q16 takes three launches (convert + bin,
    stretch_levels_kernel, stretch_apply_kernel: pointwise.hip); the levels and the map are those
    of the general route (slab reduction, three LUT kernels, LUT apply), bit for bit — on frames
    whose shares overflow the packed counters' capacity into the extra-pixel lists, on a constant
    frame (one bin holds everything), with thresholds no bin reaches and thresholds every bin
    reaches."""
    import bench
    rng = np.random.default_rng(9)
    rows, cols = (1031, 1021) if frame == "odd_size" else (1024, 1280)
    px = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    if frame == "constant":
        px[:, :, :3] = (30000, 20000, 10000)
    elif frame == "two_levels":
        px[:, :, :3] = np.where((np.arange(cols) % 3 == 0)[None, :, None], 60000, 900)
    elif frame == "dark":
        px[:, :, :3] = rng.integers(0, 40, (rows, cols, 3), dtype=np.uint16)
    n = rows * cols
    results = []
    im.set_precision(im.PRECISION_FAST)
    try:
        for general in (False, True):
            if general:
                options.set("MAGICKHIP_NO_STRETCH_LEVELS", "1")
            img = im.Image(to_device(px))
            launched = bench.kernel_profile(im, lambda: im.transform_colorspace_contrast_stretch_image(
                img, "Lab", black * n, n - white * n), 1)
            assert "colorspace_histogram" in launched, launched
            results.append(img.numpy().copy())
    finally:
        im.set_precision(im.PRECISION_EXACT)
    same = results[0] == results[1]
    assert same.all(), "%s black %g white %g: %d samples differ, first %s: %s vs %s" % (
        frame, black, white, int((~same).sum()), np.argwhere(~same)[0].tolist(),
        results[0][~same][:4].tolist(), results[1][~same][:4].tolist())


def test_page_locked_host_buffers(im, refmod):
    """A host image whose pixels are MhHostAlloc memory (the shim's pixel-cache allocator) goes
    up and comes down with one DMA transfer each, not through the staging threads: same result,
    and the memory is handed back when the array dies."""
    import gc
    px = make_pixels(1100, 1301, 4, Q16, seed=12)             # 11 MB: above the staging threshold
    before = im.host_allocated_bytes()
    pinned = im.host_alloc(px.shape, px.dtype)
    pinned[:] = px
    out = im.host_alloc(px.shape, px.dtype)
    assert im.host_allocated_bytes() == before + 2 * px.nbytes
    result = im.Image(out)
    im.blur_image(im.Image(pinned), 0.0, 2.0, out=result)
    assert_parity(result.pixels, refmod.RefImage(px).blur(0.0, 2.0).numpy(), True, "blur, page-locked buffers")
    plain = im.blur_image(im.Image(px), 0.0, 2.0).numpy()
    assert np.array_equal(plain, out)
    del result, pinned, out
    gc.collect()
    assert im.host_allocated_bytes() == before


# ------------------------------------------------ Equalize / ContrastStretch
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("kind", ["random", "smooth"])
@pytest.mark.parametrize("channels", [1, 2, 3, 4])
def test_equalize(im, refmod, dtype, kind, channels):
    px = make_pixels(48, 64, channels, dtype, kind=kind)
    dev, ref = run_pair(im, refmod, px)
    got = im.equalize_image(dev).numpy()
    want = ref.equalize().numpy()
    assert_parity(got, want, True, "equalize")


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("kind", ["random", "smooth"])
@pytest.mark.parametrize("channels", [1, 3, 4])
def test_contrast_stretch(im, refmod, dtype, kind, channels):
    rows, cols = 48, 64
    px = make_pixels(rows, cols, channels, dtype, kind=kind)
    dev, ref = run_pair(im, refmod, px)
    black = 0.02 * rows * cols
    white = rows * cols - 0.01 * rows * cols     # as `-contrast-stretch 2%x1%` passes them
    got = im.contrast_stretch_image(dev, black, white).numpy()
    want = ref.contrast_stretch(black, white).numpy()
    assert_parity(got, want, True, "contrast-stretch")


@pytest.mark.parametrize("shape", [(1100, 1000), (1024, 1025), (3000, 5600)])
@pytest.mark.parametrize("kind", ["random", "flat", "two-level"])
def test_intensity_histogram_packed_counters(im, refmod, shape, kind, options):
    """Frames above a megapixel bin their intensity in one pass into 16-bit LDS counters, two to a
    word (histogram_packed_kernel); a share of more than 65 535 pixels per workgroup, a frame that
    is ONE level (every pixel of a workgroup in one counter) and an odd pixel count must give the
    same table as the 32-bit half-range kernel and the reference's ContrastStretch / Equalize."""
    rows, cols = shape
    if kind == "random":
        px = make_pixels(rows, cols, 4, Q16, seed=rows)
    elif kind == "flat":
        px = np.full((rows, cols, 4), 31111, dtype=np.uint16)
        px[rows // 2, cols // 3] = (1, 2, 3, 4)
    else:
        px = np.where((np.arange(rows * cols).reshape(rows, cols, 1) % 7) < 3, 65535, 2).astype(np.uint16)
        px = np.ascontiguousarray(np.broadcast_to(px, (rows, cols, 4)))
        px[1, 1] = (9, 8, 7, 6)                  # (an all-gray frame is the CPU path's: SetImageGray)
    n = rows * cols
    want_stretch = refmod.RefImage(px).contrast_stretch(0.02 * n, n - 0.01 * n).numpy()
    want_equal = refmod.RefImage(px).equalize().numpy()
    for packed in (True, False):
        if not packed:
            options.set("MAGICKHIP_NO_PACKED_HISTOGRAM", "1")
        dev = im.Image(to_device(px))
        im.contrast_stretch_image(dev, 0.02 * n, n - 0.01 * n)
        assert_parity(dev.numpy(), want_stretch, True, "contrast stretch %s packed=%s" % (kind, packed))
        dev = im.Image(to_device(px))
        im.equalize_image(dev)
        assert_parity(dev.numpy(), want_equal, True, "equalize %s packed=%s" % (kind, packed))


def test_contrast_stretch_per_channel_mask(im, refmod):
    rows, cols = 40, 40
    px = make_pixels(rows, cols, 4, Q16, kind="smooth")
    ref = refmod.RefImage(px).set_channel_mask("RGB")
    want = ref.contrast_stretch(30.0, rows * cols - 30.0).numpy()
    dev = im.Image(to_device(px), channel_mask=0x7, copy_channels=(3,))
    got = im.contrast_stretch_image(dev, 30.0, rows * cols - 30.0).numpy()
    assert_parity(got, want, True, "contrast-stretch -channel RGB")


@pytest.mark.parametrize("shape", [(64, 64), (37, 41), (1, 1), (1, 2), (3, 1), (5, 7), (130, 259)])
def test_fast_lab_on_rgb_frames(im, refmod, shape):
    """FAST sRGB -> Lab on a three-channel Q16 frame (6-byte pixels: colorspace_lab_fast_rgb_kernel, four pixels =
    three 8-byte words a lane, the last npixels mod 4 on their own): within one level of the reference
    (colorspace.c:1089-1128 / gem.c ConvertRGBToLab) and sample for sample what the RGBA kernel gives."""
    import bench
    rows, cols = shape
    px = make_pixels(rows, cols, 3, Q16, seed=rows * 31 + cols)
    px.reshape(-1, 3)[:4] = ((0, 0, 0), (65535, 65535, 65535), (65535, 0, 0), (1, 2, 3))[: min(4, rows * cols)]
    want = refmod.RefImage(px).colorspace("Lab").numpy()
    rgba = np.concatenate([px, np.full((rows, cols, 1), 4660, dtype=np.uint16)], axis=2)
    im.set_precision(im.PRECISION_FAST)
    try:
        dev = im.Image(to_device(px))
        launched = set(bench.kernel_profile(im, lambda: im.transform_image_colorspace(dev, "Lab"), 1))
        wide = im.Image(to_device(rgba))
        im.transform_image_colorspace(wide, "Lab")
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert launched == {"colorspace"}, launched
    got = dev.numpy()
    assert_parity(got, want, False, "fast Lab, RGB %s" % (shape,))
    assert np.array_equal(got, wide.numpy()[:, :, :3]), "RGB and RGBA FAST Lab differ"
    assert np.array_equal(wide.numpy()[:, :, 3], rgba[:, :, 3])


def test_lab_then_contrast_stretch_chain(im, refmod):
    """BASELINE config C4's per-image pipeline."""
    rows, cols = 64, 64
    px = make_pixels(rows, cols, 4, Q16)
    dev, ref = run_pair(im, refmod, px)
    im.transform_image_colorspace(dev, "Lab")
    im.contrast_stretch_image(dev, 0.02 * rows * cols, rows * cols * 0.99)
    ref.colorspace("Lab").contrast_stretch(0.02 * rows * cols, rows * cols * 0.99)
    assert_parity(dev.numpy(), ref.numpy(), True, "Lab + contrast-stretch")


def test_histogram_matches_numpy(im):
    px = make_pixels(100, 120, 4, Q16, kind="smooth")
    dev = im.Image(to_device(px))
    h = im.histogram(dev, False).cpu().numpy()
    for c in range(4):
        want = np.bincount(px[:, :, c].ravel(), minlength=65536)
        assert np.array_equal(h[:, c], want)


# ------------------------------------------------- full-size property checks
def test_blur_full_size_properties(im):
    """BASELINE C2 geometry (8192x8192 RGBA Q16, sigma=10): properties that need
    no oracle run.  A constant image is a fixed point of the normalised blur;
    blurring is invariant under transposition for a symmetric separable kernel."""
    import torch
    n = 8192
    const = torch.full((n, n, 4), 12345, dtype=torch.int16, device="cuda").view(torch.uint16)
    out = im.blur_image(im.Image(const), 0.0, 10.0).pixels
    assert int((out.view(torch.int16) != 12345).sum()) == 0
    g = torch.Generator(device="cuda").manual_seed(7)
    a = torch.randint(0, 32768, (2048, 1024, 4), generator=g, device="cuda", dtype=torch.int16)
    a[:, :, 3] = -1            # opaque alpha (65535): the two passes then commute exactly enough
    img = im.Image(a.view(torch.uint16).contiguous())
    b1 = im.blur_image(img, 0.0, 3.0).pixels.view(torch.int16)
    at = a.transpose(0, 1).contiguous()
    b2 = im.blur_image(im.Image(at.view(torch.uint16)), 0.0, 3.0).pixels.view(torch.int16)
    diff = (b1.transpose(0, 1).to(torch.int32) - b2.to(torch.int32)).abs().max()
    assert int(diff) <= 1


def test_blur_fast_full_size_against_exact(im):
    """BASELINE C2 at full size, the configuration bench.py times (matrix-core passes, 128 ring
    steps per strip, every strip and segment boundary of the 8192^2 frame), on uniform-random
    RGBA with a band of tiny alpha (0..3 levels) and a fully transparent band.  EXACT is pinned
    bit for bit to the reference by the small-size tests; FAST must satisfy, against EXACT:

    * each pass on its own (the row kernel, and the column kernel fed EXACT's intermediate)
      within +-1 level everywhere, tiny alpha included;
    * the two-pass blur within +-1 level EVERYWHERE: the row pass recomputes an alpha below 8192
      levels exactly (fp64, the reference's order) whenever its f32 sum lies too close to a
      rounding tie to decide the level (mfma_common.hpp), so the weights the column pass sees
      are the reference's own wherever one level matters;
    * a constant image comes back unchanged."""
    import torch
    n = 8192
    g = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randint(-32768, 32768, (n, n, 4), generator=g, device="cuda", dtype=torch.int16)
    a[: n // 8, :, 3] = torch.randint(0, 4, (n // 8, n), generator=g, device="cuda", dtype=torch.int16)   # tiny alpha
    a[n // 8: n // 4, :, 3] = 0                                                                    # transparent band
    img = im.Image(a.view(torch.uint16))

    def levels(image):
        return image.pixels.view(torch.int16).to(torch.int32) & 0xffff

    def worst(x, y):
        d = (x - y).abs()
        return int(d.max()), float((d == 0).double().mean()), d

    row_exact_img = im.convolve_image(img, "Blur:0x10")
    row_exact = levels(row_exact_img)
    col_exact = levels(im.convolve_image(row_exact_img, "Blur:0x10+90"))
    exact = levels(im.blur_image(img, 0.0, 10.0))
    assert torch.equal(exact, col_exact)            # BlurImage is exactly these two passes
    im.set_precision(im.PRECISION_FAST)
    try:
        row_fast = levels(im.convolve_image(img, "Blur:0x10"))
        col_fast = levels(im.convolve_image(row_exact_img, "Blur:0x10+90"))
        fast = levels(im.blur_image(img, 0.0, 10.0))
        const = torch.full((n, n, 4), 23456, dtype=torch.int16, device="cuda").view(torch.uint16)
        same = im.blur_image(im.Image(const), 0.0, 10.0).pixels
    finally:
        im.set_precision(im.PRECISION_EXACT)
    for name, got, want in (("row pass", row_fast, row_exact), ("column pass", col_fast, col_exact)):
        m, same_fraction, _ = worst(got, want)
        assert m <= 1, "%s: max |FAST - EXACT| = %d" % (name, m)
        assert same_fraction > 0.98, name
    m, same_fraction, _ = worst(fast, exact)
    assert m <= 1, "blur: max |FAST - EXACT| = %d" % m           # tiny-alpha band included
    # (the one-launch FAST form keeps its intermediate colour unrounded — convolve_fused_hybrid.hip —
    # and agrees with the reference on ~97 % of the samples; the contract is the +-1 above)
    assert same_fraction > 0.95
    assert int((same.view(torch.int16) != 23456).sum()) == 0


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("kind", ["random", "smooth"])
def test_histogram_operators_large_frame(im, refmod, dtype, kind):
    """Frames of >= 1 Mpixel take the LDS-privatised intensity histogram (two half-range
    passes per workgroup): same counts, same operators."""
    rows, cols = 1040, 1100
    px = make_pixels(rows, cols, 4, dtype, kind=kind)
    dev, ref = run_pair(im, refmod, px)
    h = im.histogram(dev, True).cpu().numpy()
    assert int(h[:, 0].sum()) == rows * cols and np.array_equal(h[:, 0], h[:, 3])
    assert_parity(im.equalize_image(dev).numpy(), ref.equalize().numpy(), True, "equalize (large)")
    dev, ref = run_pair(im, refmod, px)
    n = rows * cols
    got = im.contrast_stretch_image(dev, 0.02 * n, n - 0.01 * n).numpy()
    assert_parity(got, ref.contrast_stretch(0.02 * n, n - 0.01 * n).numpy(), True, "contrast-stretch (large)")


@pytest.mark.parametrize("background", [False, True])
@pytest.mark.parametrize("counts", [True, False])
def test_equalize_float_frame_from_running_counts(im, refmod, background, counts, options):
    """EqualizeImage on a float frame of a megapixel or more evaluates its map per sample from the
    running counts held in LDS (a uint32 every 16 bins + uint16 offsets) instead of gathering from
    a 65536-float table: the table's own three fp64 operations, the same bits.  A flat background
    puts more than 65535 pixels into one 16-bin group: the workgroups fall back to the table."""
    import bench
    rows, cols = 1030, 1100
    px = make_pixels(rows, cols, 4, HDRI, kind="random", seed=5)
    px[:7, :9, :3] = [[-3.0, 70000.0, 0.49]]                 # clamped indices
    if background:
        px[:, :500, :3] = np.float32(12345.25)
    if not counts:
        options.set("MAGICKHIP_NO_EQUALIZE_COUNTS", "1")
    dev, ref = run_pair(im, refmod, px)
    holder = {}
    launched = bench.kernel_profile(im, lambda: holder.update(out=im.equalize_image(dev)), 1)
    assert "apply_lut" in launched, launched
    got, want = holder["out"].numpy(), ref.equalize().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), int((got != want).sum())


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("alpha", [True, False])
@pytest.mark.parametrize("shape,target,filt", [
    ((37, 53, 4), (212, 148), "Lanczos"), ((64, 300, 4), (1200, 256), "Lanczos"),
    ((23, 600, 4), (2400, 92), "Lanczos"), ((90, 100, 4), (200, 180), "Mitchell"),
    ((29, 33, 4), (330, 290), "Lanczos"), ((41, 50, 4), (150, 164), "Triangle"),
    ((1, 1, 4), (40, 40), "Lanczos"), ((150, 70, 4), (141, 600), "Catrom"),
])
def test_resize_fast_one_launch_on_the_matrix_pipe(im, refmod, options, dtype, alpha, shape, target, filt):
    """FAST enlargements of four-channel frames that the vector-pipe form (resize_stream.hip: whole-number
    horizontal factors up to 4) does not take run VerticalFilter and HorizontalFilter as banded
    matrix products on the fp64 matrix pipe in ONE launch (resize_mfma.hip): the Quantum-rounded
    intermediate stays in registers.  Within one level / one float ULP of the reference, and in
    practice identical; partial tiles, strips and row groups, a mixed enlargement, one source pixel.
    (Every geometry is sent here: MAGICKHIP_NO_RESIZE_STREAM.)"""
    import bench
    options.set("MAGICKHIP_NO_RESIZE_STREAM", "1")
    px = make_pixels(shape[0], shape[1], 4, dtype, seed=shape[1] + target[0])
    if alpha:
        px[: shape[0] // 2, :, 3] = 65535                # half opaque
        px[:, : shape[1] // 5, 3] = 0                    # a transparent band
    dev = im.Image(to_device(px), has_alpha=alpha)
    if alpha:
        want = refmod.RefImage(px).resize(target[0], target[1], filt).numpy()
    else:
        want = np.concatenate([refmod.RefImage(px[:, :, c].copy()).resize(target[0], target[1], filt).numpy()
                               .reshape(target[1], target[0], 1) for c in range(4)], axis=2)
    im.set_precision(im.PRECISION_FAST)
    holder = {}
    try:
        launched = set(bench.kernel_profile(
            im, lambda: holder.update(out=im.resize_image(dev, target[0], target[1], filt)), 1))
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert launched == {"resize_mfma"}, launched
    same = assert_parity(holder["out"].numpy(), want, False, "one-launch resize %s -> %s %s" % (shape, target, filt),
                         max_ulp=1)
    assert same > 0.999


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("alpha", [True, False])
@pytest.mark.parametrize("shape,target,filt", [
    ((37, 53, 4), (212, 148), "Lanczos"),        # 4x both ways, one partial strip
    ((64, 300, 4), (1200, 256), "Lanczos"),      # six strips of 58 source columns, the last partial
    ((23, 600, 4), (2400, 92), "Lanczos"),
    ((90, 100, 4), (200, 180), "Mitchell"),      # 2x, five neighbours
    ((41, 50, 4), (150, 164), "Triangle"),       # 3x: a weight of 1e-12 beside a transparent band
    ((150, 61, 4), (244, 701), "Lanczos"),       # 4x across, 4.67x down (seven rows under the window): more than one chunk of rows
    ((20, 116, 4), (464, 90), "Lanczos"),        # exactly two full strips (cut at column 64: three)
    ((12, 58, 4), (232, 54), "Catrom"),
    ((33, 130, 4), (260, 99), "Lanczos"),        # 2x across, 3x down
])
def test_resize_fast_one_launch_on_the_vector_pipe(im, refmod, dtype, alpha, shape, target, filt):
    """FAST enlargements of four-channel frames by a whole-number horizontal factor (2, 3, 4) run both
    filters in ONE launch on the fp64 vector pipe (resize_stream.hip): a lane owns a source column, the
    window rows live in registers, the weights in scalar registers, the Quantum-rounded intermediate
    crosses lanes through the wave's own LDS row.  Within one level / one float ULP of the reference
    and in practice identical; clipped windows at the image edges take their listed weights."""
    import bench
    px = make_pixels(shape[0], shape[1], 4, dtype, seed=shape[1] + target[0])
    if alpha:
        px[: shape[0] // 2, :, 3] = 65535                # half opaque
        px[:, : shape[1] // 5, 3] = 0                    # a transparent band
    dev = im.Image(to_device(px), has_alpha=alpha)
    if alpha:
        want = refmod.RefImage(px).resize(target[0], target[1], filt).numpy()
    else:
        want = np.concatenate([refmod.RefImage(px[:, :, c].copy()).resize(target[0], target[1], filt).numpy()
                               .reshape(target[1], target[0], 1) for c in range(4)], axis=2)
    im.set_precision(im.PRECISION_FAST)
    holder = {}
    try:
        launched = set(bench.kernel_profile(
            im, lambda: holder.update(out=im.resize_image(dev, target[0], target[1], filt)), 1))
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert "resize_stream" in launched and launched <= {"resize_stream", "resize_stream_careful"}, launched
    same = assert_parity(holder["out"].numpy(), want, False, "vector-pipe resize %s -> %s %s" % (shape, target, filt),
                         max_ulp=1)
    assert same > 0.999


def test_resize_vector_pipe_float_frame_with_nan_and_inf(im, refmod, options):
    """Zero-weight padding must not spread a NaN / Inf / huge sample beyond the outputs whose window
    holds it: the items that meet one are recomputed in the reference's own windows."""
    options.set("MAGICKHIP_RESIZE_STREAM_ROWS", "32")
    px = make_pixels(70, 130, 4, HDRI, seed=11)
    px[10, 20, 1] = np.nan
    px[33, 64, 3] = np.inf
    px[50, 100, 0] = -np.inf
    px[60, 5, 2] = 3.0e30
    px[69, 129, 2] = np.nan
    dev = im.Image(to_device(px), has_alpha=True)
    want = refmod.RefImage(px).resize(520, 175, "Lanczos").numpy()
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.resize_image(dev, 520, 175, "Lanczos").numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.isinf(got), np.isinf(want))
    finite = np.isfinite(want)
    assert finite.mean() > 0.7
    g, w = got[finite].astype(np.float64), want[finite].astype(np.float64)
    assert np.all(np.abs(g - w) <= np.spacing(np.abs(want[finite]).astype(np.float32)).astype(np.float64) + 1e-30)


def test_resize_vector_pipe_weights_born_of_cancellation_and_tiny_frames(im, refmod):
    """Triangle at 3x: the middle output of a source column has a tap at distance 1-MagickEpsilon whose
    weight, 1e-12, differs by a part in a thousand from binade to binade of the column index — and
    decides the result where the other tap's pixel is transparent (PerceptibleReciprocal's clamp).  The
    strips are cut at the powers of two and carry their own weights (bit-identical to the table's
    inside a binade); frames of a few columns whose every window is clipped are declined and the
    matrix-pipe form takes them."""
    import bench
    im.set_precision(im.PRECISION_FAST)
    try:
        for shape, target, filt, stream in (((41, 50, 4), (150, 164), "Triangle", True),
                                            ((9, 5, 4), (20, 36), "Lanczos", False),
                                            ((12, 158, 4), (474, 40), "Lanczos", True),
                                            ((7, 3, 4), (6, 7), "Lanczos", False)):
            for dtype in (Q16, HDRI):
                px = make_pixels(shape[0], shape[1], 4, dtype, seed=5)
                px[:, : shape[1] // 5, 3] = 0                # a transparent band
                dev = im.Image(to_device(px), has_alpha=True)
                holder = {}
                launched = set(bench.kernel_profile(
                    im, lambda: holder.update(out=im.resize_image(dev, target[0], target[1], filt)), 1))
                assert ("resize_stream" in launched) == stream, (shape, launched)
                assert_parity(holder["out"].numpy(), refmod.RefImage(px).resize(target[0], target[1], filt).numpy(),
                              False, "vector-pipe form or its fallback %s %s" % (shape, filt), max_ulp=1)
    finally:
        im.set_precision(im.PRECISION_EXACT)


@pytest.mark.parametrize("case", [
    # (rows, columns), (target columns, rows), filter, alpha kind, float frame
    ((139, 70), (39, 60), "Triangle", "tiny", False),          # two passes (a reduction)
    ((56, 105), (63, 4), "Triangle", "tiny", False),
    ((64, 21), (10, 28), "Box", "tiny", False),
    ((12, 182), (78, 5), "Box", "binary", True),
    ((72, 49), (147, 269), "Catrom", "binary", False),         # one launch, vector pipe (3x)
    ((84, 143), (572, 420), "Catrom", "tiny", False),          # ... 4x
    ((60, 90), (180, 150), "Triangle", "tiny", False),         # ... 2x
    ((114, 92), (276, 524), "Lanczos", "binary", False),       # alpha sums that cancel under the window
    ((60, 90), (180, 150), "Box", "tiny", True),
    ((72, 49), (120, 200), "Catrom", "tiny", False),           # one launch, matrix pipe
    ((50, 70), (163, 129), "Triangle", "binary", False),
    ((50, 70), (163, 129), "Hermite", "tiny", True),
    ((60, 90), (270, 180), "Lanczos", "blocks", False),        # 3x: one output in three sits on a source pixel's centre
    ((60, 90), (270, 233), "Lanczos", "blocks", True),
])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_resize_fast_intermediate_on_rounding_boundaries(im, refmod, case, seed):
    """FAST ResizeImage on frames that put the INTERMEDIATE on rounding boundaries: polynomial filters at
    rational positions over small integers produce exact x.5 sums in families, the last bits of the
    summation order decide the level, and one level of a small intermediate alpha is thousands of levels
    of the colours the second filter weights with it (tests/stress_parity.py found differences of up to
    13 783 levels in round 5).  The first filter therefore runs in the reference's own order on the
    two-pass route, the one-launch forms find such values (and alpha sums that cancel to nothing under
    a window) and recompute their rows: within one level / one float ULP on every route."""
    shape, target, filt, kind, is_float = case
    rng = np.random.default_rng(seed * 1000 + shape[0])
    px = rng.integers(0, 65536, (shape[0], shape[1], 4)).astype(np.uint16)
    if kind == "tiny":
        px[:, :, 3] = rng.integers(0, 4, shape)
    elif kind == "blocks":                               # a sprite: opaque rectangles on a transparent ground
        px[:, :, 3] = 0
        for _ in range(6):
            y, x = int(rng.integers(0, shape[0] - 8)), int(rng.integers(0, shape[1] - 8))
            px[y: y + int(rng.integers(3, 25)), x: x + int(rng.integers(3, 25)), 3] = 65535
    else:
        px[:, :, 3] = np.where(rng.random(shape) < 0.5, 0, 65535)
    if is_float:
        px = px.astype(np.float32)
    want = refmod.RefImage(px).resize(target[0], target[1], filt).numpy()
    dev = im.Image(to_device(px), has_alpha=True)
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.resize_image(dev, target[0], target[1], filt).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(got, want, False, "FAST resize %s -> %s %s, %s alpha" % (shape, target, filt, kind), max_ulp=1,
                  residue=65535.0e-9 if is_float else 0.0)


@pytest.mark.parametrize("is_float", [False, True])
@pytest.mark.parametrize("factor,period", [(2, 5), (2, 3), (4, 7), (3, 4), (4, 1)])
def test_resize_fast_transparent_columns_beside_planted_ties(im, refmod, factor, period, is_float):
    """Planted for the ballot fix of round 5 (resize_stream.hip, finish_fast): the intermediate ALPHA of every
    ordinary column sits exactly on x.5 (Triangle at a whole-number enlargement weights two rows with k/2f and
    1-k/2f: rows alternate between a and a + f give (2f*a + f)/(2f) = a + 1/2), and every `period`-th column is
    fully transparent, so the lane that takes PerceptibleReciprocal's branch — the first lane of a strip among
    them, which is where the wave's mark mask is read — sits beside lanes whose value must be marked.  A lost
    mark leaves the fused sums' last bits to choose between alpha levels 1 and 2 (or 3 and 4 ...): half the weight
    of a neighbour in the horizontal filter, thousands of levels of colour (resize.c:3494-3530, :3709-3745)."""
    rows, cols = 90, 700                                 # several strips of source columns per row chunk
    rng = np.random.default_rng(100 * factor + period)
    px = rng.integers(0, 65536, (rows, cols, 4)).astype(np.uint16)
    y = np.arange(rows)[:, None]
    x = np.arange(cols)[None, :]
    px[:, :, 3] = 1 + factor * (y & 1) + 2 * factor * ((x // 3) % 3)
    if period > 1:
        px[:, ::period, 3] = 0
    else:                                                # every strip's first lane transparent: columns 0 mod 8 ... and 1 mod 64
        px[:, (x[0] % 8 == 0) | (x[0] % 64 == 1), 3] = 0
    frame = px.astype(np.float32) if is_float else px
    target = (factor * cols, factor * rows + (7 if factor == 2 else 0))
    want = refmod.RefImage(frame).resize(target[0], target[1], "Triangle").numpy()
    dev = im.Image(to_device(frame), has_alpha=True)
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.resize_image(dev, target[0], target[1], "Triangle").numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(got, want, False, "FAST resize x%d Triangle, transparent columns every %d" % (factor, period),
                  max_ulp=1, residue=65535.0e-9 if is_float else 0.0)


def test_resize_fast_x3_alpha_clamped_by_the_first_filter(im, refmod):
    """A 3x enlargement has one output in three whose window is (~0 .. ~0, 1, ~0 .. ~0).  Where the first filter's
    negative lobes clamp an intermediate alpha to 0 — one pixel in 77 on random alpha — that output's alpha sum is
    1e-8 out of terms of 1e-8: PerceptibleReciprocal's clamp acts (resize.c:3522-3528 via pixel-accessor.h:242-254),
    and until round 6 the walk reported every such lane: three quarters of the frame went down the careful launch
    (12 ms against 3.2 per 8192^2).  The report is withdrawn where the window's own terms say no summation order
    matters (resize_stream.hip, finish_fast) — and kept where it does: a transparent pixel between two EQUAL opaque
    ones, whose terms w and -w cancel and leave the sign of the alpha sum to the order."""
    import bench
    rows, cols = 96, 640
    rng = np.random.default_rng(333)
    random_alpha = rng.integers(0, 65536, (rows, cols, 4)).astype(np.uint16)
    cancelling = rng.integers(0, 65536, (rows, cols, 4)).astype(np.uint16)
    cancelling[:, :, 3] = 65535
    cancelling[:, 1::4, 3] = 0                          # transparent columns between opaque ones
    cancelling[:, 0::4, :3] = cancelling[:, 2::4, :3]   # ... of equal colour either side
    im.set_precision(im.PRECISION_FAST)
    try:
        for label, px, few_blocks in (("random alpha", random_alpha, True), ("cancelling neighbours", cancelling, False)):
            want = refmod.RefImage(px).resize(3 * cols, 3 * rows, "Lanczos").numpy()
            dev = im.Image(to_device(px), has_alpha=True)
            holder = {}
            prof = bench.kernel_profile(im, lambda: holder.update(out=im.resize_image(dev, 3 * cols, 3 * rows, "Lanczos")), 1)
            assert "resize_stream" in prof, prof
            assert_parity(holder["out"].numpy(), want, False, "FAST resize x3 Lanczos, %s" % label, max_ulp=1)
            if few_blocks:
                # (the careful launch's workgroups leave at once where nothing is marked)
                assert prof["resize_stream_careful"]["avg_ms"] < 0.5 * prof["resize_stream"]["avg_ms"], prof
    finally:
        im.set_precision(im.PRECISION_EXACT)


def test_resize_fast_one_launch_forms_wait_for_a_frame_that_fills_the_chip(im, refmod, options):
    """The library's own routing (the suites run with MAGICKHIP_RESIZE_ONE_LAUNCH_MIN_PIXELS=0): a FAST enlargement
    of a small frame keeps the two passes — the one-launch walks were 2-9 times slower there
    (tools/probe_resize_rows.py) — a frame of six megapixels takes the streaming form."""
    import bench
    options.set("MAGICKHIP_RESIZE_ONE_LAUNCH_MIN_PIXELS", None)
    im.set_precision(im.PRECISION_FAST)
    try:
        two_passes = {"resize_horizontal", "resize_vertical"}
        for shape, target, filt, expected in (((300, 400), (800, 600), "Lanczos", two_passes),
                                              ((300, 400), (700, 500), "Lanczos", two_passes),
                                              ((2500, 2500), (5000, 5000), "Lanczos", {"resize_stream", "resize_stream_careful"}),
                                              # quarters as weights: every fourth sum of integer samples on a rounding boundary
                                              ((2500, 2500), (5000, 5000), "Triangle", two_passes),
                                              # thirds: integer sums never reach a half (a float frame's can)
                                              ((2500, 2500), (7500, 7500), "Triangle", {"resize_stream", "resize_stream_careful"})):
            px = make_pixels(shape[0], shape[1], 4, Q16)
            dev = im.Image(to_device(px), has_alpha=True)
            holder = {}
            launched = set(bench.kernel_profile(
                im, lambda: holder.update(out=im.resize_image(dev, target[0], target[1], filt)), 1))
            assert launched == expected, (shape, filt, launched)
            if shape[0] <= 400:
                want = refmod.RefImage(px).resize(target[0], target[1], filt).numpy()
                assert_parity(holder["out"].numpy(), want, False, "FAST resize, default routing %s" % (shape,), max_ulp=1)
    finally:
        im.set_precision(im.PRECISION_EXACT)


def test_resize_fast_falls_back_to_two_passes(im, refmod):
    """What the matrix-pipe form declines keeps the two-pass kernels: reductions, barely-enlarging
    geometries whose windows are wider than its ring, three-channel frames."""
    import bench
    im.set_precision(im.PRECISION_FAST)
    try:
        for shape, target in (((60, 80, 4), (33, 21)), ((40, 40, 4), (41, 43)), ((30, 30, 3), (120, 120))):
            px = make_pixels(shape[0], shape[1], shape[2], Q16)
            dev, ref = run_pair(im, refmod, px)
            holder = {}
            launched = set(bench.kernel_profile(
                im, lambda: holder.update(out=im.resize_image(dev, target[0], target[1], "Lanczos")), 1))
            assert launched == {"resize_horizontal", "resize_vertical"}, launched
            assert_parity(holder["out"].numpy(), ref.resize(target[0], target[1], "Lanczos").numpy(), False,
                          "two-pass resize %s" % (shape,), max_ulp=1)
    finally:
        im.set_precision(im.PRECISION_EXACT)


def test_histogram_large_frame_linear_rgb_intensity(im, refmod):
    """A linear-RGB frame's intensity goes through EncodePixelGamma (pixel.c:2446-2452): the
    packed-table kernel's general form (the whole GetPixelIntensity switch behind a call)."""
    rows, cols = 1030, 1050
    px = make_pixels(rows, cols, 4, Q16, seed=5)
    dev, ref = run_pair(im, refmod, px, colorspace="RGB")
    assert_parity(im.equalize_image(dev).numpy(), ref.equalize().numpy(), True, "equalize (large, linear RGB)")


# ----------------------------------- full-size property checks, configs C3 / C4 / C5
def test_resize_full_size_properties(im):
    """BASELINE C3 geometry (8192^2 -> 32768^2 Lanczos, float Quantum RGBA, 17 GB result):
    normalised weights reproduce a constant image exactly, and an image that is constant along
    x resizes like its 1-D column profile (every output column identical)."""
    import torch
    n = 8192
    const = torch.full((n, n, 4), 4321.0, dtype=torch.float32, device="cuda")
    out = im.resize_image(im.Image(const), 4 * n, 4 * n, "Lanczos").pixels
    assert out.shape == (4 * n, 4 * n, 4)
    assert float((out - 4321.0).abs().max()) == 0.0
    del out, const
    torch.cuda.empty_cache()
    g = torch.Generator(device="cuda").manual_seed(11)
    profile = torch.rand((n, 1, 4), generator=g, device="cuda") * 60000.0 + 100.0
    profile[:, :, 3] = 65535.0
    img = profile.expand(n, 64, 4).contiguous()
    out = im.resize_image(im.Image(img), 256, 4 * n, "Lanczos").pixels
    assert float((out - out[:, :1, :]).abs().max()) == 0.0          # columns stay identical
    tall = im.resize_image(im.Image(profile.contiguous()), 1, 4 * n, "Lanczos").pixels
    assert float((out[:, :1, :] - tall).abs().max()) == 0.0         # == the 1-D resize of the profile


def test_morphology_full_size_properties(im):
    """BASELINE C5 geometry (16384^2 RGBA Q16, Disk:15): a single bright pixel dilates into the
    709-cell disk; Erode and Dilate are dual under negation for this symmetric kernel."""
    import torch
    n = 16384
    img = torch.zeros((n, n, 4), dtype=torch.int16, device="cuda")
    img[5000, 7000, :] = -1                                           # 65535
    out = im.morphology_image(im.Image(img.view(torch.uint16)), "Dilate", 1, "Disk:15").pixels
    lit = (out.view(torch.int16)[:, :, 0] != 0)
    assert int(lit.sum()) == 709
    ys, xs = torch.nonzero(lit, as_tuple=True)
    assert int(ys.min()) == 4985 and int(ys.max()) == 5015 and int(xs.min()) == 6985 and int(xs.max()) == 7015
    del img, out, lit
    torch.cuda.empty_cache()
    m = 4096
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randint(-32768, 32768, (m, m, 4), generator=g, device="cuda", dtype=torch.int16)
    neg = (~a)                                                        # 65535 - value on the uint16 view
    dil = im.morphology_image(im.Image(a.view(torch.uint16)), "Dilate", 1, "Disk:15").pixels.view(torch.int16)
    ero = im.morphology_image(im.Image(neg.contiguous().view(torch.uint16)), "Erode", 1, "Disk:15").pixels.view(torch.int16)
    assert bool(((~ero) == dil).all())


def test_lab_contrast_stretch_full_size_properties(im):
    """BASELINE C4 geometry (4096^2 RGBA Q16), checked against the oracle without running the
    oracle at full size: in a gray ramp image (R=G=B=A=v, every v of 0..65535 exactly 256 times)
    the result of sRGB->Lab + ContrastStretch(2% x 1%) at a pixel depends only on v, and the
    histogram is 256 x that of a 256x256 image holding each v once — so the full-size device
    result must equal the oracle's 256x256 result looked up by v."""
    import torch
    from oracle import restate as R
    n = 4096
    small = np.arange(65536, dtype=np.uint16).reshape(256, 256, 1).repeat(4, axis=2)
    want = R.transform_image_colorspace(small, "srgb", "lab")
    want_lab = want.copy()
    cnt = 256 * 256
    want = R.contrast_stretch_image(want, 0.02 * cnt, cnt - 0.01 * cnt, colorspace="lab")
    v = (torch.arange(n * n, device="cuda", dtype=torch.int64) & 0xFFFF).to(torch.int32)
    ramp = v.to(torch.int16).view(n, n, 1).expand(n, n, 4).contiguous().view(torch.uint16)
    img = im.Image(ramp)
    h = im.histogram(img, True)
    assert int(h[:, 0].sum()) == n * n
    im.transform_image_colorspace(img, "Lab")
    lut = torch.from_numpy(want_lab.reshape(65536, 4).astype(np.int32)).cuda()
    got = img.pixels.view(torch.int16).to(torch.int32) & 0xFFFF
    d = (got.view(-1, 4) - lut[v.to(torch.int64)]).abs()
    assert int(d.max()) <= 1 and float((d == 0).float().mean()) > 0.9999, "sRGB->Lab at full size"
    im.contrast_stretch_image(img, 0.02 * n * n, n * n - 0.01 * n * n)
    lut = torch.from_numpy(want.reshape(65536, 4).astype(np.int32)).cuda()
    got = img.pixels.view(torch.int16).to(torch.int32) & 0xFFFF
    d = (got.view(-1, 4) - lut[v.to(torch.int64)]).abs()
    assert int(d.max()) <= 1 and float((d == 0).float().mean()) > 0.999, "Lab + ContrastStretch at full size"


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("colorspace", ["RGB", "sRGB"])
@pytest.mark.parametrize("shape", [(60, 70, 3), (1030, 1040, 4)])
def test_histogram_operators_linear_rgb(im, refmod, dtype, colorspace, shape):
    """GetPixelIntensity encodes linear RGB before weighting (pixel.c:2418-2424); the large
    frame takes the LDS histogram and the shared-column LUT apply."""
    px = make_pixels(*shape, dtype, kind="smooth")
    dev, ref = run_pair(im, refmod, px, colorspace=colorspace)
    assert_parity(im.equalize_image(dev).numpy(), ref.equalize().numpy(), True, "equalize " + colorspace)
    dev, ref = run_pair(im, refmod, px, colorspace=colorspace)
    n = shape[0] * shape[1]
    got = im.contrast_stretch_image(dev, 0.03 * n, n - 0.02 * n).numpy()
    assert_parity(got, ref.contrast_stretch(0.03 * n, n - 0.02 * n).numpy(), True, "stretch " + colorspace)


def test_resize_callback_filter_matches_builtin(im, refmod):
    """MhAcquireResizeFilterFromCallback (what the MagickCore shim uses): weights supplied by the
    caller — here the reference's own GetResizeFilterWeight through the oracle driver."""
    import ctypes
    from imagemagick_amd import _lib
    lib = _lib.load()
    px = make_pixels(40, 52, 4, Q16)
    ref = refmod.RefImage(px)
    cb_type = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p, ctypes.c_double)

    def weight(_user, x):
        return float(ref.filter_weights("Lanczos", [x])[0][0])
    cb = cb_type(weight)
    support = ref.filter_weights("Lanczos", [0.0])[1]
    flt = lib.MhAcquireResizeFilterFromCallback(cb, None, support)
    assert flt
    try:
        src = im.Image(to_device(px))
        out = src.like(rows=97, columns=130)
        _lib.check(lib.MagickHipResizeImageWithFilter(ctypes.byref(src.descriptor()),
                                                      ctypes.byref(out.descriptor()), flt))
    finally:
        lib.MhDestroyResizeFilter(flt)
    assert_parity(out.numpy(), ref.resize(130, 97, "Lanczos").numpy(), True, "callback filter")


@pytest.mark.parametrize("case", ["tiny_alpha", "sparse_alpha", "binary_alpha", "checker", "dark", "smooth"])
@pytest.mark.parametrize("sigma", [2.0, 10.0])
def test_blur_fast_adversarial(im, refmod, case, sigma):
    """FAST (f32) BlurImage against the reference on inputs that stress the alpha-weighted
    normalisation: must stay within +-1 level (the contract of the mode bench.py runs)."""
    rng = np.random.default_rng(17)
    rows, cols = 150, 170
    px = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    if case == "tiny_alpha":
        px[:, :, 3] = rng.integers(0, 4, (rows, cols))
    elif case == "sparse_alpha":
        px[:, :, 3] = np.where(rng.random((rows, cols)) < 0.02, 65535, 0)
    elif case == "binary_alpha":
        px[:, :, 3] = np.where(rng.random((rows, cols)) < 0.5, 65535, 0)
    elif case == "checker":
        y, x = np.mgrid[0:rows, 0:cols]
        px[:, :, :3] = (((x + y) & 1) * 65535)[:, :, None]
    elif case == "dark":
        px[:, :, :3] = rng.integers(0, 8, (rows, cols, 3))
    elif case == "smooth":
        px = make_pixels(rows, cols, 4, Q16, kind="smooth")
    dev, ref = run_pair(im, refmod, px)
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.blur_image(dev, 0.0, sigma).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(got, ref.blur(0.0, sigma).numpy(), False, "fast blur %s sigma=%g" % (case, sigma))


# ------------------------------------------------- GrayscaleImage / FunctionImage
@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("method", ["Rec709Luma", "Rec601Luma", "Rec709Luminance", "Rec601Luminance",
                                    "Average", "Brightness", "Lightness", "MS", "RMS"])
@pytest.mark.parametrize("colorspace,channels", [("sRGB", 4), ("RGB", 3)])
def test_grayscale(im, refmod, dtype, method, colorspace, channels):
    px = make_pixels(37, 50, channels, dtype)
    dev, ref = run_pair(im, refmod, px, colorspace=colorspace)
    got = im.grayscale_image(dev, method).numpy()
    want = ref.grayscale(method).numpy()          # re-laid out as gray[+alpha] by SetImageColorspace
    assert_parity(np.ascontiguousarray(got[:, :, 0]), np.ascontiguousarray(want[:, :, 0]), True,
                  "grayscale %s %s" % (method, colorspace))
    if channels == 4:
        assert np.array_equal(got[:, :, 3], want[:, :, -1])


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("target", ["Gray", "LinearGray"])
@pytest.mark.parametrize("channels", [3, 4])
def test_colorspace_srgb_to_gray(im, refmod, dtype, target, channels):
    """sRGB -> GRAY / LinearGRAY (colorspace.c:843-957) through the library: the gray value in the
    first channel (the layout change to one channel is SetImageColorspace's, the caller's part)."""
    px = make_pixels(41, 57, channels, dtype, seed=channels)
    dev, ref = run_pair(im, refmod, px)
    im.transform_image_colorspace(dev, target)
    want = ref.colorspace(target).numpy()         # gray[+alpha]
    got = dev.numpy()
    assert_parity(np.ascontiguousarray(got[:, :, 0]), np.ascontiguousarray(want[:, :, 0]), True,
                  "sRGB -> %s" % target, max_ulp=1)
    if channels == 4:
        assert np.array_equal(got[:, :, 3], want[:, :, -1])
    with pytest.raises(im.MagickHipError):        # only from sRGB
        im.transform_image_colorspace(im.Image(to_device(px), colorspace="Lab"), target)


@pytest.mark.parametrize("dtype", [Q16, HDRI])
@pytest.mark.parametrize("function,params", [("Polynomial", (0.3, -1.2, 1.5, 0.1)), ("Polynomial", (2.0, 0.0)),
                                             ("Sinusoid", (3.0, 90.0, 0.4, 0.5)), ("Sinusoid", (1.0,)),
                                             ("Arcsin", (0.8, 0.45, 1.0, 0.5)), ("Arctan", (4.0, 0.5, 1.0, 0.5))])
def test_function(im, refmod, dtype, function, params):
    px = make_pixels(33, 47, 4, dtype)
    dev, ref = run_pair(im, refmod, px)
    got = im.function_image(dev, function, params).numpy()
    want = ref.function(function, params).numpy()
    # sin/asin/atan are the device's, not libm's: a last-bit difference of the double flips a
    # Q16 rounding only on an exact tie; float Quantum may differ by one ULP
    exact = function == "Polynomial"
    assert_parity(got, want, exact, "function " + function, max_ulp=0 if exact else 1)


def test_operators_are_reentrant_across_threads_and_streams(im, refmod):
    """SURVEY §8b threading: operators may be called concurrently from user threads; each call
    here runs on its own torch stream (MhImage::stream) with its own image.  EXACT and FAST
    threads run AT THE SAME TIME: the precision belongs to the call
    (MhImage::precision), not to the process — the even threads ask for EXACT and must be
    bit-identical, the odd ones for FAST (the hybrid blur kernel) and must be within +-1, while
    the library-wide default is flipped back and forth underneath them."""
    import threading
    import torch
    jobs = []
    for i in range(6):
        px = make_pixels(120 + 8 * i, 150, 4, Q16, seed=100 + i)
        jobs.append((px, refmod.RefImage(px).blur(0.0, 3.0).resize(90, 70, "Lanczos").numpy(),
                     refmod.RefImage(px).blur(0.0, 3.0).numpy()))
    results = [None] * len(jobs)
    blurred = [None] * len(jobs)
    launched = [None] * len(jobs)
    errors = []

    def work(k):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                precision = im.PRECISION_FAST if k % 2 else im.PRECISION_EXACT
                dev = im.Image(to_device(jobs[k][0]), precision=precision)
                blur = im.blur_image(dev, 0.0, 3.0)
                out = im.resize_image(blur, 90, 70, "Lanczos")
                stream.synchronize()
                results[k] = out.numpy()
                blurred[k] = blur.numpy()
        except Exception as exc:                      # surfaced below
            errors.append(exc)

    for round_ in range(3):
        im.set_precision(im.PRECISION_FAST if round_ % 2 else im.PRECISION_EXACT)     # the default: nobody's business
        try:
            threads = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            im.set_precision(im.PRECISION_EXACT)
        assert not errors, errors
        for k, (_, want, want_blur) in enumerate(jobs):
            if k % 2 == 0:
                assert_parity(blurred[k], want_blur, True, "EXACT thread %d: blur" % k)
                assert_parity(results[k], want, True, "EXACT thread %d" % k)
            else:
                assert_parity(blurred[k], want_blur, False, "FAST thread %d: blur" % k)
    # ... and the FAST calls really took the FAST kernel (not the default's)
    import bench
    dev = im.Image(to_device(jobs[1][0]), precision=im.PRECISION_FAST)
    assert set(bench.kernel_profile(im, lambda: im.blur_image(dev, 0.0, 3.0), 1)) == {"blur_fused_hybrid"}
    dev = im.Image(to_device(jobs[1][0]), precision=im.PRECISION_EXACT)
    im.set_precision(im.PRECISION_FAST)
    try:
        # (+ the two fp64 passes queued behind the exact kernel for the frames it gives up: empty here)
        launched = set(bench.kernel_profile(im, lambda: im.blur_image(dev, 0.0, 3.0), 1))
        assert "blur_fused_exact" in launched and launched <= {"blur_fused_exact", "conv_row", "conv_column"}, launched
    finally:
        im.set_precision(im.PRECISION_EXACT)
