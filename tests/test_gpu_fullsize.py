"""BASELINE.json's configurations at their REAL sizes against the compiled reference
(oracle/_ref): C2 8192^2 blur, C3 8192^2 -> 32768^2 Lanczos (float Quantum), C4 4096^2
sRGB->Lab + ContrastStretch, C5 16384^2 Dilate Disk:15 and UnsharpMask.

The device runs the whole frame, so every strip / segment / tile boundary of the full-size
launch grids is on the checked path.  The reference runs the whole frame where that takes
seconds on the GPU box's cores (C2: 13 s, C4), and bands of it where it does not:
  * translation-invariant stencils (Dilate, UnsharpMask): crops with the operator's reach as
    halo, compared on their interior;
  * resize: crops that START at row 0 / column 0 (the contribution windows are computed from
    absolute coordinates with an added 1e-12, so only a crop with the same origin reproduces
    them bit for bit), compared up to the filter support before the crop's far edge.
Bar: EXACT bit-identical; FAST within +-1 Quantum level (1 float ULP for float Quantum) with no
exempted region."""
import os

import numpy as np
import pytest

from conftest import to_device, ulp_diff_f32

pytestmark = pytest.mark.gpu


def _levels(t):
    import torch
    return t.view(torch.int16).to(torch.int32) & 0xffff


def _compare_q16(got_dev, want_np, exact, what):
    """got_dev: device uint16 tensor; want_np: host uint16 array of the same shape."""
    import torch
    want = torch.from_numpy(want_np.view(np.int16)).cuda()
    d = (_levels(got_dev) - _levels(want.view(torch.uint16))).abs()
    worst = int(d.max())
    same = float((d == 0).double().mean())
    limit = 0 if exact else 1
    assert worst <= limit, "%s: max |device - reference| = %d (limit %d), %d samples over" % (
        what, worst, limit, int((d > limit).sum()))
    return same


@pytest.fixture(scope="module")
def c2_case(refmod):
    """8192^2 uniform-random RGBA with a band of tiny alpha (0..3 levels: the row pass's alpha
    lands on rounding ties there), a band of small alpha (0..600 levels), a fully transparent
    band and an opaque band; and the reference's BlurImage(0,10) of it."""
    n = 8192
    rng = np.random.default_rng(2024)
    px = rng.integers(0, 65536, (n, n, 4), dtype=np.uint16)
    px[: n // 16, :, 3] = rng.integers(0, 4, (n // 16, n))
    px[n // 16: n // 8, :, 3] = rng.integers(0, 600, (n // 16, n))
    px[n // 8: n // 8 + 200, :, 3] = 0
    px[n // 4: n // 4 + 300, :, 3] = 65535
    refmod.set_thread_limit(os.cpu_count() or 1)
    want = refmod.RefImage(px).blur(0.0, 10.0).numpy()
    return px, want


def test_c2_blur_exact_full_size(im, c2_case):
    px, want = c2_case
    got = im.blur_image(im.Image(to_device(px)), 0.0, 10.0).pixels
    _compare_q16(got, want, True, "C2 BlurImage EXACT")


def test_gray_blur_and_unsharp_full_size_take_the_four_band_form(im, refmod):
    """A 4099 x 4096 one-channel Q16 frame (16.8 Mpixel: above the route's default threshold, a height that is
    not a multiple of four) with nothing set: BlurImage(0,10) and UnsharpMaskImage(0,10,1.5,0.01) run as four row
    bands through the one-launch kernels (operators.cpp fused_blur_gray_bands) — FAST BlurImage within one
    level, everything else bit-identical, on the whole frame; and a frame below the threshold keeps its own two
    passes."""
    import bench
    rows, cols = 4099, 4096
    rng = np.random.default_rng(409)
    px = rng.integers(0, 65536, (rows, cols, 1), dtype=np.uint16)
    px[:3] = 65535                                     # the frame's edges unlike what lies inside
    px[-2:] = 0
    px[1024:1030] = 0                                  # ... and the rows either side of a band boundary
    px[1030:1036] = 65535
    refmod.set_thread_limit(os.cpu_count() or 1)
    ref = refmod.RefImage(px)
    want_blur = ref.blur(0.0, 10.0).numpy().reshape(rows, cols, 1)
    want_unsharp = ref.unsharp(0.0, 10.0, 1.5, 0.01).numpy().reshape(rows, cols, 1)
    image = im.Image(to_device(px))
    holder = {}
    for precision, exact in ((im.PRECISION_FAST, False), (im.PRECISION_EXACT, True)):
        im.set_precision(precision)
        try:
            launched = set(bench.kernel_profile(im, lambda: holder.update(b=im.blur_image(image, 0.0, 10.0)), 1))
            assert launched == {"gray_bands_pack", "blur_fused_exact" if exact else "blur_fused_hybrid",
                                "gray_bands_unpack"}, launched
            launched = set(bench.kernel_profile(
                im, lambda: holder.update(u=im.unsharp_mask_image(image, 0.0, 10.0, 1.5, 0.01)), 1))
            assert launched == {"gray_bands_pack", "unsharp_fused_exact", "gray_bands_unpack"}, launched
        finally:
            im.set_precision(im.PRECISION_EXACT)
        _compare_q16(holder["b"].pixels, want_blur, exact, "gray BlurImage, precision %d" % precision)
        _compare_q16(holder["u"].pixels, want_unsharp, True, "gray UnsharpMaskImage, precision %d" % precision)
    small = im.Image(to_device(px[:1024, :1024].copy()))
    launched = set(bench.kernel_profile(im, lambda: holder.update(s=im.blur_image(small, 0.0, 10.0)), 1))
    assert "gray_bands_pack" not in launched, launched
    # ... and Erode / Dilate Disk:15 through the union-of-rectangles kernel (morphology.hip try_rects_gray_bands)
    for method in ("Dilate", "Erode"):
        launched = set(bench.kernel_profile(im, lambda: holder.update(m=im.morphology_image(image, method, 1, "Disk:15")), 1))
        assert launched == {"gray_bands_pack", "morph_rects", "gray_bands_unpack"}, launched
        want = ref.morphology(method, 1, "Disk:15").numpy().reshape(rows, cols, 1)
        _compare_q16(holder["m"].pixels, want, True, "gray %s Disk:15" % method)


def test_c2_blur_exact_tiny_alpha_frame_is_given_up_to_the_fp64_passes(im, refmod):
    """EXACT BlurImage on an 8192^2 frame whose alpha is 0..3 levels EVERYWHERE: the certificate of the
    exact-integer kernel cannot decide one sample in twelve there and their reference-order recomputation
    cost 22 ms (round 4).  The kernel now gives such a frame up after a few groups and the two fp64 passes
    queued behind it compute it: bit-identical (reference bands: top rows, left columns), <= 3 ms."""
    import bench
    import torch
    n, band = 8192, 400
    rng = np.random.default_rng(77)
    px = rng.integers(0, 65536, (n, n, 4), dtype=np.uint16)
    px[:, :, 3] = rng.integers(0, 4, (n, n))
    refmod.set_thread_limit(os.cpu_count() or 1)
    want_top = refmod.RefImage(px[:band]).blur(0.0, 10.0).numpy()
    want_left = refmod.RefImage(np.ascontiguousarray(px[:, :band])).blur(0.0, 10.0).numpy()
    keep = band - 48                                  # 39 rows / columns of the window + margin
    dev = im.Image(to_device(px))
    holder = {}
    holder.update(out=im.blur_image(dev, 0.0, 10.0))          # (the kernels' code objects load on first use)
    prof = bench.kernel_profile(im, lambda: holder.update(out=im.blur_image(dev, 0.0, 10.0)), 3)
    total_ms = sum(v["avg_ms"] for v in prof.values())
    got = holder["out"].pixels
    _compare_q16(got[:keep], want_top[:keep], True, "tiny-alpha frame, top band")
    _compare_q16(got[:, :keep].contiguous(), np.ascontiguousarray(want_left[:, :keep]), True, "tiny-alpha frame, left band")
    assert "blur_fused_exact" in prof and any(k.startswith("conv_") for k in prof), prof
    assert total_ms <= 3.0, prof
    # an ordinary frame: the passes behind the kernel leave at once
    ordinary = im.Image(to_device(rng.integers(0, 65536, (n, n, 4), dtype=np.uint16)))
    prof = bench.kernel_profile(im, lambda: holder.update(out=im.blur_image(ordinary, 0.0, 10.0)), 3)
    behind = sum(v["avg_ms"] for k, v in prof.items() if k.startswith("conv_"))
    assert behind <= 0.12, prof
    del holder, got
    torch.cuda.empty_cache()


@pytest.mark.parametrize("path", ["fused", "two_pass", "vector"])
def test_c2_blur_fast_full_size(im, c2_case, path):
    """The mode bench.py times, on every path FAST can take: both passes in one launch
    (convolve_fused_hybrid.hip), one launch per pass on the matrix cores, and the f32 vector kernels."""
    px, want = c2_case
    env = {"fused": {}, "two_pass": {"MAGICKHIP_NO_FUSED_BLUR": "1"}, "vector": {"MAGICKHIP_NO_MFMA": "1"}}[path]
    old = {k: im.get_option(k) for k in env}
    for k, v in env.items():
        im.set_option(k, v)                 # (the library reads the environment only at start-up)
    im.set_precision(im.PRECISION_FAST)
    try:
        got = im.blur_image(im.Image(to_device(px)), 0.0, 10.0).pixels
    finally:
        im.set_precision(im.PRECISION_EXACT)
        for k, v in old.items():
            im.set_option(k, v)
    same = _compare_q16(got, want, False, "C2 BlurImage FAST (%s)" % path)
    # the contract is +-1 on every sample (_compare_q16); the identical share is what to expect of a
    # healthy kernel: the one-launch form does not round its intermediate colour (round 4) and ends
    # with 96.9 % identical samples, the two-pass forms round it like the reference: > 97 %
    assert same > (0.95 if path == "fused" else 0.97)


def _ordered_bits(t):
    """float32 device tensor -> int64 whose order is the floats' order (distance = ULPs)."""
    import torch
    bits = t.view(torch.int32).to(torch.int64)
    return torch.where(bits < 0, -(bits & 0x7fffffff), bits)


def _compare_whole_frame(out, want_ref, limit, what, band=1024):
    """Device result against the reference's pixel cache, `band` rows at a time (a 17 GB frame does not
    get a second host copy): max ULP (float) / level (Q16) difference over EVERY sample."""
    import torch
    rows = out.shape[0]
    is_float = out.dtype == torch.float32
    buf = np.empty((band,) + tuple(out.shape[1:]), dtype=np.float32 if is_float else np.uint16)
    worst = over = 0
    for y0 in range(0, rows, band):
        n = min(band, rows - y0)
        want_ref.numpy_rows(y0, n, buf[:n])
        if is_float:
            d = (_ordered_bits(out[y0: y0 + n]) - _ordered_bits(torch.from_numpy(buf[:n]).cuda())).abs()
        else:
            d = (_levels(out[y0: y0 + n]) - _levels(torch.from_numpy(buf[:n].view(np.int16)).cuda().view(torch.uint16))).abs()
        worst = max(worst, int(d.max()))
        over += int((d > limit).sum())
        del d
    assert worst <= limit, "%s: max difference %d (limit %d), %d samples over" % (what, worst, limit, over)


def test_c3_resize_full_size(im, refmod):
    """8192^2 -> 32768^2 Lanczos, float Quantum RGBA (17 GB result): the WHOLE frame against the reference's
    (resize.c:3549-3759 then :3333-3547, about 11 s on the GPU box's cores; its result stays in the
    reference's pixel cache and is compared a band of rows at a time on the device) — every strip x chunk item
    of the one-launch grid, EXACT bit-identical, FAST within one float ULP."""
    import torch
    n = 8192
    g = torch.Generator(device="cuda").manual_seed(33)
    src = torch.rand((n, n, 4), generator=g, device="cuda", dtype=torch.float32) * 65535.0
    src[:, : n // 2, 3] = 65535.0                     # half opaque, half varying alpha
    refmod.set_thread_limit(os.cpu_count() or 1, True)
    want = refmod.RefImage(src.cpu().numpy()).resize(4 * n, 4 * n, "Lanczos")
    for precision, limit in ((im.PRECISION_EXACT, 0), (im.PRECISION_FAST, 1)):
        im.set_precision(precision)
        try:
            out = im.resize_image(im.Image(src), 4 * n, 4 * n, "Lanczos").pixels
        finally:
            im.set_precision(im.PRECISION_EXACT)
        _compare_whole_frame(out, want, limit, "C3 resize, precision %d" % precision)
        del out
        torch.cuda.empty_cache()


@pytest.mark.parametrize("is_float,factor", [(False, 3), (True, 4)])
def test_resize_fast_full_size_sprite_frame(im, refmod, is_float, factor):
    """FAST ResizeImage of an 8192^2 SPRITE frame (opaque rectangles with binary-alpha fringes on a transparent
    ground; x3 Q16 and x4 float): at every rectangle's edge the intermediate alpha crosses rounding
    boundaries and alpha sums cancel under the Lanczos window, so the one-launch kernel marks thousands of
    items on the real grid and the careful launch redoes them (resize.c:3494-3530, :3709-3745).  The whole
    result against the reference's, within one level / one float ULP (the residue of a cancellation:
    absolute 65535e-9)."""
    import bench
    import torch
    n = 8192
    rng = np.random.default_rng(7 + factor)
    px = rng.integers(0, 65536, (n, n, 4), dtype=np.uint16)
    alpha = np.zeros((n, n), dtype=np.uint16)
    for _ in range(1500):
        y, x = int(rng.integers(0, n - 8)), int(rng.integers(0, n - 8))
        h, w = int(rng.integers(3, 200)), int(rng.integers(3, 200))
        alpha[y: y + h, x: x + w] = 65535
    fringe = rng.random((n, n)) < 0.5                 # binary alpha inside every fourth rectangle row band
    alpha[::4] = np.where(fringe[::4], alpha[::4], 0)
    px[:, :, 3] = alpha
    if is_float:
        px = px.astype(np.float32)
    refmod.set_thread_limit(os.cpu_count() or 1, is_float)
    want = refmod.RefImage(px).resize(factor * n, factor * n, "Lanczos")
    image = im.Image(to_device(px))
    holder = {}
    im.set_precision(im.PRECISION_FAST)
    try:
        launched = bench.kernel_profile(
            im, lambda: holder.update(out=im.resize_image(image, factor * n, factor * n, "Lanczos").pixels), 1)
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert set(launched) == {"resize_stream", "resize_stream_careful"}, launched
    out = holder.pop("out")
    if is_float:
        # values that are the residue of a cancellation agree absolutely (conftest.assert_parity's `residue`)
        want_rows = np.empty((1024, factor * n, 4), dtype=np.float32)
        worst = 0
        for y0 in range(0, factor * n, 1024):
            want.numpy_rows(y0, 1024, want_rows)
            w = torch.from_numpy(want_rows).cuda()
            got = out[y0: y0 + 1024]
            d = (_ordered_bits(got) - _ordered_bits(w)).abs()
            d = torch.where((got.double() - w.double()).abs() <= 65535.0e-9, torch.zeros_like(d), d)
            worst = max(worst, int(d.max()))
            del d, w
        assert worst <= 1, "sprite frame x%d float: max %d ULP" % (factor, worst)
    else:
        _compare_whole_frame(out, want, 1, "sprite frame x%d Q16" % factor)


def test_c4_lab_contrast_stretch_full_size(im, refmod):
    """One 4096^2 image of the C4 batch against the reference run on the whole frame."""
    n = 4096
    rng = np.random.default_rng(44)
    px = rng.integers(0, 65536, (n, n, 4), dtype=np.uint16)
    refmod.set_thread_limit(os.cpu_count() or 1)
    ref = refmod.RefImage(px).colorspace("Lab")
    want_lab = ref.numpy()
    want = ref.contrast_stretch(0.02 * n * n, n * n - 0.01 * n * n).numpy()
    img = im.Image(to_device(px))
    im.transform_image_colorspace(img, "Lab")
    _compare_q16(img.pixels, want_lab, True, "C4 sRGB->Lab")
    im.contrast_stretch_image(img, 0.02 * n * n, n * n - 0.01 * n * n)
    _compare_q16(img.pixels, want, True, "C4 Lab + ContrastStretch")


def test_c4_fast_lab_within_one_level_and_stretch_exact(im, refmod):
    """FAST sRGB->Lab computes in f32 (colorspace_lab_fast_kernel): every sample within one level
    of the reference's fp64 result on the whole 4096^2 frame.  ContrastStretch is integer counts
    and an fp64 map in both precisions: given that same Lab frame it is bit-identical to the
    reference's (the one-pass packed histogram included)."""
    n = 4096
    rng = np.random.default_rng(45)
    px = rng.integers(0, 65536, (n, n, 4), dtype=np.uint16)
    px[0, :4096:16, :3] = 0                       # black, and the dark linear segments of both curves
    px[1, :, 0] = np.arange(n, dtype=np.uint16)
    px[1, :, 1] = np.arange(n, dtype=np.uint16) // 3
    px[1, :, 2] = 5
    px[2, :, :3] = (np.arange(n, dtype=np.uint32) * 16 + 7).astype(np.uint16)[:, None]      # grays
    px[3, :64, :3] = 65535
    refmod.set_thread_limit(os.cpu_count() or 1)
    want_lab = refmod.RefImage(px).colorspace("Lab").numpy()
    img = im.Image(to_device(px))
    im.set_precision(im.PRECISION_FAST)
    try:
        im.transform_image_colorspace(img, "Lab")
        _compare_q16(img.pixels, want_lab, False, "C4 FAST sRGB->Lab")
        got_lab = img.numpy().copy()
        assert float((got_lab == want_lab).mean()) > 0.9
        im.contrast_stretch_image(img, 0.02 * n * n, n * n - 0.01 * n * n)
    finally:
        im.set_precision(im.PRECISION_EXACT)
    want = refmod.RefImage(got_lab, "Lab").contrast_stretch(0.02 * n * n, n * n - 0.01 * n * n).numpy()
    _compare_q16(img.pixels, want, True, "C4 ContrastStretch of the FAST Lab frame")
    # the one-call form converts and bins in one kernel: the same frame, bit for bit
    import bench
    fused = im.Image(to_device(px))
    im.set_precision(im.PRECISION_FAST)
    try:
        launched = set(bench.kernel_profile(im, lambda: im.transform_colorspace_contrast_stretch_image(
            fused, "Lab", 0.02 * n * n, n * n - 0.01 * n * n), 1))
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert "colorspace_histogram" in launched and "histogram" not in launched, launched
    assert fused.colorspace == "lab"
    _compare_q16(fused.pixels, want, True, "C4 fused sRGB->Lab + ContrastStretch")
    # ... and EXACT (or any frame the fused kernel does not take) is the two operators in sequence
    plain = im.Image(to_device(px[:700, :900].copy()))
    im.transform_colorspace_contrast_stretch_image(plain, "Lab", 100.0, 200.0)
    want_plain = refmod.RefImage(px[:700, :900].copy()).colorspace("Lab").contrast_stretch(100.0, 200.0).numpy()
    _compare_q16(plain.pixels, want_plain, True, "one-call form, EXACT")


def _band_starts(n, band):
    return [0, n // 2 - band // 2, n - band]


def test_c5_dilate_disk15_full_size(im, refmod):
    """16384^2 RGBA Q16 Dilate Disk:15 (the convex-kernel fast path): three full-width bands and
    one full-height band of the reference, 15-pixel halo."""
    import torch
    n, band, reach = 16384, 192, 15
    rng = np.random.default_rng(55)
    px = rng.integers(0, 65536, (n, n, 4), dtype=np.uint16)
    refmod.set_thread_limit(os.cpu_count() or 1)
    out = im.morphology_image(im.Image(to_device(px)), "Dilate", 1, "Disk:15").pixels
    for y0 in _band_starts(n, band):
        lo, hi = max(y0 - reach, 0), min(y0 + band + reach, n)
        want = refmod.RefImage(px[lo:hi]).morphology("Dilate", 1, "Disk:15").numpy()
        _compare_q16(out[y0:y0 + band], want[y0 - lo:y0 - lo + band], True, "C5 Dilate rows %d.." % y0)
    x0 = n // 3
    lo, hi = x0 - reach, x0 + 64 + reach
    want = refmod.RefImage(np.ascontiguousarray(px[:, lo:hi])).morphology("Dilate", 1, "Disk:15").numpy()
    _compare_q16(out[:, x0:x0 + 64].contiguous(), np.ascontiguousarray(want[:, reach:reach + 64]), True,
                 "C5 Dilate columns %d.." % x0)


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_c5_convolve_disk15_full_size(im, refmod, precision):
    """16384^2 RGBA Q16 ConvolveMorphology Disk:15 with convolve:scale='!' (SURVEY 8d's MAC-bound
    variant of C5; 709 cells of 1/709): exact integer sums on the i8 matrix cores in BOTH precision
    modes — bit-identical to three full-width bands and one full-height band of the reference,
    15-pixel halo."""
    n, band, reach = 16384, 192, 15
    rng = np.random.default_rng(56)
    px = rng.integers(0, 65536, (n, n, 4), dtype=np.uint16)
    refmod.set_thread_limit(os.cpu_count() or 1)
    im.set_precision(im.PRECISION_FAST if precision == "fast" else im.PRECISION_EXACT)
    try:
        out = im.morphology_image(im.Image(to_device(px)), "Convolve", 1, "Disk:15", scale=(1.0, 1)).pixels
    finally:
        im.set_precision(im.PRECISION_EXACT)

    def reference(pixels):
        return refmod.RefImage(pixels).set_artifact("convolve:scale", "!").morphology("Convolve", 1, "Disk:15").numpy()
    for y0 in _band_starts(n, band):
        lo, hi = max(y0 - reach, 0), min(y0 + band + reach, n)
        want = reference(px[lo:hi])
        _compare_q16(out[y0:y0 + band], want[y0 - lo:y0 - lo + band], True, "C5 Convolve rows %d.." % y0)
    x0 = n // 3
    lo, hi = x0 - reach, x0 + 64 + reach
    want = reference(np.ascontiguousarray(px[:, lo:hi]))
    _compare_q16(out[:, x0:x0 + 64].contiguous(), np.ascontiguousarray(want[:, reach:reach + 64]), True,
                 "C5 Convolve columns %d.." % x0)


@pytest.mark.parametrize("precision", ["exact", "fast"])
def test_c5_unsharp_full_size(im, refmod, precision):
    """16384^2 RGBA Q16 UnsharpMask(0x10+1.0+0.02): bands of the reference with the blur's
    39-pixel reach as halo.  Bit-identical in both modes: FAST UnsharpMaskImage runs its one launch
    on the exact blur (a blurred sample one level off would move the result by `gain` levels and
    flip the threshold test, effect.c:4364-4369)."""
    import torch
    n, band, reach = 16384, 128, 39
    rng = np.random.default_rng(56)
    px = rng.integers(0, 65536, (n, n, 4), dtype=np.uint16)
    px[:, : n // 4, 3] = 65535
    refmod.set_thread_limit(os.cpu_count() or 1)
    im.set_precision(im.PRECISION_FAST if precision == "fast" else im.PRECISION_EXACT)
    try:
        out = im.unsharp_mask_image(im.Image(to_device(px)), 0.0, 10.0, 1.0, 0.02).pixels
    finally:
        im.set_precision(im.PRECISION_EXACT)
    limit = 0
    for y0 in _band_starts(n, band):
        lo, hi = max(y0 - reach, 0), min(y0 + band + reach, n)
        crop = refmod.RefImage(px[lo:hi])
        want = crop.unsharp(0.0, 10.0, 1.0, 0.02).numpy()[y0 - lo:y0 - lo + band].astype(np.int64)
        got = out[y0:y0 + band].cpu().numpy().astype(np.int64)
        d = np.abs(got - want)
        assert int(d.max()) <= limit, "C5 UnsharpMask %s rows %d..: max diff %d" % (precision, y0, int(d.max()))
