#!/usr/bin/env python3
"""Randomised differential run against the compiled reference.  tests/test_gpu_stress_slice.py runs a fixed-seed
slice of every operator family below inside the suite; this script runs them for as long as asked.  Shapes, kernel lengths and operators are drawn at random, every result
is compared with the reference: FAST blur / unsharp within +-1 (unsharp: 1+gain off the threshold
edge), EXACT and the morphology / histogram operators bit-identical.
   python tests/stress_parity.py [seconds] [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repository root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import imagemagick_amd as im
from oracle import ref as refmod

rng = np.random.default_rng(1)
_ready = False


def setup(seed=1):
    """Load the library (once), pin the starting precision and reseed the case generator."""
    global rng, _ready
    rng = np.random.default_rng(seed)
    if not _ready:
        im.load()
        refmod.set_thread_limit(os.cpu_count() or 1)
        im.set_option("MAGICKHIP_RESIZE_ONE_LAUNCH_MIN_PIXELS", "0")     # small frames through the one-launch resize kernels too
        im.set_option("MAGICKHIP_GRAY_BANDS_MIN_PIXELS", "0")            # ... and small gray frames through the four-band form of the fused blur
        im.set_option("MAGICKHIP_RGB_PAD_MIN_PIXELS", "0")               # ... small RGB frames through the padded form of Erode / Dilate
        im.set_option("MAGICKHIP_RGB_PAD_FLOAT_ALWAYS", "1")             #     (float: also under kernels morph_convex would take)
        _ready = True
    im.set_precision(im.PRECISION_EXACT)           # (the library's default is FAST; the cases below switch per call)


def dev(px, **kw):
    t = torch.from_numpy(px.view(np.int16)).cuda().view(torch.uint16)
    return im.Image(t, **kw)


def pixels(rows, cols, kind):
    px = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    if kind == 1:
        px[:, :, 3] = 65535
    elif kind == 2:
        px[:, :, 3] = rng.integers(0, 4, (rows, cols), dtype=np.uint16)          # tiny alpha
    elif kind == 3:
        px[:, :, 3] = np.where(rng.random((rows, cols)) < 0.5, 0, 65535)          # binary alpha
    elif kind == 4:
        px[:] = (np.add.outer(np.arange(rows), np.arange(cols)) % 2 * 40000 + 100)[:, :, None]
    elif kind == 5:                                # a sprite: opaque rectangles on a transparent ground
        px[:, :, 3] = 0
        for _ in range(6):
            y, x = int(rng.integers(0, max(1, rows - 2))), int(rng.integers(0, max(1, cols - 2)))
            px[y: y + int(rng.integers(1, 25)), x: x + int(rng.integers(1, 25)), 3] = 65535
    return px


def dev_float(px, **kw):
    return im.Image(torch.from_numpy(px).cuda(), **kw)


def float_pixels(rows, cols, kind):
    """Float Quantum frames: fractional levels, values beyond the Quantum range and below zero,
    small / zero alpha, neighbouring floats (values on float-rounding midpoints after a blur)."""
    px = (rng.random((rows, cols, 4)) * 65535.0).astype(np.float32)
    if kind == 1:
        px[:, :, 3] = 65535.0
    elif kind == 2:
        px[:, :, 3] = (10.0 ** rng.uniform(-9, 0, (rows, cols))).astype(np.float32)
    elif kind == 3:
        px[:, :, :3] = (rng.random((rows, cols, 3)) * 90000.0 - 12000.0).astype(np.float32)
        px[:, :, 3] = np.where(rng.random((rows, cols)) < 0.3, 0.0, 65535.0)
    elif kind == 5:                                # a sprite: opaque rectangles on a transparent ground
        px[:, :, 3] = 0.0
        for _ in range(6):
            y, x = int(rng.integers(0, max(1, rows - 2))), int(rng.integers(0, max(1, cols - 2)))
            px[y: y + int(rng.integers(1, 25)), x: x + int(rng.integers(1, 25)), 3] = 65535.0
    elif kind == 4:
        low = np.float32(rng.uniform(1.0, 60000.0))
        board = (np.add.outer(np.arange(rows), np.arange(cols)) % 2) == 1
        px[:, :, :3] = np.where(board, np.nextafter(low, np.float32(np.inf)), low)[:, :, None]
        px[:, :, 3] = 65535.0
    return px


def check_bits(name, got, want, detail):
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        bad = np.argwhere(~same)
        print("MISMATCH %s %s: %d float samples differ, first at %s: %r vs %r" % (
            name, detail, len(bad), bad[0].tolist(), got[tuple(bad[0])], want[tuple(bad[0])]), flush=True)
        return 1
    return 0


def check_ulp(name, got, want, limit, detail, residue=0.0):
    """Float results within `limit` float ULPs of the reference's (same NaN pattern, same infinities), or —
    the residue of a cancellation, 1e-20 out of terms of 1e-8 — within `residue` absolutely."""
    if not np.array_equal(np.isnan(got), np.isnan(want)):
        print("MISMATCH %s %s: NaN pattern differs" % (name, detail), flush=True)
        return 1
    if residue > 0.0:
        with np.errstate(invalid="ignore"):
            close = np.abs(got.astype(np.float64) - want.astype(np.float64)) <= residue
        got = np.where(close, want, got)
    def ordered(a):
        bits = np.nan_to_num(a, nan=0.0).view(np.int32).astype(np.int64)
        return np.where(bits < 0, -(bits & 0x7fffffff), bits)
    d = np.abs(ordered(got) - ordered(want))
    if d.max(initial=0) > limit:
        bad = np.argwhere(d > limit)
        print("MISMATCH %s %s: max %d ULP (limit %d), %d samples, first at %s: %r vs %r" % (
            name, detail, d.max(), limit, len(bad), bad[0].tolist(), got[tuple(bad[0])], want[tuple(bad[0])]), flush=True)
        return 1
    return 0


def check(name, got, want, limit, detail):
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    if d.max(initial=0) > limit:
        bad = np.argwhere(d > limit)
        print("MISMATCH %s %s: max %d (limit %d), %d samples, first at %s" % (
            name, detail, d.max(), limit, len(bad), bad[0].tolist()), flush=True)
        return 1
    return 0


# STRESS_OPS=0,1,2 restricts the operators (0 FAST blur, 1 EXACT blur, 2 FAST unsharp, 3 Erode/Dilate on RGBA, gray
# (four row bands) and RGB (a fourth, empty channel) frames, 4 histogram operators, 5 FAST 2-D convolve on the same three
# layouts, 6 FAST Lab, 7 EXACT GaussianBlur / Sharpen (separable +
# tie check), 8 float-Quantum blur / unsharp, 9 float-Quantum Erode / Dilate (RGBA and RGB), 10 float-Quantum
# GaussianBlur / Sharpen, 11 float-Quantum ContrastStretch / Equalize, 12 integer-cell 2-D convolve on the
# i8 matrix cores in both modes and three layouts, 13 2-D convolve with random real cells on every layout and
# Quantum type: the fused fp64 kernel, 14 ResizeImage: whole-number enlargements (one launch on the vector
# pipe under FAST), other enlargements (matrix pipe), reductions and mixed geometries, Q16 and float,
# alpha-weighted or four plain channels, both modes, 15 FAST BlurImage / GaussianBlurImage / UnsharpMaskImage on
# every layout: gray, gray + alpha, RGB, RGBA, four plain channels)
NUMBER_OF_OPS = 16


def blur_radius(sigma):
    """BlurImage's radius argument: 0 (the width from the sigma, gem.c:281-299) most of the time; otherwise a
    radius of its own — short ones cut the Gaussian off, long ones give kernels whose outer taps the reference
    zeroes (|t| < 1e-12, morphology.c:2494-2495) or leaves tiny."""
    if rng.random() < 0.6:
        return 0.0
    return float(np.ceil(rng.uniform(1.0, min(40.0, 6.0 * sigma + 4.0))))


def run_case(op=None):
    """One random case of operator family `op` (None: any); returns the number of mismatches (0 or 1)."""
    failures = 0
    rows, cols = int(rng.integers(1, 260)), int(rng.integers(1, 330))
    if rng.random() < 0.2:
        rows, cols = int(rng.integers(1, 40)), int(rng.integers(300, 1400))
    kind = int(rng.integers(0, 6))
    px = pixels(rows, cols, kind)
    if op is None:
        op = int(rng.integers(0, 16))
    detail = "%dx%d kind %d" % (rows, cols, kind)
    ref = refmod.RefImage(px)
    if op == 0:                                    # FAST blur, every kernel length of the fused launch
        sigma = float(rng.uniform(0.3, 13.4))
        radius = blur_radius(sigma)
        im.set_precision(im.PRECISION_FAST)
        got = im.blur_image(dev(px), radius, sigma).numpy()
        im.set_precision(im.PRECISION_EXACT)
        failures += check("fast blur", got, ref.blur(radius, sigma).numpy(), 1, detail + " %gx%.3f" % (radius, sigma))
    elif op == 1:                                  # EXACT blur (Tie64)
        sigma = float(rng.uniform(0.3, 13.4))
        radius = blur_radius(sigma)
        got = im.blur_image(dev(px), radius, sigma).numpy()
        failures += check("exact blur", got, ref.blur(radius, sigma).numpy(), 0, detail + " %gx%.3f" % (radius, sigma))
    elif op == 2:                                  # FAST unsharp in the fused launch
        sigma = float(rng.uniform(0.5, 12.0))
        gain, threshold = float(rng.uniform(0.3, 3.0)), float(rng.uniform(0.0, 0.2))
        radius = blur_radius(sigma)
        im.set_precision(im.PRECISION_FAST)
        got = im.unsharp_mask_image(dev(px), radius, sigma, gain, threshold).numpy()
        im.set_precision(im.PRECISION_EXACT)
        want = ref.unsharp(radius, sigma, gain, threshold).numpy()
        blurred = ref.blur(radius, sigma).numpy().astype(np.int64)
        edge = np.abs(2 * np.abs(px.astype(np.int64) - blurred) - 65535.0 * threshold) <= 2.0
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        d[edge] = 0
        failures += check("fast unsharp", d, np.zeros_like(d), int(np.ceil(1.0 + gain)),
                          detail + " %gx%.3f gain %.2f thr %.3f" % (radius, sigma, gain, threshold))
    elif op == 3:                                  # symmetric convex kernels (rects)
        family = ["Disk:%.1f" % rng.uniform(0.5, 16.0), "Square:%d" % rng.integers(1, 9),
                  "Diamond:%d" % rng.integers(1, 12), "Octagon:%d" % rng.integers(1, 10),
                  "Plus:%d" % rng.integers(1, 12), "Rectangle:%dx%d" % (2 * rng.integers(0, 9) + 1, 2 * rng.integers(0, 9) + 1)]
        kernel = family[int(rng.integers(0, len(family)))]
        method = "Dilate" if rng.random() < 0.5 else "Erode"
        layout = int(rng.integers(0, 3))           # RGBA; one channel (four row bands); RGB (a fourth, empty channel)
        if layout == 0:
            got = im.morphology_image(dev(px), method, 1, kernel).numpy()
            failures += check(method, got, ref.morphology(method, 1, kernel).numpy(), 0, detail + " " + kernel)
        else:
            frame = np.ascontiguousarray(px[:, :, :1] if layout == 1 else px[:, :, :3])
            iterations = 1 if rng.random() < 0.8 else int(rng.integers(2, 4))
            got = im.morphology_image(dev(frame), method, iterations, kernel).numpy().reshape(frame.shape)
            want = refmod.RefImage(frame).morphology(method, iterations, kernel).numpy().reshape(frame.shape)
            failures += check(method, got, want, 0, detail + " %s x%d c%d" % (kernel, iterations, frame.shape[2]))
    elif op == 4:                                  # histogram operators above a megapixel
        rows2, cols2 = int(rng.integers(1000, 1500)), int(rng.integers(1050, 1900))
        px2 = pixels(rows2, cols2, int(rng.integers(0, 4)))
        px2[0, 0] = (1, 2, 3, 4)
        n = rows2 * cols2
        black, white = float(rng.uniform(0, 0.1)) * n, n - float(rng.uniform(0, 0.1)) * n
        if rng.random() < 0.5:
            got = im.contrast_stretch_image(dev(px2), black, white).numpy()
            want = refmod.RefImage(px2).contrast_stretch(black, white).numpy()
        else:
            got = im.equalize_image(dev(px2)).numpy()
            want = refmod.RefImage(px2).equalize().numpy()
        failures += check("histogram op", got, want, 0, "%dx%d" % (rows2, cols2))
    elif op == 5:                                  # FAST 2-D convolve on the matrix cores
        family = ["Disk:%.1f" % rng.uniform(2.0, 15.9), "Octagon:%d" % rng.integers(2, 12),
                  "Diamond:%d" % rng.integers(2, 14), "Plus:%d" % rng.integers(2, 14),
                  "Ring:%d,%d" % (rng.integers(2, 6), rng.integers(7, 15))]
        kernel = family[int(rng.integers(0, len(family)))]
        layout = int(rng.integers(0, 4))           # RGBA three times in four; one channel (four row bands); RGB
        frame = px if layout < 2 else np.ascontiguousarray(px[:, :, :1] if layout == 2 else px[:, :, :3])
        im.set_precision(im.PRECISION_FAST)
        got = im.morphology_image(dev(frame), "Convolve", 1, kernel, scale=(1.0, 1)).numpy().reshape(frame.shape)
        im.set_precision(im.PRECISION_EXACT)
        want = (ref if layout < 2 else refmod.RefImage(frame)).set_artifact("convolve:scale", "!") \
            .morphology("Convolve", 1, kernel).numpy().reshape(frame.shape)
        failures += check("fast convolve 2-D", got, want, 1, detail + " %s c%d" % (kernel, frame.shape[2]))
    elif op == 7:                                  # EXACT 2-D separable kernels (+ the odd centre cell)
        sigma = float(rng.uniform(0.8, 4.5))
        if rng.random() < 0.6:
            got = im.gaussian_blur_image(dev(px), 0.0, sigma).numpy()
            failures += check("exact gaussian", got, ref.gaussian_blur(0.0, sigma).numpy(), 0, detail + " sigma %.3f" % sigma)
        else:
            got = im.sharpen_image(dev(px), 0.0, sigma).numpy()
            failures += check("exact sharpen", got, ref.sharpen(0.0, sigma).numpy(), 0, detail + " sigma %.3f" % sigma)
    elif op == 8:                                  # float Quantum: blur / unsharp on the fp64 kernels
        fpx = float_pixels(rows, cols, kind)
        fref = refmod.RefImage(fpx)
        sigma = float(rng.uniform(0.5, 12.0))
        radius = blur_radius(sigma)
        if rng.random() < 0.5:
            got = im.blur_image(dev_float(fpx), radius, sigma).numpy()
            failures += check_bits("float blur", got, fref.blur(radius, sigma).numpy(), detail + " %gx%.3f" % (radius, sigma))
        else:
            gain, threshold = float(rng.uniform(0.3, 3.0)), float(rng.uniform(0.0, 0.2))
            got = im.unsharp_mask_image(dev_float(fpx), radius, sigma, gain, threshold).numpy()
            failures += check_bits("float unsharp", got, fref.unsharp(radius, sigma, gain, threshold).numpy(),
                                   detail + " %gx%.3f gain %.2f thr %.3f" % (radius, sigma, gain, threshold))
    elif op == 9:                                  # float Quantum: union-of-rectangles Erode / Dilate
        fpx = float_pixels(rows, cols, kind)
        family = ["Disk:%.1f" % rng.uniform(0.5, 16.0), "Square:%d" % rng.integers(1, 9),
                  "Diamond:%d" % rng.integers(1, 12), "Octagon:%d" % rng.integers(1, 10),
                  "Rectangle:%dx%d" % (2 * rng.integers(0, 9) + 1, 2 * rng.integers(0, 9) + 1)]
        kernel = family[int(rng.integers(0, len(family)))]
        method = "Dilate" if rng.random() < 0.5 else "Erode"
        if rng.random() < 0.4:                     # RGB: a fourth, empty channel
            fpx = np.ascontiguousarray(fpx[:, :, :3])
        got = im.morphology_image(dev_float(fpx), method, 1, kernel).numpy().reshape(fpx.shape)
        failures += check_bits("float " + method, got, refmod.RefImage(fpx).morphology(method, 1, kernel).numpy().reshape(fpx.shape),
                               detail + " %s c%d" % (kernel, fpx.shape[2]))
    elif op == 10:                                 # float Quantum: separable 2-D kernels
        fpx = float_pixels(rows, cols, kind)
        sigma = float(rng.uniform(0.8, 4.5))
        if rng.random() < 0.6:
            got = im.gaussian_blur_image(dev_float(fpx), 0.0, sigma).numpy()
            failures += check_bits("float gaussian", got, refmod.RefImage(fpx).gaussian_blur(0.0, sigma).numpy(),
                                   detail + " sigma %.3f" % sigma)
        else:
            got = im.sharpen_image(dev_float(fpx), 0.0, sigma).numpy()
            failures += check_bits("float sharpen", got, refmod.RefImage(fpx).sharpen(0.0, sigma).numpy(),
                                   detail + " sigma %.3f" % sigma)
    elif op == 11:                                 # float Quantum: histogram operators above a megapixel
        rows2, cols2 = int(rng.integers(1000, 1400)), int(rng.integers(1050, 1500))
        fpx = float_pixels(rows2, cols2, int(rng.integers(0, 2)))
        n = rows2 * cols2
        black, white = float(rng.uniform(0, 0.1)) * n, n - float(rng.uniform(0, 0.1)) * n
        if rng.random() < 0.5:
            got = im.contrast_stretch_image(dev_float(fpx), black, white).numpy()
            want = refmod.RefImage(fpx).contrast_stretch(black, white).numpy()
        else:
            got = im.equalize_image(dev_float(fpx)).numpy()
            want = refmod.RefImage(fpx).equalize().numpy()
        failures += check_bits("float histogram op", got, want, "%dx%d" % (rows2, cols2))
    elif op == 12:                                 # integer-cell 2-D convolve (i8 matrix cores), both modes: bit-identical
        choice = int(rng.integers(0, 4))
        if choice == 0:
            kernel = ["Disk:%.1f" % rng.uniform(2.0, 16.4), "Octagon:%d" % rng.integers(2, 16),
                      "Diamond:%d" % rng.integers(2, 16), "Plus:%d" % rng.integers(2, 16),
                      "Ring:%d,%d" % (rng.integers(2, 6), rng.integers(7, 16)),
                      "Rectangle:%dx%d" % (rng.integers(5, 66), rng.integers(2, 9))][int(rng.integers(0, 6))]
        else:
            kw, kh = int(rng.integers(5, 34)), int(rng.integers(5, 12))
            cells = rng.integers(0 if choice < 3 else -9, 10, (kh, kw)).astype(np.float64)
            cells[rng.random((kh, kw)) < 0.1] = np.nan
            if not (np.nansum(np.abs(cells)) > 0) or abs(np.nansum(cells)) < 1:
                cells[kh // 2, kw // 2] = 77.0
            kernel = "%dx%d+%d+%d: %s" % (kw, kh, rng.integers(0, kw), rng.integers(0, kh), " ".join(
                ",".join("nan" if np.isnan(v) else "%d" % v for v in r) for r in cells))
        layout = int(rng.integers(0, 3))               # RGBA alpha-weighted, four plain channels, RGB
        fast = rng.random() < 0.3
        if rng.random() < 0.3:
            # the same on a float-Quantum frame of integer samples (sometimes with one that is not:
            # the generic kernel behind the integer one then does the frame)
            fpx = px.astype(np.float32)
            if rng.random() < 0.25:
                fpx[int(rng.integers(0, rows)), int(rng.integers(0, cols)), int(rng.integers(0, 4))] = \
                    [0.5, -1.0, 65536.0, 12345.678][int(rng.integers(0, 4))]
            if layout == 2:
                fpx = np.ascontiguousarray(fpx[:, :, :3])
            if layout == 1:
                want = np.concatenate([refmod.RefImage(fpx[:, :, c].copy()).set_artifact("convolve:scale", "!")
                                       .morphology("Convolve", 1, kernel).numpy().reshape(rows, cols, 1) for c in range(4)], axis=2)
            else:
                want = refmod.RefImage(fpx).set_artifact("convolve:scale", "!").morphology("Convolve", 1, kernel).numpy()
            got = im.morphology_image(dev_float(fpx, has_alpha=layout == 0) if layout != 2 else dev_float(fpx),
                                      "Convolve", 1, kernel, scale=(1.0, 1)).numpy()
            failures += check_bits("integer convolve 2-D, float frame", got, want, detail + " layout %d %s" % (layout, kernel[:60]))
            return failures
        if layout == 0:
            image, want = dev(px), ref.set_artifact("convolve:scale", "!").morphology("Convolve", 1, kernel).numpy()
        elif layout == 1:
            image = dev(px, has_alpha=False)
            want = np.concatenate([refmod.RefImage(px[:, :, c].copy()).set_artifact("convolve:scale", "!")
                                   .morphology("Convolve", 1, kernel).numpy().reshape(rows, cols, 1) for c in range(4)], axis=2)
        else:
            rgb = np.ascontiguousarray(px[:, :, :3])
            image = dev(rgb)
            want = refmod.RefImage(rgb).set_artifact("convolve:scale", "!").morphology("Convolve", 1, kernel).numpy()
        if fast:
            im.set_precision(im.PRECISION_FAST)
        got = im.morphology_image(image, "Convolve", 1, kernel, scale=(1.0, 1)).numpy()
        im.set_precision(im.PRECISION_EXACT)
        failures += check("integer convolve 2-D", got, want, 1 if fast else 0, detail + " layout %d %s %s" % (
            layout, "fast" if fast else "exact", kernel[:60]))
    elif op == 13:                                 # any-cell 2-D convolve (fused fp64 + tie check): bit-identical
        kw, kh = int(rng.integers(3, 24)), int(rng.integers(3, 16))
        if kw * kh < 25:
            kw, kh = 7, 5
        signed_cells = rng.random() < 0.3
        cells = rng.uniform(-1.0 if signed_cells else 0.0, 1.0, (kh, kw))
        cells[rng.random((kh, kw)) < 0.1] = np.nan
        cells[kh // 2, kw // 2] = 3.0
        kernel = "%dx%d+%d+%d: %s" % (kw, kh, rng.integers(0, kw), rng.integers(0, kh), " ".join(
            ",".join("nan" if np.isnan(v) else "%.17g" % v for v in r) for r in cells))
        channels = int(rng.integers(1, 5))
        blend = channels in (2, 4) and rng.random() < 0.6
        is_float = rng.random() < 0.5
        frame = float_pixels(rows, cols, kind) if is_float else px
        if channels < 4:
            frame = np.ascontiguousarray(frame[:, :, 4 - channels:] if blend else frame[:, :, :channels])
        if blend or channels in (1, 3):
            want = refmod.RefImage(frame).set_artifact("convolve:scale", "!").morphology("Convolve", 1, kernel).numpy()
        else:
            want = np.concatenate([refmod.RefImage(frame[:, :, c].copy()).set_artifact("convolve:scale", "!")
                                   .morphology("Convolve", 1, kernel).numpy().reshape(rows, cols, 1) for c in range(channels)], axis=2)
        image = (dev_float if is_float else dev)(frame, has_alpha=blend)
        got = im.morphology_image(image, "Convolve", 1, kernel, scale=(1.0, 1)).numpy()
        what = detail + " c%d blend=%s %s" % (channels, blend, kernel[:50])
        failures += check_bits("fused 2-D convolve, float", got, want, what) if is_float else \
            check("fused 2-D convolve", got, want, 0, what)
    elif op == 14:                                 # ResizeImage
        filt = ["Lanczos", "Mitchell", "Catrom", "Triangle", "Hermite", "Gaussian", "Spline", "Lanczos2", "Cubic",
                "Box", "Point", "Hann", "Robidoux"][int(rng.integers(0, 13))]
        rows2, cols2 = int(rng.integers(1, 150)), int(rng.integers(1, 200))
        shape = int(rng.integers(0, 4))
        if shape == 0:                             # whole-number horizontal factor, the vertical one at least as large
            f = int(rng.integers(2, 5))
            target = (f * cols2, int(rng.integers(f * rows2, 5 * rows2 + 2)))
        elif shape == 1:                           # any enlargement
            target = (int(rng.integers(cols2, 4 * cols2 + 2)), int(rng.integers(rows2, 4 * rows2 + 2)))
        elif shape == 2:                           # reduction
            target = (int(rng.integers(1, cols2 + 1)), int(rng.integers(1, rows2 + 1)))
        else:                                      # mixed
            target = (int(rng.integers(1, 3 * cols2 + 2)), int(rng.integers(1, 3 * rows2 + 2)))
        is_float = rng.random() < 0.4
        alpha = rng.random() < 0.6
        fast = rng.random() < 0.6
        frame = float_pixels(rows2, cols2, int(rng.integers(0, 6))) if is_float else pixels(rows2, cols2, kind)
        if alpha:
            want = refmod.RefImage(frame).resize(target[0], target[1], filt).numpy()
        else:
            want = np.concatenate([refmod.RefImage(frame[:, :, c].copy()).resize(target[0], target[1], filt).numpy()
                                   .reshape(target[1], target[0], 1) for c in range(4)], axis=2)
        image = (dev_float if is_float else dev)(frame, has_alpha=alpha)
        if fast:
            im.set_precision(im.PRECISION_FAST)
        got = im.resize_image(image, target[0], target[1], filt).numpy()
        im.set_precision(im.PRECISION_EXACT)
        what = "%dx%d -> %dx%d %s %s %s kind %d" % (cols2, rows2, target[0], target[1], filt,
                                                      "alpha" if alpha else "plain", "fast" if fast else "exact", kind)
        if is_float:
            scale = float(np.nanmax(np.abs(np.where(np.isfinite(frame), frame, 0.0)))) if frame.size else 0.0
            failures += check_ulp("float resize", got, want, 1, what, residue=1.0e-9 * scale) if fast else \
                check_bits("float resize", got, want, what)
        else:
            failures += check("resize", got, want, 1 if fast else 0, what)
    elif op == 15:                                 # FAST blur family on every layout
        channels = int(rng.integers(1, 5))
        blend = channels in (2, 4) and rng.random() < 0.6
        frame = np.ascontiguousarray(px[:, :, 4 - channels:] if blend else px[:, :, :channels])
        which = int(rng.integers(0, 3))
        sigma = float(rng.uniform(0.3, 13.4)) if which != 1 else float(rng.uniform(0.8, 4.5))
        gain, threshold = float(rng.uniform(0.3, 3.0)), float(rng.uniform(0.0, 0.2))
        radius = blur_radius(sigma) if which != 1 else 0.0

        def reference(a):
            r = refmod.RefImage(a)
            return (r.blur(radius, sigma) if which == 0 else r.gaussian_blur(0.0, sigma) if which == 1 else
                    r.unsharp(radius, sigma, gain, threshold)).numpy()
        if blend or channels in (1, 3):
            want = reference(frame).reshape(rows, cols, channels)
        else:
            want = np.concatenate([reference(frame[:, :, c].copy()).reshape(rows, cols, 1) for c in range(channels)], axis=2)
        image = dev(frame, has_alpha=blend)
        im.set_precision(im.PRECISION_FAST)
        got = (im.blur_image(image, radius, sigma) if which == 0 else im.gaussian_blur_image(image, 0.0, sigma) if which == 1 else
               im.unsharp_mask_image(image, radius, sigma, gain, threshold)).numpy().reshape(rows, cols, channels)
        im.set_precision(im.PRECISION_EXACT)
        what = detail + " %s c%d blend=%s %gx%.3f" % (("blur", "gaussian", "unsharp")[which], channels, blend, radius, sigma)
        if which == 2:
            blurred = (refmod.RefImage(frame).blur(radius, sigma).numpy().reshape(rows, cols, channels).astype(np.int64)
                       if (blend or channels in (1, 3)) else None)
            d = np.abs(got.astype(np.int64) - want.astype(np.int64))
            if blurred is not None:
                edge = np.abs(2 * np.abs(frame.astype(np.int64) - blurred) - 65535.0 * threshold) <= 2.0
                d[edge] = 0
                failures += check("fast unsharp, layout", d, np.zeros_like(d), int(np.ceil(1.0 + gain)), what)
        else:
            failures += check("fast blur, layout", got, want, 1, what)
    else:                                          # FAST Lab
        im.set_precision(im.PRECISION_FAST)
        d2 = dev(px)
        im.transform_image_colorspace(d2, "Lab")
        im.set_precision(im.PRECISION_EXACT)
        failures += check("fast lab", d2.numpy(), ref.colorspace("Lab").numpy(), 1, detail)
    return failures


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    setup(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    only_ops = [int(t) for t in os.environ.get("STRESS_OPS", "").split(",") if t.strip()]
    t0 = time.time()
    cases = failures = 0
    while time.time() - t0 < budget:
        failures += run_case(only_ops[int(rng.integers(0, len(only_ops)))] if only_ops else None)
        cases += 1
    print("%d cases, %d failures, %.0f s" % (cases, failures, time.time() - t0))
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
