#!/usr/bin/env python3
"""Times the pointwise / histogram operators alone on a 4096^2 RGBA Q16 image with HIP events
(20 calls each, device-resident).   python tools/time_pointwise.py [edge]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagemagick_amd as im

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
im.load()
gen = torch.Generator(device="cuda").manual_seed(1)
src = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)


def timeit(name, fn, reps=20, bytes_moved=None):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    extra = "" if bytes_moved is None else "  %.0f GB/s" % (bytes_moved / ms / 1e6)
    print("%-28s %.4f ms%s" % (name, ms, extra), flush=True)


img = im.Image(src.clone())
rw = 2.0 * n * n * 8


def cs(a, b):
    def f():
        img.colorspace = a.lower()
        im.transform_image_colorspace(img, b)
    return f


from imagemagick_amd import _lib
for a, b in (("sRGB", "Lab"), ("sRGB", "XYZ"), ("sRGB", "RGB"), ("RGB", "sRGB"), ("Lab", "sRGB"), ("XYZ", "sRGB")):
    timeit("%s->%s" % (a, b), cs(a, b), bytes_moved=rw)
img.colorspace = "srgb"
timeit("equalize", lambda: im.equalize_image(img), bytes_moved=1.5 * rw)
npx = n * n
timeit("contrast_stretch (Lab)", lambda: (setattr(img, "colorspace", "lab"),
                                          im.contrast_stretch_image(img, 0.02 * npx, npx - 0.01 * npx)),
       bytes_moved=1.5 * rw)
img.colorspace = "srgb"
timeit("grayscale rec709luma", lambda: im.grayscale_image(img, "Rec709Luma"), bytes_moved=rw)
timeit("function polynomial", lambda: im.function_image(img, "Polynomial", (0.3, -1.2, 1.5, 0.1)), bytes_moved=rw)
timeit("clone (copy)", lambda: src.clone(), bytes_moved=rw)
