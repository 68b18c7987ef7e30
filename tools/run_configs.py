#!/usr/bin/env python3
"""Times the other BASELINE.json configs on one GPU (kernel profile via the library's
hipEvent records):  C4 = 4096^2 RGBA Q16 sRGB->Lab + ContrastStretch 2%x1%;
C5 = 16384^2 RGBA Q16 Dilate Disk:15 + UnsharpMask(0x10+1+0.02)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed

im.load()
which = sys.argv[1] if len(sys.argv) > 1 else "c4,c5"
prec = sys.argv[2] if len(sys.argv) > 2 else "fast"
im.set_precision(im.PRECISION_FAST if prec == "fast" else im.PRECISION_EXACT)
gen = torch.Generator(device="cuda").manual_seed(5)

if "c4" in which:
    n = 4096
    src = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)

    one_call = os.environ.get("C4_TWO_CALLS") is None

    def c4():
        img = im.Image(src.clone())
        if one_call:
            im.transform_colorspace_contrast_stretch_image(img, "Lab", 0.02 * n * n, n * n - 0.01 * n * n)
        else:
            im.transform_image_colorspace(img, "Lab")
            im.contrast_stretch_image(img, 0.02 * n * n, n * n - 0.01 * n * n)
    sec = timed(torch, c4, 5)
    prof = kernel_profile(im, c4, 3)
    print("C4 one 4096^2 image: %.3f ms  %.1f Mpixels/s  kernels(ms): %s" % (
        sec * 1e3, n * n / sec / 1e6, {k: round(v["avg_ms"], 3) for k, v in prof.items()}))

if "c5" in which:
    n = int(os.environ.get("C5_SIZE", "16384"))
    src = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
    img = im.Image(src)
    hold = {}

    def dilate():
        hold["o"] = im.morphology_image(img, "Dilate", 1, "Disk:15")
    sec = timed(torch, dilate, 2)
    prof = kernel_profile(im, dilate, 2)
    print("C5 Dilate Disk:15 %d^2: %.3f ms  %.1f Mpixels/s  kernels(ms): %s" % (
        n, sec * 1e3, n * n / sec / 1e6, {k: round(v["avg_ms"], 3) for k, v in prof.items()}))

    def convolve():
        hold["o"] = im.morphology_image(img, "Convolve", 1, "Disk:15", scale=(1.0, 1))
    sec = timed(torch, convolve, 2)
    prof = kernel_profile(im, convolve, 2)
    print("C5 Convolve Disk:15 %d^2: %.3f ms  %.1f Mpixels/s  kernels(ms): %s" % (
        n, sec * 1e3, n * n / sec / 1e6, {k: round(v["avg_ms"], 3) for k, v in prof.items()}))

    def unsharp():
        hold["o"] = im.unsharp_mask_image(img, 0.0, 10.0, 1.0, 0.02)
    sec = timed(torch, unsharp, 2)
    prof = kernel_profile(im, unsharp, 2)
    print("C5 UnsharpMask(0x10) %d^2: %.3f ms  %.1f Mpixels/s  kernels(ms): %s" % (
        n, sec * 1e3, n * n / sec / 1e6, {k: round(v["avg_ms"], 3) for k, v in prof.items()}))
