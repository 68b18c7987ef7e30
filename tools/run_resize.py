#!/usr/bin/env python3
"""Runs only the C3 workload (8192^2 -> 32768^2 Lanczos, float Quantum RGBA) a few times;
used under rocprofv3.   python tools/run_resize.py [fast|exact] [reps] [size]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagemagick_amd as im

prec = sys.argv[1] if len(sys.argv) > 1 else "fast"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
im.load()
im.set_precision(im.PRECISION_FAST if prec == "fast" else im.PRECISION_EXACT)
gen = torch.Generator(device="cuda").manual_seed(1)
src = torch.rand((m, m, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
img = im.Image(src)
out = im.resize_image(img, 4 * m, 4 * m, "Lanczos")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = None
    out = im.resize_image(img, 4 * m, 4 * m, "Lanczos")
torch.cuda.synchronize()
sec = (time.perf_counter() - t0) / reps
print("resize %dx%d -> x4 %s: %.3f ms  %.1f Mpixels/s out" % (m, m, prec, sec * 1e3, 16.0 * m * m / sec / 1e6))

from bench import kernel_profile
hold = {}


def call():
    hold["o"] = None
    hold["o"] = im.resize_image(img, 4 * m, 4 * m, "Lanczos")


prof = kernel_profile(im, call, 2)
print("   kernels(ms):", {k: round(v["avg_ms"], 3) for k, v in prof.items()})
