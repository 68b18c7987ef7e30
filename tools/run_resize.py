#!/usr/bin/env python3
"""Runs only the C3 workload (8192^2 -> 32768^2 Lanczos, float Quantum RGBA) a few times;
used under rocprofv3.   python tools/run_resize.py [fast|exact] [reps] [size] [factor] [float|q16]"""
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
f = int(sys.argv[4]) if len(sys.argv) > 4 else 4
q16 = len(sys.argv) > 5 and sys.argv[5] == "q16"
im.load()
im.set_precision(im.PRECISION_FAST if prec == "fast" else im.PRECISION_EXACT)
gen = torch.Generator(device="cuda").manual_seed(1)
if q16:
    src = torch.randint(-32768, 32768, (m, m, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
else:
    src = torch.rand((m, m, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
img = im.Image(src)
out = im.resize_image(img, f * m, f * m, "Lanczos")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    out = None
    out = im.resize_image(img, f * m, f * m, "Lanczos")
torch.cuda.synchronize()
sec = (time.perf_counter() - t0) / reps
print("resize %dx%d -> x%d %s %s: %.3f ms  %.1f Mpixels/s out" % (m, m, f, "q16" if q16 else "float", prec, sec * 1e3, float(f * f) * m * m / sec / 1e6))

from bench import kernel_profile
hold = {}


def call():
    hold["o"] = None
    hold["o"] = im.resize_image(img, f * m, f * m, "Lanczos")


prof = kernel_profile(im, call, 2)
print("   kernels(ms):", {k: round(v["avg_ms"], 3) for k, v in prof.items()})
