#!/usr/bin/env python3
"""FAST ResizeImage (Lanczos, RGBA) by whole-number factors, Q16 and float frames: kernel times.
   python tools/time_resize_factors.py [out_size] [filter]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagemagick_amd as im
from bench import kernel_profile

out = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
filt = sys.argv[2] if len(sys.argv) > 2 else "Lanczos"
im.load()
im.set_precision(im.PRECISION_FAST)
gen = torch.Generator(device="cuda").manual_seed(1)
for factor in (2, 3, 4):
    m = out // factor
    for dtype in ("q16", "float"):
        if dtype == "q16":
            src = torch.randint(0, 65536, (m, m, 4), generator=gen, device="cuda", dtype=torch.int32).to(torch.uint16)
        else:
            src = torch.rand((m, m, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
        img = im.Image(src)
        hold = {}

        def call():
            hold["o"] = None
            hold["o"] = im.resize_image(img, factor * m, factor * m, filt)

        call()
        torch.cuda.synchronize()
        prof = kernel_profile(im, call, 3)
        print("x%d %-5s %5d -> %5d %s:" % (factor, dtype, m, factor * m, filt),
              {k: round(v["avg_ms"], 3) for k, v in prof.items()}, flush=True)
        hold.clear()
        del img, src
        torch.cuda.empty_cache()
