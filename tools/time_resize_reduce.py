"""Reductions: 8192^2 -> 2048^2 Lanczos (Q16, float; FAST, EXACT), the reference's own device benchmark geometry
(2048x1536 -> 640x480) and a 2x / 3x reduction — ms per call and the kernels' hipEvent times.
    python tools/time_resize_reduce.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
g = torch.Generator(device="cuda").manual_seed(3)
for (h, w), (th, tw) in (((8192, 8192), (2048, 2048)), ((8192, 8192), (4096, 4096)), ((8192, 8192), (2731, 2731)),
                         ((1536, 2048), (480, 640))):
    for label in ("q16", "hdri"):
        if label == "q16":
            a = torch.randint(-32768, 32768, (h, w, 4), generator=g, device="cuda", dtype=torch.int16).view(torch.uint16)
        else:
            a = torch.rand((h, w, 4), generator=g, device="cuda", dtype=torch.float32) * 65535.0
        image = im.Image(a)
        for mode, precision in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
            im.set_precision(precision)
            f = lambda: im.resize_image(image, tw, th, "Lanczos")
            sec = timed(torch, f, 10)
            prof = kernel_profile(im, f, 3)
            print("%dx%d -> %dx%d %-4s %-5s %.4f ms  %s" % (w, h, tw, th, label, mode, sec * 1e3,
                  {k: round(v["avg_ms"], 4) for k, v in prof.items()}), flush=True)
