#!/usr/bin/env python3
"""Timing sweep over operator parameters, FAST and EXACT, to find cliffs — configurations where a route change makes
the call many times slower than its neighbours (how round 6 found the 3x enlargement of Q16 frames with varying alpha at
15 ms beside 2x at 1.9 and 4x at 5.0).   python tools/sweep_cliffs.py [resize|blur|morph] [size]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed

what = sys.argv[1] if len(sys.argv) > 1 else "resize"
m = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
im.load()
gen = torch.Generator(device="cuda").manual_seed(5)


def frame(channels, is_float):
    if is_float:
        return torch.rand((m, m, channels), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
    return torch.randint(-32768, 32768, (m, m, channels), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)


def both_modes(call):
    out = []
    for precision in (im.PRECISION_FAST, im.PRECISION_EXACT):
        im.set_precision(precision)
        try:
            call()
            sec = timed(torch, call, 3)
            prof = kernel_profile(im, call, 1)
            out.append((sec * 1e3, " ".join("%s=%.3f" % (k.replace("resize_", ""), v["avg_ms"]) for k, v in sorted(prof.items()))))
        except Exception as exc:
            out.append((float("nan"), "%s" % type(exc).__name__))
    im.set_precision(im.PRECISION_FAST)
    return out


if what == "resize":
    for is_float in (False, True):
        for channels, alpha in ((4, True), (4, False), (3, False), (1, False)):
            img = im.Image(frame(channels, is_float), has_alpha=alpha)
            for filt in ("Lanczos", "Mitchell", "Triangle", "Box"):
                for tx, ty in ((2.0, 2.0), (3.0, 3.0), (4.0, 4.0), (1.5, 1.5), (2.0, 3.0), (3.0, 2.0), (0.5, 0.5), (0.3, 0.3), (1.0, 2.0), (2.0, 1.0)):
                    w, h = int(m * tx), int(m * ty)
                    hold = {}

                    def call():
                        hold["o"] = None
                        hold["o"] = im.resize_image(img, w, h, filt)

                    (fast, fk), (exact, ek) = both_modes(call)
                    flag = "  <<<" if fast > 1.5 * exact else ""
                    print("%-5s c%d %-5s %-8s x%.1f,%.1f  fast %8.3f ms  exact %8.3f ms  ns/outpx %6.3f  [%s | %s]%s" % (
                        "float" if is_float else "q16", channels, "alpha" if alpha else "plain", filt, tx, ty, fast, exact,
                        fast * 1e6 / (w * h), fk, ek, flag), flush=True)
elif what == "blur":
    for is_float in (False, True):
        for channels, alpha in ((4, True), (4, False), (3, False), (2, True), (1, False)):
            img = im.Image(frame(channels, is_float), has_alpha=alpha)
            for radius, sigma in ((0, 0.5), (0, 1), (0, 2), (0, 3), (0, 5), (0, 8), (0, 10), (0, 10.3), (0, 12), (0, 20), (30, 2), (4, 10), (0, 40)):
                def call():
                    im.blur_image(img, float(radius), float(sigma))

                (fast, fk), (exact, ek) = both_modes(call)
                flag = "  <<<" if fast > 1.2 * exact else ""
                print("%-5s c%d %-5s blur %gx%-5g fast %8.3f ms  exact %8.3f ms  [%s | %s]%s" % (
                    "float" if is_float else "q16", channels, "alpha" if alpha else "plain", radius, sigma, fast, exact, fk, ek, flag), flush=True)
