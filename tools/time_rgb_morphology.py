"""Erode / Dilate on RGBA, RGB (6-byte pixels: padded to four channels, morphology.hip try_rects_rgb_padded) and two plain
channels, Q16 and float RGB, ms per call and the kernels that ran.   python tools/time_rgb_morphology.py"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
gen = torch.Generator(device="cuda").manual_seed(3)
for n in (256, 512, 1024, 2048, 8192):
    q4 = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
    f3 = torch.rand((n, n, 3), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
    for label, px, alpha, pad in (("rgba", q4, True, None), ("rgb", q4[:, :, :3].contiguous(), False, True),
                                  ("rgb own", q4[:, :, :3].contiguous(), False, False),
                                  ("rgb flt", f3, False, True), ("rgb flt own", f3, False, False)):
        img = im.Image(px, has_alpha=alpha)
        im.set_option("MAGICKHIP_RGB_PAD_MIN_PIXELS", "0")
        im.set_option("MAGICKHIP_NO_RGB_PAD", None if pad in (None, True) else "1")
        for method, kernel in (("Dilate", "Disk:15"), ("Erode", "Disk:5"), ("Dilate", "Square:1")):
            f = lambda: im.morphology_image(img, method, 1, kernel)
            for _ in range(3):
                f()
            sec = timed(torch, f, 5)
            prof = kernel_profile(im, f, 1)
            print("%5d^2 %-11s %-6s %-9s %8.4f ms %s" % (n, label, method, kernel, sec * 1e3, {k: round(v["avg_ms"], 3) for k, v in prof.items()}), flush=True)
