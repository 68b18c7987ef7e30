#!/usr/bin/env python3
"""How much of a FAST whole-number enlargement goes down the careful kernel, per factor and input kind (Q16).
    python tools/probe_resize_careful.py [size]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagemagick_amd as im
from bench import kernel_profile

m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
im.load()
im.set_precision(im.PRECISION_FAST)
gen = torch.Generator(device="cuda").manual_seed(1)
base = torch.randint(-32768, 32768, (m, m, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
ramp = ((torch.arange(m, device="cuda").view(m, 1, 1) * 7 + torch.arange(m, device="cuda").view(1, m, 1) * 3 +
         torch.arange(4, device="cuda").view(1, 1, 4) * 1000) % 65536).to(torch.int32)
ramp = (ramp - 65536 * (ramp >= 32768)).to(torch.int16).view(torch.uint16).contiguous()
opaque = base.clone()
opaque.view(torch.int16)[:, :, 3] = -1
kinds = (("random rgba", base, True), ("opaque alpha", opaque, True), ("plain4", base, False), ("ramp rgba", ramp, True))
for f in (2, 3, 4):
    for label, px, alpha in kinds:
        img = im.Image(px, has_alpha=alpha)
        hold = {}

        def call():
            hold["o"] = None
            hold["o"] = im.resize_image(img, f * m, f * m, "Lanczos")

        call()
        prof = kernel_profile(im, call, 2)
        print("x%d %-13s %s" % (f, label, {k: round(v["avg_ms"], 3) for k, v in prof.items()}), flush=True)
