#!/usr/bin/env python3
"""FAST whole-number enlargements: rows a wave walks (MAGICKHIP_RESIZE_STREAM_ROWS) against frame size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile
im.load()
im.set_option("MAGICKHIP_RESIZE_ONE_LAUNCH_MIN_PIXELS", "0" if len(sys.argv) <= 1 else None)   # (the rows sweep: every size through the walk)
FACTORS = tuple(float(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (2, 3, 4)
ROWS = (128, 64, 32, 16) if len(sys.argv) <= 1 else (0,)      # 0: the library picks
gen = torch.Generator(device="cuda").manual_seed(5)
for m in (512, 1024, 2048, 4096, 8192):
    for is_float in (False, True):
        if is_float:
            px = torch.rand((m, m, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
        else:
            px = torch.randint(-32768, 32768, (m, m, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
        img = im.Image(px, has_alpha=True)
        for f in FACTORS:
            if f * m > 24576:
                continue
            hold = {}

            def call():
                hold["o"] = None
                hold["o"] = im.resize_image(img, int(f * m), int(f * m), "Lanczos")

            line = []
            im.set_precision(im.PRECISION_EXACT)
            call()
            prof = kernel_profile(im, call, 2)
            line.append("two passes %.3f" % sum(v["avg_ms"] for v in prof.values()))
            im.set_precision(im.PRECISION_FAST)
            for rows in ROWS:
                im.set_option("MAGICKHIP_RESIZE_STREAM_ROWS", str(rows))
                call()
                prof = kernel_profile(im, call, 2)
                line.append("rows %d: %.3f" % (rows, sum(v["avg_ms"] for v in prof.values())))
            print("%5d^2 %-5s x%g  %s" % (m, "float" if is_float else "q16", f, "   ".join(line)), flush=True)
        del img, px
        hold = {}
        torch.cuda.empty_cache()
