"""FAST / EXACT Convolve with small outer-product kernels (3 x 3 ... 9 x 9) on a 4096^2 frame, with and without the separated
route (MAGICKHIP_NO_SEPARABLE): ms per call and the kernels that ran — where the two passes stop paying
(operators.cpp separable_convolve).   python tools/probe_small_separable.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
n = 4096
gen = torch.Generator(device="cuda").manual_seed(3)
q4 = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
K = {"3x3": "3x3: 1,2,1 2,4,2 1,2,1", "5x5": "5x5: 1,4,6,4,1 4,16,24,16,4 6,24,36,24,6 4,16,24,16,4 1,4,6,4,1",
     "7x7": "Gaussian:3x1.2", "9x9": "Gaussian:4x1.5", "3x3 box": "Square:1", "5x5 box": "Square:2"}
for label, px, alpha in (("rgba", q4, True), ("plain4", q4, False), ("rgb", q4[:, :, :3].contiguous(), False), ("gray", q4[:, :, :1].contiguous(), False)):
    img = im.Image(px, has_alpha=alpha)
    for kname, kernel in K.items():
        row = []
        for mode, prec in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
            im.set_precision(prec)
            for sep in (None, "1"):
                im.set_option("MAGICKHIP_NO_SEPARABLE", sep)
                f = lambda: im.morphology_image(img, "Convolve", 1, kernel, scale=(1.0, 1))
                for _ in range(3):
                    f()
                sec = timed(torch, f, 10)
                prof = kernel_profile(im, f, 1)
                row.append("%s%s %.3f %s" % (mode, "" if sep is None else " nosep", sec * 1e3, "+".join(sorted(prof))))
        print("%-6s %-8s %s" % (label, kname, " | ".join(row)), flush=True)
im.set_option("MAGICKHIP_NO_SEPARABLE", None)
