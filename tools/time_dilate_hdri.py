import os, sys
sys.path.insert(0, os.getcwd())
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
n = 16384
gen = torch.Generator(device="cuda").manual_seed(3)
a = torch.rand((n, n, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
img = im.Image(a)
hold = {}
for kernel in ("Disk:15", "Disk:7", "Square:3"):
    def f():
        hold["o"] = None
        hold["o"] = im.morphology_image(img, "Dilate", 1, kernel)
    sec = timed(torch, f, 2)
    prof = kernel_profile(im, f, 2)
    print(kernel, "%.3f ms" % (sec * 1e3), {k: round(v["avg_ms"], 3) for k, v in prof.items()}, flush=True)
