import os, sys
sys.path.insert(0, os.getcwd())
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
n = 4096
gen = torch.Generator(device="cuda").manual_seed(3)
q16 = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(q16)
hold = {}
for sigma in (1.0, 2.0, 3.0, 5.0):
    for rmin, cmin in ((9, 9), (9, 99), (99, 9), (99, 99)):
        im.set_option("MAGICKHIP_FOLD_R8_ROW_MIN", str(rmin)); im.set_option("MAGICKHIP_FOLD_R8_COLUMN_MIN", str(cmin))
        def f():
            hold["o"] = None
            hold["o"] = im.gaussian_blur_image(img, 0.0, sigma)
        sec = timed(torch, f, 3)
        prof = kernel_profile(im, f, 2)
        print("sigma", sigma, "row R8>=%d col R8>=%d" % (rmin, cmin), "%.3f ms" % (sec * 1e3), {k: round(v["avg_ms"], 3) for k, v in prof.items()}, flush=True)
