"""GaussianBlurImage in EXACT mode (Q16) and on a float frame: the outer-product kernel as two
folded fp64 passes against round 3's four launches."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
L = im.load()
n = 8192
gen = torch.Generator(device="cuda").manual_seed(3)
flt = torch.rand((n, n, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
q16 = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
hold = {}
for label, px in (("q16", q16), ("hdri", flt)):
    img = im.Image(px)
    for sigma, fold in ((1.0, True), (1.0, False), (3.0, True), (3.0, False), (10.0, True), (10.0, False)):
        im.set_option("MAGICKHIP_NO_SEPARABLE_FOLD", None if fold else "1")
        def f():
            hold["o"] = None
            hold["o"] = im.gaussian_blur_image(img, 0.0, sigma)
        f(); torch.cuda.synchronize()
        L.MhSeparableRecomputed(1)
        f(); torch.cuda.synchronize()
        count = L.MhSeparableRecomputed(0)
        sec = timed(torch, f, 2)
        prof = kernel_profile(im, f, 1)
        print(label, "sigma", sigma, "folded" if fold else "4 launches", "%.3f ms" % (sec * 1e3), "recomputed", count, "of", n * n * 4, {k: round(v["avg_ms"], 2) for k, v in prof.items()}, flush=True)
