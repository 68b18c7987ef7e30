"""Diagnostic (library built with -DMH_HYBRID_KNOCK): what each part of the hybrid FAST blur's alpha
path costs.  One line per knock mask, 8192^2 RGBA sigma 10."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import timed
im.load()
n = 8192
gen = torch.Generator(device="cuda").manual_seed(3)
a = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(a)
out = img.like()
im.set_precision(im.PRECISION_FAST)
for mask in [int(m) for m in os.environ.get("KNOCK_MASKS", "0,1,31,63,95,159,255").split(",")]:
    im.set_option("MAGICKHIP_HYBRID_KNOCK", str(mask))
    f = lambda: im.blur_image(img, 0.0, 10.0, out=out)
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    print("knock %2d: %.4f ms" % (mask, timed(torch, f, 40) * 1e3), flush=True)
plain = im.Image(a, has_alpha=False)
im.set_option("MAGICKHIP_HYBRID_KNOCK", None)
f = lambda: im.blur_image(plain, 0.0, 10.0, out=out)
for _ in range(20):
    f()
print("plain4: %.4f ms" % (timed(torch, f, 40) * 1e3))
