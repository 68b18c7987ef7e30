"""FAST BlurImage(0,10) / UnsharpMask on 8192^2 RGB Q16 (6-byte pixels): single fused launch
against the two-launch matrix-core form (MAGICKHIP_NO_FUSED_BLUR=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
im.set_precision(im.PRECISION_FAST)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
gen = torch.Generator(device="cuda").manual_seed(3)
a = torch.randint(-32768, 32768, (n, n, 3), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(a)
hold = {}
for label, env in (("one launch", None), ("two launches", "1")):
    im.set_option("MAGICKHIP_NO_FUSED_BLUR", env)      # (the library reads the environment once, at start-up)
    for name, fn in (("blur", lambda: hold.update(o=im.blur_image(img, 0.0, 10.0))),
                     ("unsharp", lambda: hold.update(o=im.unsharp_mask_image(img, 0.0, 10.0, 1.0, 0.02)))):
        sec = timed(torch, fn, 5)
        prof = kernel_profile(im, fn, 3)
        print("RGB %-8s %-13s %.3f ms  %.1f Mpixels/s  kernels(ms) %s" % (name, label, sec * 1e3, n * n / sec / 1e6,
              {k: round(v["avg_ms"], 3) for k, v in prof.items()}), flush=True)
