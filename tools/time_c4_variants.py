"""sRGB->Lab on a 4096^2 RGBA Q16 frame: the 512 KB decode table (three 8-byte gathers per
pixel) against the in-register Chebyshev decode (MAGICKHIP_NO_COLOR_TABLES=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile
im.load()
n = 4096
gen = torch.Generator(device="cuda").manual_seed(5)
src = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
work = src.clone()
for label, env in (("decode table", None), ("Chebyshev in registers", "1")):
    im.set_option("MAGICKHIP_NO_COLOR_TABLES", env)      # (the library reads the environment once, at start-up)

    def f():
        work.copy_(src)
        img = im.Image(work)
        im.transform_image_colorspace(img, "Lab")
    for _ in range(3):
        f()
    prof = kernel_profile(im, f, 5)
    print(label, {k: round(v["avg_ms"], 4) for k, v in prof.items()}, flush=True)
