"""ConvolveMorphology Disk:15 on a float-Quantum 16384^2 RGBA frame: integer samples (what a 16-bit
file decodes to) take the exact-integer i8 kernel, a frame with one fractional sample falls back to
the generic kernel (timed on 4096^2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
gen = torch.Generator(device="cuda").manual_seed(3)
hold = {}
for n, spoil in ((16384, False), (4096, False), (4096, True)):
    a = torch.randint(0, 65536, (n, n, 4), generator=gen, device="cuda", dtype=torch.int32).to(torch.float32)
    if spoil:
        a[n // 2, n // 2, 1] += 0.5
    img = im.Image(a)

    def f():
        hold["o"] = None
        hold["o"] = im.morphology_image(img, "Convolve", 1, "Disk:15", scale=(1.0, 1))
    sec = timed(torch, f, 2)
    prof = kernel_profile(im, f, 2)
    print("float RGBA %5d^2 Disk:15 %-22s %.3f ms  %.1f Mpixels/s  kernels(ms) %s" % (
        n, "one fractional sample" if spoil else "integer samples", sec * 1e3, n * n / sec / 1e6,
        {k: round(v["avg_ms"], 3) for k, v in prof.items()}), flush=True)
    del img, a
    hold.clear()
    torch.cuda.empty_cache()
