"""Float-Quantum BlurImage(0xsigma) and UnsharpMask on n^2 frames (RGBA, RGB): ms per call and per kernel.
    python tools/time_blur_float.py [n] [sigma,sigma,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sigmas = tuple(float(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (10.0, 2.0)
g = torch.Generator(device="cuda").manual_seed(3)
for channels, alpha, label in ((4, True, "rgba"), (3, False, "rgb")):
    a = torch.rand((n, n, channels), generator=g, device="cuda", dtype=torch.float32) * 65535.0
    image = im.Image(a, has_alpha=alpha)
    out = image.like()
    for sigma in sigmas:
        for mode, precision in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
            im.set_precision(precision)
            f = lambda: im.blur_image(image, 0.0, sigma, out=out)
            for _ in range(5):
                f()
            sec = timed(torch, f, 20)
            prof = kernel_profile(im, f, 10)
            print("%-5s sigma %-4g %-5s %.4f ms  kernels %s" % (label, sigma, mode, sec * 1e3,
                  {k: "%.4f [%.4f]" % (v["avg_ms"], v["min_ms"]) for k, v in prof.items()}), flush=True)
