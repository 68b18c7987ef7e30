"""FAST / EXACT BlurImage(0xsigma) on n^2 frames: RGBA, four plain channels, RGB — ms per call (hipEvent kernel time);
UnsharpMaskImage(0xsigma+1+0.02) on the RGBA frame.
    python tools/time_blur_quick.py [n] [sigma,sigma,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
REPS = int(os.environ.get("TBQ_REPS", "5"))          # launches behind every kernel average (min in brackets)
sigmas = tuple(float(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else (10.0, 5.0, 2.0)
g = torch.Generator(device="cuda").manual_seed(3)
for channels, alpha, label in ((4, True, "rgba"), (4, False, "plain4"), (3, False, "rgb")):
    a = torch.randint(-32768, 32768, (n, n, channels), generator=g, device="cuda", dtype=torch.int16).view(torch.uint16)
    image = im.Image(a, has_alpha=alpha)
    out = image.like()
    for sigma in sigmas:
        for mode, precision in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
            im.set_precision(precision)
            f = lambda: im.blur_image(image, 0.0, sigma, out=out)
            for _ in range(10):
                f()
            sec = timed(torch, f, 30)
            prof = kernel_profile(im, f, REPS)
            print("%-6s sigma %-4g %-5s %.4f ms  kernels %s" % (label, sigma, mode, sec * 1e3,
                  {k: "%.4f [%.4f]" % (v["avg_ms"], v["min_ms"]) for k, v in prof.items()}), flush=True)
    if alpha:
        im.set_precision(im.PRECISION_FAST)
        for sigma in sigmas:
            f = lambda: im.unsharp_mask_image(image, 0.0, sigma, 1.0, 0.02)
            for _ in range(5):
                f()
            sec = timed(torch, f, 20)
            prof = kernel_profile(im, f, REPS)
            print("%-6s sigma %-4g %-5s %.4f ms  kernels %s" % (label, sigma, "unsharp", sec * 1e3,
                  {k: "%.4f [%.4f]" % (v["avg_ms"], v["min_ms"]) for k, v in prof.items()}), flush=True)
