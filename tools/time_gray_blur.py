"""One-channel (gray) Q16 BlurImage / UnsharpMaskImage / Erode / Dilate: the frame's own kernels against the four-row-band
form (operators.cpp fused_blur_gray_bands, morphology.hip try_rects_gray_bands), ms per call, FAST and EXACT.
    python tools/time_gray_blur.py [sigma,sigma,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
sigmas = tuple(float(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (10.0, 2.0)
g = torch.Generator(device="cuda").manual_seed(3)
for n in (512, 1024, 1448, 2048, 4096, 8192):
    a = torch.randint(-32768, 32768, (n, n, 1), generator=g, device="cuda", dtype=torch.int16).view(torch.uint16)
    image = im.Image(a)
    out = image.like()
    for sigma in sigmas:
        for op in ("blur", "unsharp"):
            row = []
            for mode, precision in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
                im.set_precision(precision)
                for bands in (False, True):
                    im.set_option("MAGICKHIP_GRAY_BANDS_MIN_PIXELS", "0")
                    im.set_option("MAGICKHIP_NO_GRAY_BANDS", None if bands else "1")
                    if op == "blur":
                        f = lambda: im.blur_image(image, 0.0, sigma, out=out)
                    else:
                        f = lambda: im.unsharp_mask_image(image, 0.0, sigma, 1.0, 0.02)
                    for _ in range(5):
                        f()
                    sec = timed(torch, f, 20)
                    row.append("%s %s %.4f" % (mode, "bands" if bands else "passes", sec * 1e3))
            print("%5d^2 sigma %-4g %-7s %s" % (n, sigma, op, "  ".join(row)), flush=True)
    for method, kernel in (("Dilate", "Disk:15"), ("Erode", "Disk:5"), ("Dilate", "Square:1"), ("Open", "Disk:5")):
        row = []
        for bands in (False, True):
            im.set_option("MAGICKHIP_GRAY_BANDS_MIN_PIXELS", "0")
            im.set_option("MAGICKHIP_NO_GRAY_BANDS", None if bands else "1")
            f = lambda: im.morphology_image(image, method, 1, kernel)
            for _ in range(3):
                f()
            row.append("%s %.4f" % ("bands" if bands else "own  ", timed(torch, f, 10) * 1e3))
        print("%5d^2 %-6s %-9s %s" % (n, method, kernel, "  ".join(row)), flush=True)
    for name, op in (("gaussian 0x3", lambda: im.gaussian_blur_image(image, 0.0, 3.0)),
                     ("sharpen 0x2", lambda: im.sharpen_image(image, 0.0, 2.0)),
                     ("convolve Disk:5", lambda: im.morphology_image(image, "Convolve", 1, "Disk:5", scale=(1.0, 1))),
                     ("convolve 3x3", lambda: im.morphology_image(image, "Convolve", 1, "3x3: 1,2,1 2,4,2 1,2,1", scale=(1.0, 1))),
                     ("convolve LoG:0x2", lambda: im.morphology_image(image, "Convolve", 1, "LoG:0x2"))):
        row = []
        for mode, precision in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
            im.set_precision(precision)
            for bands in (False, True):
                im.set_option("MAGICKHIP_GRAY_BANDS_MIN_PIXELS", "0")
                im.set_option("MAGICKHIP_NO_GRAY_BANDS", None if bands else "1")
                for _ in range(3):
                    op()
                row.append("%s %s %.4f" % (mode, "bands" if bands else "own", timed(torch, op, 10) * 1e3))
        print("%5d^2 %-16s %s" % (n, name, "  ".join(row)), flush=True)
    im.set_option("MAGICKHIP_NO_GRAY_BANDS", None)
    del image, out, a
    torch.cuda.empty_cache()
