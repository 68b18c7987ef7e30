"""Round 4: the FAST BlurImage forms side by side on n^2 RGBA Q16 — hybrid (f16 colour + exact
alpha, convolve_fused_hybrid.hip), round 3's exact row pass + f16 column pass (MAGICKHIP_NO_HYBRID),
and EXACT (bit-identical) — with the differences of the FAST results against EXACT.
    python tools/time_blur_r4.py [n] [sigma]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
lib = im.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
gen = torch.Generator(device="cuda").manual_seed(3)


def frame(kind, channels=4):
    a = torch.randint(-32768, 32768, (n, n, channels), generator=gen, device="cuda", dtype=torch.int16)
    if kind == "opaque":
        a[:, :, 3] = -1
    elif kind == "tiny":
        a[:, :, 3] = torch.randint(0, 4, (n, n), generator=gen, device="cuda", dtype=torch.int16)
        a[n // 4:n // 2, :, 3] = 0
    return a.view(torch.uint16)


def run(label, image, precision, opts, reps=40):
    for k, v in opts.items():
        im.set_option(k, v)
    im.set_precision(precision)
    out = image.like()

    def f():
        im.blur_image(image, 0.0, sigma, out=out)
    try:
        for _ in range(25):
            f()
        torch.cuda.synchronize()
        lib.MhExactBlurRecomputed(1)
        f()
        recomputed = lib.MhExactBlurRecomputed(0)
        sec = timed(torch, f, reps)
        prof = kernel_profile(im, f, 5)
    finally:
        for k in opts:
            im.set_option(k, None)
        im.set_precision(im.PRECISION_EXACT)
    print("%-34s %.4f ms  %8.1f Mpixel/s  recomputed %d  kernels %s" % (
        label, sec * 1e3, n * n / sec / 1e6, recomputed, {k: round(v["avg_ms"], 4) for k, v in prof.items()}), flush=True)
    return out.pixels.clone()


def compare(label, got, want):
    d = (got.view(torch.int16).to(torch.int32) & 0xffff) - (want.view(torch.int16).to(torch.int32) & 0xffff)
    d = d.abs()
    print("    %-30s max |diff| %d, identical %.4f %%, beyond 1: %d" % (
        label, int(d.max()), 100.0 * float((d == 0).float().mean()), int((d > 1).sum())), flush=True)


for kind in ("random", "opaque", "tiny"):
    image = im.Image(frame(kind))
    print("--- %s alpha, %d^2 RGBA, sigma %g" % (kind, n, sigma), flush=True)
    exact = run("EXACT (i8 both passes)", image, im.PRECISION_EXACT, {})
    hybrid = run("FAST hybrid", image, im.PRECISION_FAST, {})
    r3 = run("FAST r3 (exact row + f16 column)", image, im.PRECISION_FAST, {"MAGICKHIP_NO_HYBRID": "1"})
    compare("hybrid vs EXACT", hybrid, exact)
    compare("r3 FAST vs EXACT", r3, exact)
    del exact, hybrid, r3
for channels, alpha in ((4, False), (3, False)):
    image = im.Image(frame("random", channels), has_alpha=alpha)
    print("--- plain, %d channels" % channels, flush=True)
    opts = {"MAGICKHIP_FUSED_RGB": "1"} if channels == 3 else {}
    exact = run("EXACT", image, im.PRECISION_EXACT, opts)
    hybrid = run("FAST hybrid", image, im.PRECISION_FAST, opts)
    compare("hybrid vs EXACT", hybrid, exact)
    del exact, hybrid
