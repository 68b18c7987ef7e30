"""C5's MAC-bound variant: ConvolveMorphology Disk:15 (normalised, `convolve:scale='!'`) on
16384^2 Q16: the exact-integer i8 matrix-core kernel (convolve2d_exact.hip, both modes), the f16
matrix-core kernel (convolve2d_mfma.hip, FAST) and — on frames up to 8192^2 or with TIME_GENERIC
set — the generic 2-D kernel.
  python tools/time_convolve2d.py [n] [kernel,kernel...] [rgba|plain4|rgb]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
kernels = sys.argv[2].split(",") if len(sys.argv) > 2 else ["Disk:15"]
layout = sys.argv[3] if len(sys.argv) > 3 else "rgba"
channels = 3 if layout == "rgb" else 4
gen = torch.Generator(device="cuda").manual_seed(3)
a = torch.randint(-32768, 32768, (n, n, channels), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(a, has_alpha=layout == "rgba")
hold = {}
lib = im._lib.load()
for kernel in kernels:
    outs = {}
    for label, env, mode in (("i8 exact", {}, im.PRECISION_EXACT),
                             ("f16 (FAST)", {"MAGICKHIP_NO_EXACT_2D": "1"}, im.PRECISION_FAST),
                             ("fused fp64", {"MAGICKHIP_NO_MFMA_2D": "1"}, im.PRECISION_EXACT),
                             ("generic", {"MAGICKHIP_NO_MFMA_2D": "1", "MAGICKHIP_NO_TIE_2D": "1"}, im.PRECISION_EXACT)):
        if label == "generic" and n > 8192 and os.environ.get("TIME_GENERIC") is None:
            continue
        for k in ("MAGICKHIP_NO_EXACT_2D", "MAGICKHIP_NO_MFMA_2D", "MAGICKHIP_NO_TIE_2D"):
            im.set_option(k, env.get(k))      # (the library reads the environment once, at start-up)
        im.set_precision(mode)

        def f():
            hold["o"] = im.morphology_image(img, "Convolve", 1, kernel, scale=(1.0, 1))
        lib.MhConvolve2DRecomputed(1)
        sec = timed(torch, f, 2)
        prof = kernel_profile(im, f, 2)
        recomputed = lib.MhConvolve2DRecomputed(0)
        outs[label] = hold["o"].pixels.clone()
        print("%-10s %-6s %-11s %.3f ms  %.1f Mpixels/s  kernels(ms) %s%s" % (
            kernel, layout, label, sec * 1e3, n * n / sec / 1e6, {k: round(v["avg_ms"], 3) for k, v in prof.items()},
            "  recomputed %d" % recomputed if label == "i8 exact" else ""), flush=True)
    im.set_precision(im.PRECISION_EXACT)
    base = outs["i8 exact"].view(torch.int16).to(torch.int32) & 0xffff
    for label, other in outs.items():
        if label != "i8 exact":
            d = base - (other.view(torch.int16).to(torch.int32) & 0xffff)
            print("   i8 exact against %-11s max |difference| %d, identical %.6f" % (label, int(d.abs().max()), float((d == 0).float().mean())))
