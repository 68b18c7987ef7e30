"""C5's MAC-bound variant: ConvolveMorphology Disk:15 (normalised, `convolve:scale='!'`) on
16384^2 RGBA Q16, FAST: the matrix-core kernel against the generic 2-D kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
im.set_precision(im.PRECISION_FAST)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
kernels = sys.argv[2].split(",") if len(sys.argv) > 2 else ["Disk:15"]
gen = torch.Generator(device="cuda").manual_seed(3)
a = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(a)
hold = {}
for kernel in kernels:
    outs = []
    for label, env in (("matrix cores", None), ("generic", "1")):
        if env is None:
            os.environ.pop("MAGICKHIP_NO_MFMA_2D", None)
        else:
            os.environ["MAGICKHIP_NO_MFMA_2D"] = env
            if n > 8192 and os.environ.get("TIME_GENERIC") is None:
                continue

        def f():
            hold["o"] = im.morphology_image(img, "Convolve", 1, kernel, scale=(1.0, 1))
        sec = timed(torch, f, 2)
        prof = kernel_profile(im, f, 2)
        outs.append(hold["o"].pixels.clone())
        print("%-10s %-13s %.3f ms  %.1f Mpixels/s  kernels(ms) %s" % (kernel, label, sec * 1e3, n * n / sec / 1e6,
              {k: round(v["avg_ms"], 3) for k, v in prof.items()}), flush=True)
    if len(outs) == 2:
        d = (outs[0].view(torch.int16).to(torch.int32) & 0xffff) - (outs[1].view(torch.int16).to(torch.int32) & 0xffff)
        print("   max |difference| %d, identical %.4f" % (int(d.abs().max()), float((d == 0).float().mean())))
