"""EXACT BlurImage(0,10) on 8192^2 RGBA Q16: the fused-sum kernels with the tie check (Tie64)
against the separately rounded fp64 kernels (MAGICKHIP_NO_TIE64=1); same bits, compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
im.set_precision(im.PRECISION_EXACT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
gen = torch.Generator(device="cuda").manual_seed(3)
a = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(a)
hold = {}
results = {}
modes = (("Tie64 (fused sums + tie check)", None), ("Exact64 (separately rounded)", "1"))
if len(sys.argv) > 2 and sys.argv[2] == "tie":          # under rocprofv3: the default kernels only
    modes = modes[:1]
for label, env in modes:
    im.set_option("MAGICKHIP_NO_TIE64", env)      # (the library reads the environment once, at start-up)

    def f():
        hold["o"] = im.blur_image(img, 0.0, 10.0)
    sec = timed(torch, f, 5)
    prof = kernel_profile(im, f, 3)
    results[label] = hold["o"].pixels.clone()
    print("%-34s %.3f ms  %.1f Mpixels/s  kernels(ms) %s" % (label, sec * 1e3, n * n / sec / 1e6,
          {k: round(v["avg_ms"], 3) for k, v in prof.items()}), flush=True)
vals = list(results.values())
if len(vals) > 1:
    print("identical:", bool(torch.equal(vals[0], vals[1])))
