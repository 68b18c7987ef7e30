"""C5 Dilate Disk:15 on 16384^2 RGBA Q16: the union-of-rectangles tile kernel against the
strip walk (MAGICKHIP_STRIPS=1) and the
plane-per-width kernel (MAGICKHIP_NO_RECTS=1); same bits, compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
kernels = sys.argv[2].split(",") if len(sys.argv) > 2 else ["Disk:15"]
gen = torch.Generator(device="cuda").manual_seed(3)
a = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(a)
hold = {}
for kernel in kernels:
    results = []
    for label, env in (("strips", {"MAGICKHIP_STRIPS": "1"}), ("tiles", {}), ("planes", {"MAGICKHIP_NO_RECTS": "1"})):
        for name in ("MAGICKHIP_NO_RECTS", "MAGICKHIP_STRIPS"):
            im.set_option(name, env.get(name))

        def f():
            hold["o"] = im.morphology_image(img, "Dilate", 1, kernel)
        sec = timed(torch, f, 3)
        prof = kernel_profile(im, f, 2)
        results.append(hold["o"].pixels.clone())
        print("%-10s %-7s %.3f ms  %.1f Mpixels/s  kernels(ms) %s" % (kernel, label, sec * 1e3, n * n / sec / 1e6,
              {k: round(v["avg_ms"], 3) for k, v in prof.items()}), flush=True)
    print("   identical:", bool(torch.equal(results[0], results[1])) and bool(torch.equal(results[0], results[2])))
