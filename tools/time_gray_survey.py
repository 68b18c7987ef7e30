import os, sys
sys.path.insert(0, "/root/repo")
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
n = 8192
gen = torch.Generator(device="cuda").manual_seed(3)
q4 = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
q1 = q4[:, :, :1].contiguous()
hold = {}
OPS = [
    ("dilate Disk:15", lambda i: im.morphology_image(i, "Dilate", 1, "Disk:15")),
    ("dilate Disk:5", lambda i: im.morphology_image(i, "Dilate", 1, "Disk:5")),
    ("erode Octagon:5", lambda i: im.morphology_image(i, "Erode", 1, "Octagon:5")),
    ("open Disk:5", lambda i: im.morphology_image(i, "Open", 1, "Disk:5")),
    ("dilate Square:1", lambda i: im.morphology_image(i, "Dilate", 1, "Square:1")),
    ("convolve Disk:5", lambda i: im.morphology_image(i, "Convolve", 1, "Disk:5", scale=(1.0, 1))),
    ("gaussian 0x3", lambda i: im.gaussian_blur_image(i, 0.0, 3.0)),
    ("resize x2", lambda i: im.resize_image(i, 2 * n, 2 * n, "Lanczos")),
    ("resize /2", lambda i: im.resize_image(i, n // 2, n // 2, "Lanczos")),
]
for prec_name, prec in (("fast", im.PRECISION_FAST), ("exact", im.PRECISION_EXACT)):
    im.set_precision(prec)
    print("== precision", prec_name, flush=True)
    for name, op in OPS:
        row = []
        for label, px in (("rgba", q4), ("gray", q1)):
            img = im.Image(px, has_alpha=(label == "rgba"))
            def f():
                hold["o"] = None
                hold["o"] = op(img)
            try:
                f()
                sec = timed(torch, f, 3)
                prof = kernel_profile(im, f, 1)
                row.append("%s %8.3f ms %s" % (label, sec * 1e3, {k: round(v["avg_ms"], 2) for k, v in prof.items()}))
            except Exception as exc:
                row.append("%s failed: %s" % (label, str(exc)[:60]))
            hold.clear()
        print("%-18s %s" % (name, "   |   ".join(row)), flush=True)
