"""The path's operators on frames of one pixel count (4 Mpixel) and very different shapes — 2048^2, 64 rows, 64 columns, one
row, one column: ms per call, to find shapes a tile mapping handles badly.   python tools/time_aspect_survey.py [fast|exact]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
im.set_precision(im.PRECISION_FAST if mode == "fast" else im.PRECISION_EXACT)
gen = torch.Generator(device="cuda").manual_seed(3)
SHAPES = [(2048, 2048), (64, 65536), (65536, 64), (16, 262144), (262144, 16), (1, 4194304), (4194304, 1)]
OPS = [
    ("blur 0x10", lambda i, r, c: im.blur_image(i, 0.0, 10.0)),
    ("blur 0x2", lambda i, r, c: im.blur_image(i, 0.0, 2.0)),
    ("unsharp 0x3", lambda i, r, c: im.unsharp_mask_image(i, 0.0, 3.0, 1.0, 0.02)),
    ("gaussian 0x3", lambda i, r, c: im.gaussian_blur_image(i, 0.0, 3.0)),
    ("convolve Disk:5", lambda i, r, c: im.morphology_image(i, "Convolve", 1, "Disk:5", scale=(1.0, 1))),
    ("dilate Disk:15", lambda i, r, c: im.morphology_image(i, "Dilate", 1, "Disk:15")),
    ("erode Disk:3", lambda i, r, c: im.morphology_image(i, "Erode", 1, "Disk:3")),
    ("resize x2", lambda i, r, c: im.resize_image(i, 2 * c, 2 * r, "Lanczos")),
    ("resize /2", lambda i, r, c: im.resize_image(i, max(1, c // 2), max(1, r // 2), "Lanczos")),
    ("lab", None), ("equalize", None),
]
hold = {}
for qname in ("q16", "float"):
    print("== %s, %s" % (qname, mode), flush=True)
    for name, op in OPS:
        row = []
        for rows, cols in SHAPES:
            if qname == "q16":
                px = torch.randint(-32768, 32768, (rows, cols, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
            else:
                px = torch.rand((rows, cols, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
            try:
                if op is None:
                    def f():
                        img = im.Image(px.clone())
                        if name == "lab":
                            im.transform_image_colorspace(img, "Lab")
                        else:
                            im.equalize_image(img)
                else:
                    img = im.Image(px)
                    def f():
                        hold["o"] = None
                        hold["o"] = op(img, rows, cols)
                f()
                sec = timed(torch, f, 3)
                row.append("%dx%d %7.3f" % (rows, cols, sec * 1e3))
            except Exception as exc:
                row.append("%dx%d failed: %s" % (rows, cols, str(exc)[:50]))
            hold.clear()
        print("%-16s %s" % (name, " | ".join(row)), flush=True)
