"""Float Quantum (HDRI, the reference's default build) against Q16, operator by operator, on an
n x n RGBA frame: whole-call ms and the kernels that ran.   python tools/time_hdri_survey.py [n]
(The float frame holds fractional samples: Convolve with a flat kernel takes the fused fp64 kernel
there; tools/time_convolve2d_hdri.py times the integer-sample case.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
gen = torch.Generator(device="cuda").manual_seed(3)
q16 = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
flt = torch.rand((n, n, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
hold = {}
OPS = [
    ("blur 0x10", lambda i: im.blur_image(i, 0.0, 10.0)),
    ("blur 0x2", lambda i: im.blur_image(i, 0.0, 2.0)),
    ("gaussian_blur 0x3", lambda i: im.gaussian_blur_image(i, 0.0, 3.0)),
    ("unsharp 0x10", lambda i: im.unsharp_mask_image(i, 0.0, 10.0, 1.0, 0.02)),
    ("sharpen 0x2", lambda i: im.sharpen_image(i, 0.0, 2.0)),
    ("convolve Disk:5", lambda i: im.morphology_image(i, "Convolve", 1, "Disk:5", scale=(1.0, 1))),
    ("convolve Disk:15", lambda i: im.morphology_image(i, "Convolve", 1, "Disk:15", scale=(1.0, 1))),
    ("convolve LoG:0x2", lambda i: im.morphology_image(i, "Convolve", 1, "LoG:0x2")),
    # (a zero-sum kernel on an alpha-weighted frame is the reference's own walk by necessity, DESIGN.md 8;
    # without the alpha trait it is fused sums + tie check)
    ("convolve LoG:0x2 (rgb)", lambda i: im.morphology_image(i, "Convolve", 1, "LoG:0x2")),
    ("dilate Disk:15", lambda i: im.morphology_image(i, "Dilate", 1, "Disk:15")),
    ("erode Octagon:5", lambda i: im.morphology_image(i, "Erode", 1, "Octagon:5")),
    ("open Disk:5", lambda i: im.morphology_image(i, "Open", 1, "Disk:5")),
    ("edge Diamond:2", lambda i: im.morphology_image(i, "Edge", 1, "Diamond:2")),
    ("dilate Ring:2,4", lambda i: im.morphology_image(i, "Dilate", 1, "Ring:2,4")),
    ("resize x2 Lanczos", lambda i: im.resize_image(i, 2 * n, 2 * n, "Lanczos")),
    ("resize /2 Lanczos", lambda i: im.resize_image(i, n // 2, n // 2, "Lanczos")),
]
INPLACE = [
    ("colorspace Lab", lambda i: im.transform_image_colorspace(i, "Lab")),
    ("contrast_stretch", lambda i: im.contrast_stretch_image(i, 0.02 * n * n, n * n - 0.01 * n * n)),
    ("equalize", lambda i: im.equalize_image(i)),
    ("grayscale", lambda i: im.grayscale_image(i)),
]
for prec_name, prec in (("exact", im.PRECISION_EXACT), ("fast", im.PRECISION_FAST)):
    im.set_precision(prec)
    print("== precision", prec_name, flush=True)
    for name, op in OPS:
        row = []
        for label, px in (("q16", q16), ("hdri", flt)):
            img = im.Image(px[:, :, :3].contiguous()) if name.endswith("(rgb)") else im.Image(px)
            def f():
                hold["o"] = None
                hold["o"] = op(img)
            try:
                sec = timed(torch, f, 2)
                prof = kernel_profile(im, f, 1)
                row.append("%s %8.3f ms %s" % (label, sec * 1e3, {k: round(v["avg_ms"], 2) for k, v in prof.items()}))
            except Exception as exc:
                row.append("%s failed: %s" % (label, str(exc)[:60]))
            hold.clear()
        print("%-24s %s" % (name, "   |   ".join(row)), flush=True)
    for name, op in INPLACE:
        row = []
        for label, px in (("q16", q16), ("hdri", flt)):
            work = px.clone()
            def f():
                work.copy_(px)
                op(im.Image(work))
            try:
                sec = timed(torch, f, 2)
                prof = kernel_profile(im, f, 1)
                row.append("%s %8.3f ms %s" % (label, sec * 1e3, {k: round(v["avg_ms"], 2) for k, v in prof.items()}))
            except Exception as exc:
                row.append("%s failed: %s" % (label, str(exc)[:60]))
            del work
        print("%-24s %s" % (name, "   |   ".join(row)), flush=True)
