"""Every operator of the path on every pixel layout, Q16 and float: ms per call on an n x n frame and the kernels that ran —
to find layouts that are slower than a WIDER one (how round 6 found gray and RGB frames on the slow forms of Erode /
Dilate).   python tools/time_layout_survey.py [n] [fast|exact]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from bench import kernel_profile, timed
im.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
mode = sys.argv[2] if len(sys.argv) > 2 else "fast"
im.set_precision(im.PRECISION_FAST if mode == "fast" else im.PRECISION_EXACT)
gen = torch.Generator(device="cuda").manual_seed(3)
q4 = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
f4 = torch.rand((n, n, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
LAYOUTS = [("gray", 1, False), ("gray+a", 2, True), ("rgb", 3, False), ("rgba", 4, True), ("plain4", 4, False)]
OPS = [
    ("blur 0x10", lambda i: im.blur_image(i, 0.0, 10.0)),
    ("blur 0x2", lambda i: im.blur_image(i, 0.0, 2.0)),
    ("blur 0x14", lambda i: im.blur_image(i, 0.0, 14.0)),
    ("unsharp 0x3", lambda i: im.unsharp_mask_image(i, 0.0, 3.0, 1.0, 0.02)),
    ("gaussian 0x3", lambda i: im.gaussian_blur_image(i, 0.0, 3.0)),
    ("sharpen 0x2", lambda i: im.sharpen_image(i, 0.0, 2.0)),
    ("convolve Disk:5", lambda i: im.morphology_image(i, "Convolve", 1, "Disk:5", scale=(1.0, 1))),
    ("convolve 3x3", lambda i: im.morphology_image(i, "Convolve", 1, "3x3: 1,2,1 2,4,2 1,2,1", scale=(1.0, 1))),
    ("dilate Disk:15", lambda i: im.morphology_image(i, "Dilate", 1, "Disk:15")),
    ("erode Disk:3", lambda i: im.morphology_image(i, "Erode", 1, "Disk:3")),
    ("dilate Ring:2,4", lambda i: im.morphology_image(i, "Dilate", 1, "Ring:2,4")),
    ("resize x2", lambda i: im.resize_image(i, 2 * n, 2 * n, "Lanczos")),
    ("resize x1.5", lambda i: im.resize_image(i, 3 * n // 2, 3 * n // 2, "Lanczos")),
    ("resize /2", lambda i: im.resize_image(i, n // 2, n // 2, "Lanczos")),
    ("resize /3 Mitchell", lambda i: im.resize_image(i, n // 3, n // 3, "Mitchell")),
]
INPLACE = [
    ("colorspace Lab", lambda i: im.transform_image_colorspace(i, "Lab")),
    ("colorspace linear", lambda i: im.transform_image_colorspace(i, "RGB")),
    ("contrast_stretch", lambda i: im.contrast_stretch_image(i, 0.02 * n * n, n * n - 0.01 * n * n)),
    ("equalize", lambda i: im.equalize_image(i)),
    ("grayscale", lambda i: im.grayscale_image(i)),
]
hold = {}
for qname, base in (("q16", q4), ("float", f4)):
    print("== %s, %s, %d^2" % (qname, mode, n), flush=True)
    for name, op in OPS:
        row = []
        for label, channels, alpha in LAYOUTS:
            px = base[:, :, 4 - channels:].contiguous() if alpha and channels < 4 else base[:, :, :channels].contiguous()
            img = im.Image(px, has_alpha=alpha)
            def f():
                hold["o"] = None
                hold["o"] = op(img)
            try:
                f()
                sec = timed(torch, f, 3)
                prof = kernel_profile(im, f, 1)
                row.append("%s %7.3f %s" % (label, sec * 1e3, "+".join(sorted(prof))))
            except Exception as exc:
                row.append("%s failed: %s" % (label, str(exc)[:40]))
            hold.clear()
            del img, px
        print("%-18s %s" % (name, " | ".join(row)), flush=True)
    for name, op in INPLACE:
        row = []
        for label, channels, alpha in LAYOUTS:
            if channels < 3 and name.startswith("colorspace"):
                continue
            px = base[:, :, 4 - channels:].contiguous() if alpha and channels < 4 else base[:, :, :channels].contiguous()
            try:
                def f():
                    img = im.Image(px.clone(), has_alpha=alpha)
                    op(img)
                f()
                sec = timed(torch, f, 3)
                prof = kernel_profile(im, f, 1)
                row.append("%s %7.3f %s" % (label, sec * 1e3, "+".join(sorted(prof))))
            except Exception as exc:
                row.append("%s failed: %s" % (label, str(exc)[:40]))
            del px
        print("%-18s %s" % (name, " | ".join(row)), flush=True)
