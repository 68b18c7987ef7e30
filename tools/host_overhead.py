#!/usr/bin/env python3
"""Host-side cost per operator call next to its end-to-end time, device-resident images:
for each workload prints the mean host time of a call that is NOT followed by a sync (what
the CPU spends building tables, launching, allocating) and the steady-state wall time per
call with one sync at the end.  wall >> kernel time with host ~= wall means the operator is
host-bound.   python tools/host_overhead.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagemagick_amd as im

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
im.load()
im.set_precision(im.PRECISION_FAST)
gen = torch.Generator(device="cuda").manual_seed(1)


def measure(name, fn):
    fn(); fn()
    torch.cuda.synchronize()
    host = 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        h0 = time.perf_counter()
        fn()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    print("%-34s host %8.3f ms/call   wall %8.3f ms/call" % (name, host / reps * 1e3, wall * 1e3), flush=True)


def u16(n):
    return torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)


hold = {}
img = im.Image(u16(8192))
measure("blur 0x10 8192^2 q16", lambda: hold.__setitem__("o", im.blur_image(img, 0.0, 10.0)))
measure("unsharp 0x10 8192^2 q16", lambda: hold.__setitem__("o", im.unsharp_mask_image(img, 0.0, 10.0, 1.0, 0.02)))
measure("dilate disk:15 8192^2 q16", lambda: hold.__setitem__("o", im.morphology_image(img, "Dilate", 1, "Disk:15")))
measure("resize 8192^2->4096^2 lanczos q16", lambda: hold.__setitem__("o", im.resize_image(img, 4096, 4096, "Lanczos")))
hold.clear()
img4 = im.Image(u16(4096))
work = {}


def c4():
    w = im.Image(img4.pixels.clone())
    im.transform_image_colorspace(w, "Lab")
    n = 4096 * 4096
    im.contrast_stretch_image(w, 0.02 * n, n - 0.01 * n)
    work["o"] = w


measure("c4 lab+cstretch 4096^2 q16", c4)


def eq():
    w = im.Image(img4.pixels.clone())
    im.equalize_image(w)
    work["o"] = w


measure("equalize 4096^2 q16", eq)
work.clear()
del img, img4
torch.cuda.empty_cache()
srcf = torch.rand((8192, 8192, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
imgf = im.Image(srcf)


def resize():
    hold["o"] = None
    hold["o"] = im.resize_image(imgf, 32768, 32768, "Lanczos")


measure("resize 8192^2->32768^2 lanczos f32", resize)
