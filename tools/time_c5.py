#!/usr/bin/env python3
"""Times the C5 operators (16384^2 RGBA Q16 by default): Dilate/Erode Disk:15, Open Disk:15,
Dilate Square:7, with HIP events.   python tools/time_c5.py [edge]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import imagemagick_amd as im

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
im.load()
gen = torch.Generator(device="cuda").manual_seed(1)
src = torch.randint(-32768, 32768, (n, n, 4), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(src)
hold = {}


def timeit(name, fn, reps=4):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print("%-28s %8.3f ms  %8.1f Mpixel/s  %6.0f GB/s algorithmic" % (name, ms, n * n / ms / 1e3, 2.0 * n * n * 8 / ms / 1e6),
          flush=True)


for method, kernel in (("Dilate", "Disk:15"), ("Erode", "Disk:15"), ("Dilate", "Square:7"), ("Dilate", "Disk:5"),
                       ("Open", "Disk:15")):
    timeit("%s %s" % (method, kernel), lambda: hold.__setitem__("o", im.morphology_image(img, method, 1, kernel)))
