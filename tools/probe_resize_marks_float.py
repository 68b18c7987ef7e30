import os, sys
sys.path.insert(0, os.getcwd())
import torch
import imagemagick_amd as im
im.load()
im.set_precision(im.PRECISION_FAST)
im.set_option("MAGICKHIP_RESIZE_STREAM_REPORT", "1")
im.set_option("MAGICKHIP_RESIZE_ONE_LAUNCH_MIN_PIXELS", "0")
m = 1024
gen = torch.Generator(device="cuda").manual_seed(1)
fl = torch.rand((m, m, 4), generator=gen, device="cuda", dtype=torch.float32) * 65535.0
for filt in ("Mitchell", "Catrom", "Lanczos", "Hermite", "Gaussian", "Spline", "Cubic", "Robidoux"):
    for f in (2, 3, 4):
        print(filt, "x%d float plain" % f, flush=True)
        im.resize_image(im.Image(fl, has_alpha=False), f * m, f * m, filt)
        torch.cuda.synchronize()
