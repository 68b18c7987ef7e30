"""Host-side time of back-to-back ResizeImage calls (no synchronisation in between): does the host
run ahead of the GPU (a call returns in well under the 5.4 ms of its kernels) or does something in
the call block on the stream?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
im.set_precision(im.PRECISION_FAST)
m = 8192
srcf = torch.rand((m, m, 4), device="cuda", dtype=torch.float32) * 65535.0
img = im.Image(srcf)
hold = {}
for _ in range(2):
    hold["o"] = None
    hold["o"] = im.resize_image(img, 4 * m, 4 * m, "Lanczos")
torch.cuda.synchronize()
t = [time.perf_counter()]
for _ in range(6):
    hold["o"] = None
    t.append(time.perf_counter())
    hold["o"] = im.resize_image(img, 4 * m, 4 * m, "Lanczos")
    t.append(time.perf_counter())
torch.cuda.synchronize()
t.append(time.perf_counter())
d = [round((b - a) * 1e3, 3) for a, b in zip(t[:-1], t[1:])]
print("free / call pairs (ms):", d[:-1], "final sync", d[-1], "total", round((t[-1] - t[0]) * 1e3 / 6, 3), "ms per call")
