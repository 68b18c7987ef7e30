"""BlurImage(0,sigma) on n^2 RGBA Q16 in the mode the environment selects; one line per call.
    python tools/time_blur_modes.py exact|fast|hdri [n] [sigma] [channels]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
from imagemagick_amd import _lib
from bench import kernel_profile, timed
lib = im.load()
mode = sys.argv[1] if len(sys.argv) > 1 else "exact"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
channels = int(sys.argv[4]) if len(sys.argv) > 4 else 4
im.set_precision(im.PRECISION_FAST if mode == "fast" else im.PRECISION_EXACT)
gen = torch.Generator(device="cuda").manual_seed(3)
a = torch.randint(-32768, 32768, (n, n, channels), generator=gen, device="cuda", dtype=torch.int16).view(torch.uint16)
if mode == "hdri":          # float Quantum: the same levels as floats (EXACT)
    a = torch.where(a.view(torch.int16) < 0, a.view(torch.int16).to(torch.float32) + 65536.0, a.view(torch.int16).to(torch.float32))
if os.environ.get("OPAQUE") and channels == 4:
    a = a.clone()
    a.view(torch.int16)[:, :, 3] = -1
img = im.Image(a)
out = img.like()
def f():
    im.blur_image(img, 0.0, sigma, out=out)
# clock ramp
for _ in range(30):
    f()
torch.cuda.synchronize()
lib.MhExactBlurRecomputed(1)
f()
recomputed = lib.MhExactBlurRecomputed(0)
sec = timed(torch, f, 50 if mode != "hdri" else 5)
prof = kernel_profile(im, f, 5)
print("%-5s n=%d sigma=%g ch=%d: %.4f ms  %.1f Mpixel/s  recomputed %d of %d samples  kernels(ms) %s" % (
    mode, n, sigma, channels,
    sec * 1e3, n * n / sec / 1e6, recomputed, n * n * channels * 2,
    {k: round(v["avg_ms"], 4) for k, v in prof.items()}), flush=True)
