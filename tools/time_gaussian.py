"""GaussianBlurImage (2-D kernel) at 8192^2 RGBA Q16: FAST separated vs the generic 2-D kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
a = torch.randint(-32768, 32768, (n, n, 4), device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(a)
im.set_precision(im.PRECISION_FAST)
hold = {}
def f():
    hold["o"] = im.gaussian_blur_image(img, 0.0, sigma)
f(); f()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(3):
    f()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
print(os.environ.get("MAGICKHIP_NO_SEPARABLE"), "sigma %g: %.2f ms  %.1f Gpix/s" % (sigma, dt * 1e3, n * n / dt / 1e9),
      {k: round(v["avg_ms"], 3) for k, v in bench.kernel_profile(im, f, 2).items()})
