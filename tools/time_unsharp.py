"""UnsharpMaskImage 0x10 on 16384^2 RGBA Q16 (BASELINE C5's second operator), FAST."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
a = torch.randint(-32768, 32768, (n, n, 4), device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(a)
im.set_precision(im.PRECISION_FAST)
hold = {}
def f():
    hold["o"] = None
    hold["o"] = im.unsharp_mask_image(img, 0.0, 10.0, 1.0, 0.05)
f(); f()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5):
    f()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(os.environ.get("MAGICKHIP_NO_FUSED_UNSHARP"), "%.2f ms  %.1f Gpix/s" % (dt * 1e3, n * n / dt / 1e9),
      {k: round(v["avg_ms"], 3) for k, v in bench.kernel_profile(im, f, 2).items()})
