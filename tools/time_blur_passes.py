"""Per-kernel average times of the headline blur (hipEvent profile of the library)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im
import bench
im.set_precision(im.PRECISION_FAST)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
cols = int(sys.argv[2]) if len(sys.argv) > 2 else rows
a = torch.randint(-32768, 32768, (rows, cols, 4), device="cuda", dtype=torch.int16).view(torch.uint16)
if os.environ.get("ZERO"):
    a.zero_()
img = im.Image(a)
hold = {}
def f():
    hold["o"] = im.blur_image(img, 0.0, 10.0)
for _ in range(3):
    f()
prof = bench.kernel_profile(im, f, 10)
print(rows, cols, os.environ.get("ZERO"), os.environ.get("MAGICKHIP_MFMA_PER_CU"),
      "Gpix/s per pass", {k: round(rows * cols / v["avg_ms"] / 1e6, 1) for k, v in prof.items()}, {k: round(v["avg_ms"], 4) for k, v in prof.items()})
