"""The headline call and nothing else, for profilers: N FAST (or EXACT) BlurImage(0x10) calls on one 8192^2 RGBA Q16 frame.
    python tools/blur_only.py [calls] [fast|exact] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import imagemagick_amd as im

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mode = sys.argv[2] if len(sys.argv) > 2 else "fast"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
im.load()
im.set_precision(im.PRECISION_FAST if mode == "fast" else im.PRECISION_EXACT)
g = torch.Generator(device="cuda").manual_seed(3)
a = torch.randint(-32768, 32768, (n, n, 4), generator=g, device="cuda", dtype=torch.int16).view(torch.uint16)
image = im.Image(a)
out = image.like()
for _ in range(calls):
    im.blur_image(image, 0.0, 10.0, out=out)
torch.cuda.synchronize()
print("done", calls, mode)
