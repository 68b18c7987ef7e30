"""FAST BlurImage(0,10) on 8192^2 gray (1 channel) and gray + alpha (2 channels) Q16 frames."""
import torch, time, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagemagick_amd as im
def run(ch, alpha, precision, n=8192):
    im.set_precision(precision)
    a = torch.randint(-32768, 32768, (n, n, ch), device='cuda', dtype=torch.int16).view(torch.uint16)
    img = im.Image(a, has_alpha=alpha)
    for _ in range(3):
        im.blur_image(img, 0.0, 10.0)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10):
        im.blur_image(img, 0.0, 10.0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print("channels", ch, "alpha", alpha, "fast" if precision == im.PRECISION_FAST else "exact",
          "%.3f ms %.1f Gpix/s" % (dt * 1e3, n * n / dt / 1e9), flush=True)
for precision in (im.PRECISION_FAST, im.PRECISION_EXACT):
    run(1, False, precision); run(2, True, precision)
