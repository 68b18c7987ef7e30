"""BlurImage on host (pixel-cache) buffers: upload + both passes + download, per transfer mode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import imagemagick_amd as im
im.set_precision(im.PRECISION_FAST)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(1)
host = rng.integers(0, 65536, (n, n, 4), dtype=np.uint16)
img = im.Image(host)
want = None
for _ in range(2):
    out = im.blur_image(img, 0.0, 10.0)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    out = im.blur_image(img, 0.0, 10.0)
dt = (time.perf_counter() - t0) / reps
dev = im.blur_image(im.Image(torch.from_numpy(host.view(np.int16)).cuda().view(torch.uint16)), 0.0, 10.0).numpy()
print(os.environ.get("MAGICKHIP_HOST_COPY"), os.environ.get("MAGICKHIP_TRANSFER_THREADS"),
      "%.1f ms  %.2f Gpix/s  %.1f GB/s moved  same=%s" % (dt * 1e3, n * n / dt / 1e9, 2 * host.nbytes / dt / 1e9,
                                                       bool(np.array_equal(out.numpy(), dev))))
