"""BlurImage on host (pixel-cache) buffers — what one un-chained operator costs through the
boundary: the whole-frame path (upload, kernels, download in sequence) against the banded
pipeline (batch.cpp::host_banded_operator: uploads, kernels and downloads of different row bands
overlap), for a few worker / staging-thread counts.  Prints ms, Gpixel/s and GB/s moved."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import imagemagick_amd as im

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(1)
host = rng.integers(0, 65536, (n, n, 4), dtype=np.uint16)
img = im.Image(host)
device_image = im.Image(torch.from_numpy(host.view(np.int16)).cuda().view(torch.uint16))


def run(label, env, precision):
    saved = {k: os.environ.get(k) for k in env}
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    im.set_precision(precision)
    try:
        for _ in range(2):
            out = im.blur_image(img, 0.0, 10.0)
        reps = 4
        t0 = time.perf_counter()
        for _ in range(reps):
            out = im.blur_image(img, 0.0, 10.0)
        dt = (time.perf_counter() - t0) / reps
        want = im.blur_image(device_image, 0.0, 10.0).numpy().astype(np.int64)
        worst = int(np.abs(out.numpy().astype(np.int64) - want).max())
    finally:
        im.set_precision(im.PRECISION_EXACT)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    print("%-44s %6.1f ms  %5.2f Gpix/s  %5.1f GB/s moved  max |diff to device-resident| = %d" % (
        label, dt * 1e3, n * n / dt / 1e9, 2 * host.nbytes / dt / 1e9, worst), flush=True)


fast, exact = im.PRECISION_FAST, im.PRECISION_EXACT
run("whole frame, 4 staging threads (round 1)", {"MAGICKHIP_NO_BANDED": "1"}, fast)
run("whole frame, 8 staging threads", {"MAGICKHIP_NO_BANDED": "1", "MAGICKHIP_TRANSFER_THREADS": "8"}, fast)
for workers in (2, 4, 6, 8):
    for threads in (2, 4):
        run("banded, %d workers x %d staging threads" % (workers, threads),
            {"MAGICKHIP_NO_BANDED": None, "MAGICKHIP_BANDED_WORKERS": str(workers),
             "MAGICKHIP_TRANSFER_THREADS": str(threads)}, fast)
run("banded, default, EXACT", {"MAGICKHIP_NO_BANDED": None}, exact)
