"""Phase timing inside the matrix-core blur kernels (diagnostic build, see MH_MFMA_TRACE in
convolve_mfma.hip): wave 0 of eight workgroups stamps the shader clock at the phase boundaries
of its first 48 ring steps.

  build:  hipcc ... -DMH_MFMA_TRACE -c convolve_mfma.hip ; link as lib/libmagickhip_T.so
  run:    MAGICKHIP_LIBRARY=.../libmagickhip_T.so python tools/trace_blur_steps.py [sigma]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MAGICKHIP_MFMA_TRACE", "/tmp/mfma_trace")
import numpy as np
import torch
import imagemagick_amd as im

sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
n = 8192
im.set_precision(im.PRECISION_FAST)
a = torch.randint(-32768, 32768, (n, n, 4), device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(a)
for _ in range(3):
    im.blur_image(img, 0.0, sigma)
torch.cuda.synchronize()
NAMES = ["multiply+epilogue", "barrier B2", "wait for the fetch", "convert+stage (LDS done)",
         "issue next fetch", "stores", "barrier B1"]
for which in ("row", "column"):
    t = np.fromfile(os.environ["MAGICKHIP_MFMA_TRACE"] + "." + which, dtype=np.uint64).reshape(8, 48, 8).astype(np.int64)
    steps = t[:, 8:40, :]
    ok = (steps > 0).all(axis=2)
    d = np.diff(steps, axis=2)                                  # marks 0..7 -> 7 phases
    whole = steps[:, 1:, 0] - steps[:, :-1, 0]
    print("%s pass, sigma %g: step period %.0f cycles (min %.0f, max %.0f) over %d traced steps" %
          (which, sigma, whole.mean(), whole.min(), whole.max(), ok.sum()))
    for i, name in enumerate(NAMES):
        v = d[:, :, i][ok]
        print("   %-28s mean %7.0f   p10 %7.0f   p90 %7.0f" % (name, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
