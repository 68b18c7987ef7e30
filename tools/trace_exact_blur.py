"""Phase timing inside the exact fused blur kernel (diagnostic build -DMH_EXACT_TRACE of
convolve_fused_exact.hip): waves 0, 4, 8 and 12 of the first four workgroups stamp the shader clock
at the phase boundaries of 48 steady-state iterations.

  build:  make -C imagemagick_amd/csrc VARIANT=xtrace VDEFS=-DMH_EXACT_TRACE
  run:    MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip_xtrace.so python tools/trace_exact_blur.py [exact|fast] [sigma]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MAGICKHIP_EXACT_TRACE", "/tmp/exact_trace.bin")
import numpy as np
import torch
import imagemagick_amd as im

mode = sys.argv[1] if len(sys.argv) > 1 else "exact"
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
n = 8192
im.set_precision(im.PRECISION_EXACT if mode == "exact" else im.PRECISION_FAST)
im.set_option("MAGICKHIP_EXACT_TRACE", os.environ["MAGICKHIP_EXACT_TRACE"])     # (a selector switch: not read from the environment)
a = torch.randint(-32768, 32768, (n, n, 4), device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(a)
for _ in range(3):
    im.blur_image(img, 0.0, sigma)
torch.cuda.synchronize()
NAMES = ["top: store rows / original fetch -> wait for the staged fetch", "convert + stage (LDS writes done)", "issue next fetch",
         "interval A: chain + other pass's epilogue, LDS writes done", "barrier X",
         "interval B: chain + other pass's epilogue, LDS writes done", "barrier Y"]
t = np.fromfile(os.environ["MAGICKHIP_EXACT_TRACE"], dtype=np.uint64).reshape(4, 4, 48, 12).astype(np.int64)
for w in range(4):
    steps = t[:, w, 4:44, :8]
    whole = steps[:, 1:, 0] - steps[:, :-1, 0]
    print("wave %2d (%s): iteration period %.0f cycles (min %.0f, max %.0f)" %
          (4 * w, "stager" if 4 * w < 9 else "no staging", whole.mean(), whole.min(), whole.max()))
    marks = steps.copy()
    if 4 * w >= 9:                       # waves without staging never stamp marks 1, 2
        marks[:, :, 1] = marks[:, :, 0]
        marks[:, :, 2] = marks[:, :, 0]
    d = np.diff(marks, axis=2)
    for i, name in enumerate(NAMES):
        v = d[:, :, i]
        print("   %-66s mean %7.0f   p10 %7.0f   p90 %7.0f" % (name, v.mean(), np.percentile(v, 10), np.percentile(v, 90)))
top = t[:, :, 4:44, 0]
print("loop-top skew against wave 0 (cycles):", [(round(float((top[:, w] - top[:, 0]).mean()))) for w in range(4)])
