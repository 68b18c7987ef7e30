"""Phase timing inside the hybrid FAST blur kernel (diagnostic build -DMH_HYBRID_TRACE of
convolve_fused_hybrid.hip): all 16 waves of the first four workgroups stamp the shader clock at the
phase boundaries of 48 steady-state iterations.

  build:  make -C imagemagick_amd/csrc VARIANT=htrace VDEFS=-DMH_HYBRID_TRACE
  run:    MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip_htrace.so python tools/trace_hybrid_blur.py [rgba|plain]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MAGICKHIP_HYBRID_TRACE", "/tmp/hybrid_trace.bin")
import numpy as np
import torch
import imagemagick_amd as im

layout = sys.argv[1] if len(sys.argv) > 1 else "rgba"
n = 8192
im.load()
im.set_option("MAGICKHIP_HYBRID_TRACE", os.environ["MAGICKHIP_HYBRID_TRACE"])     # (a selector switch: not read from the environment)
im.set_precision(im.PRECISION_FAST)
a = torch.randint(-32768, 32768, (n, n, 4), device="cuda", dtype=torch.int16).view(torch.uint16)
img = im.Image(a, has_alpha=layout == "rgba")
for _ in range(3):
    im.blur_image(img, 0.0, 10.0)
torch.cuda.synchronize()
NAMES = ["stage (wait for the fetch, convert, LDS writes issued)", "issue next fetch", "alpha epilogue (alpha waves)",
         "column pass -> out_tile", "wait at barrier X", "B head: alpha reads + alpha chain issued", "row chain + colour epilogue + store",
         "wait at barrier Y"]
t = np.fromfile(os.environ["MAGICKHIP_HYBRID_TRACE"], dtype=np.uint64).reshape(4, 16, 48, 12).astype(np.int64)
for w in range(16):
    steps = t[:, w, 4:44, :9].copy()
    whole = steps[:, 1:, 0] - steps[:, :-1, 0]
    role = "stager" if w < 9 else ("tile + alpha" if w >= 12 and layout == "rgba" else "tile")
    steps[:, :, 1] = np.maximum(steps[:, :, 1], steps[:, :, 0])      # (waves that never stamp a mark)
    d = np.diff(steps, axis=2)
    print("wave %2d (%-12s) period %5.0f | %s" % (w, role, whole.mean(), "  ".join("%5.0f" % d[:, :, i].mean() for i in range(8))))
print("columns:", " | ".join("%d %s" % (i, name) for i, name in enumerate(NAMES)))
top = t[:, :, 4:44, 5]
print("leaving barrier X against wave 0 (cycles):", [round(float((top[:, w] - top[:, 0]).mean())) for w in range(16)])
