"""Phase sums inside blur_fused8_kernel (diagnostic build -DMH_FUSED_TRACE): every wave of the
first four workgroups adds up the shader-clock time of its phases over the whole walk.
  build:  make -C imagemagick_amd/csrc VARIANT=trace VDEFS=-DMH_FUSED_TRACE
  run:    MAGICKHIP_LIBRARY=$PWD/imagemagick_amd/lib/libmagickhip_trace.so python tools/trace_fused8.py [sigma]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MAGICKHIP_FUSED_TRACE", "/tmp/fused8_trace.bin")
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
t = np.fromfile(os.environ["MAGICKHIP_FUSED_TRACE"], dtype=np.uint64).reshape(4, 16, 5).astype(np.float64)
iters = 2 * (n // 2 // 16) + 12 + 2
t = t.mean(axis=0) / iters
print("cycles per iteration (8 rows), mean over 4 workgroups; total per wave:", np.round(t.sum(axis=1)))
print("row waves    : store+prefetch | reads+MFMA | epilogue+ring write | - | barrier")
for w in range(8):
    print("  wave %2d  " % w, np.round(t[w]))
print("column waves : wait fetch | convert+stage | fetch issue | column tile | barrier")
for w in range(8, 16):
    print("  wave %2d  " % w, np.round(t[w]))
