import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import imagemagick_amd as im
from oracle import ref
im.load()
rng = np.random.default_rng(5)
px = rng.integers(0, 65536, (240, 320, 4), dtype=np.uint16)
im.set_precision(im.PRECISION_FAST)
dev = im.Image(torch.from_numpy(px.view(np.int16)).cuda().view(torch.uint16))
for name in ("blur:0x10", "blur:0x10+90"):
    want = ref.RefImage(px).convolve(name).numpy().astype(np.int64)
    got = im.convolve_image(dev, name).numpy().astype(np.int64)
    d = got - want
    print(name, "max |d|", np.abs(d).max(), "hist", dict(zip(*[a.tolist() for a in np.unique(d, return_counts=True)])))
    ys, xs, cs = np.nonzero(np.abs(d) >= 2)
    for y, x, c in list(zip(ys.tolist(), xs.tolist(), cs.tolist()))[:12]:
        print("   at", y, x, c, "got", got[y, x, c], "want", want[y, x, c], "alpha in", px[y, x, 3])
