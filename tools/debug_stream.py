#!/usr/bin/env python3
"""Where the vector-pipe resize differs from the reference (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import imagemagick_amd as im
from oracle import ref
from conftest import make_pixels
im.load()
def run(shape, target, filt, dtype, alpha):
    px = make_pixels(shape[0], shape[1], 4, dtype, seed=shape[1] + target[0])
    if alpha:
        px[: shape[0] // 2, :, 3] = 65535
        px[:, : shape[1] // 5, 3] = 0
    t = torch.from_numpy(px.view(np.int16) if dtype == np.uint16 else px).to("cuda:0")
    if dtype == np.uint16: t = t.view(torch.uint16)
    dev = im.Image(t, has_alpha=alpha)
    want = ref.RefImage(px).resize(target[0], target[1], filt).numpy()
    im.set_precision(im.PRECISION_FAST)
    got = im.resize_image(dev, target[0], target[1], filt).numpy()
    d = np.abs(got.astype(np.float64) - want.astype(np.float64))
    bad = np.argwhere(d > 1)
    print(shape, target, filt, dtype.__name__, alpha, "bad", len(bad), "max", d.max())
    if len(bad):
        print(" rows", np.unique(bad[:, 0])[:40], "\n cols", np.unique(bad[:, 1])[:60], "\n ch", np.unique(bad[:, 2]))
        for b in bad[:8]:
            print("  ", b, got[tuple(b)], want[tuple(b)], "alpha got/want", got[b[0], b[1], 3], want[b[0], b[1], 3])
run((41, 50, 4), (150, 164), "Triangle", np.uint16, True)
run((41, 50, 4), (150, 164), "Triangle", np.float32, True)
run((41, 50, 4), (150, 164), "Triangle", np.uint16, False)
