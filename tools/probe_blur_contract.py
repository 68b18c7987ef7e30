#!/usr/bin/env python3
"""Which kernels does FAST BlurImage / UnsharpMaskImage launch for kernels with zero or tiny outer taps and for
kernels longer than the one-launch forms take, and how far from the reference do the results land on frames
that put BOTH passes on rounding ties?   python tools/probe_blur_contract.py   (GPU box; prints a table)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import imagemagick_amd as im
import bench
from oracle import ref as refmod

im.load()
refmod.set_thread_limit(os.cpu_count() or 1)


def frames(rows, cols):
    rng = np.random.default_rng(5)
    y, x = np.mgrid[0:rows, 0:cols]
    out = {}
    t = np.empty((rows, cols, 4), np.uint16)
    for c, level in enumerate((1000, 32767, 65533, 40000)):
        t[:, :, c] = level + (x & 1) + (y & 1)
    out["xy ties"] = t
    t2 = t.copy()
    t2[:, :, 3] = 65535
    out["xy ties opaque"] = t2
    checker = np.empty((rows, cols, 4), np.uint16)
    checker[:] = ((x + y) % 2 * 40000 + 100)[:, :, None]
    out["checker"] = checker
    sparse = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    sparse[:, :, 3] = rng.integers(0, 4, (rows, cols), dtype=np.uint16)
    out["tiny alpha"] = sparse
    binary = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    binary[:, :, 3] = np.where(rng.random((rows, cols)) < 0.5, 0, 65535)
    out["binary alpha"] = binary
    return out


def dev(px, **kw):
    return im.Image(torch.from_numpy(px.view(np.int16)).cuda().view(torch.uint16), **kw)


def run(radius, sigma, layout, unsharp):
    worst = 0
    names = set()
    for name, px in frames(150, 333).items():
        if layout == "rgb":
            px = np.ascontiguousarray(px[:, :, :3])
        has_alpha = layout == "rgba"
        if layout == "plain4":
            want = np.concatenate([(refmod.RefImage(px[:, :, c].copy()).unsharp(radius, sigma, 1.0, 0.02) if unsharp else
                                    refmod.RefImage(px[:, :, c].copy()).blur(radius, sigma)).numpy().reshape(px.shape[0], px.shape[1], 1)
                                   for c in range(4)], axis=2)
        else:
            r = refmod.RefImage(px)
            want = (r.unsharp(radius, sigma, 1.0, 0.02) if unsharp else r.blur(radius, sigma)).numpy()
        image = dev(px, has_alpha=has_alpha) if px.shape[2] == 4 else dev(px)
        holder = {}
        call = (lambda: holder.update(o=im.unsharp_mask_image(image, radius, sigma, 1.0, 0.02))) if unsharp else \
            (lambda: holder.update(o=im.blur_image(image, radius, sigma)))
        names |= set(bench.kernel_profile(im, call, 1))
        d = np.abs(holder["o"].numpy().astype(np.int64).reshape(want.shape) - want.astype(np.int64))
        worst = max(worst, int(d.max()))
        if d.max() > 1:
            print("    %-16s max %d, %d samples over 1" % (name, d.max(), int((d > 1).sum())))
    return worst, sorted(n.split("<")[0] for n in names)


for precision, label in ((im.PRECISION_FAST, "FAST"), (im.PRECISION_EXACT, "EXACT")):
    im.set_precision(precision)
    for unsharp in (False, True):
        for radius, sigma in ((30, 2), (40, 3), (25, 2), (12, 2), (0, 2), (0, 10), (0, 11), (0, 12.5), (0, 13.4), (50, 20)):
            for layout in ("rgba", "plain4", "rgb"):
                worst, names = run(float(radius), float(sigma), layout, unsharp)
                print("%-5s %-7s %gx%-5g %-6s max %d  %s" % (label, "unsharp" if unsharp else "blur", radius, sigma, layout,
                                                              worst, " ".join(names)), flush=True)
