#!/usr/bin/env python3
"""Find FAST ResizeImage results more than one level from the reference's and show what is under them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import imagemagick_amd as im
from oracle import ref as refmod
im.load(); im.set_precision(im.PRECISION_EXACT)
rng = np.random.default_rng(5)

def dev(px, **kw):
    return im.Image(torch.from_numpy(px.view(np.int16)).cuda().view(torch.uint16), **kw)

def run(px, target, filt, fast):
    im.set_precision(im.PRECISION_FAST if fast else im.PRECISION_EXACT)
    try:
        return im.resize_image(dev(px), target[0], target[1], filt).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)

found = 0
for trial in range(6000):
    mode = 3 if os.environ.get('ONLY_BINARY') else trial % 3
    rows, cols = int(rng.integers(20, 150)), int(rng.integers(20, 200))
    px = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    if mode == 0:
        px[:, :, 3] = rng.integers(0, 4, (rows, cols)); filt = "Triangle"
        target = (int(rng.integers(2, cols)), int(rng.integers(2, rows)))
    elif mode == 1:
        px[:, :, 3] = np.where(rng.random((rows, cols)) < 0.5, 0, 65535); filt = "Catrom"
        f = int(rng.integers(2, 5)); target = (f * cols, int(rng.integers(f * rows, 5 * rows)))
    elif mode == 2:
        px[:, :, 3] = rng.integers(0, 4, (rows, cols)); filt = "Catrom"
        f = int(rng.integers(2, 5)); target = (f * cols, int(rng.integers(f * rows, 5 * rows)))
    else:
        px[:, :, 3] = np.where(rng.random((rows, cols)) < 0.5, 0, 65535); filt = "Lanczos"
        f = int(rng.integers(2, 5)); target = (f * cols, int(rng.integers(f * rows, 5 * rows)))
    want = refmod.RefImage(px).resize(target[0], target[1], filt).numpy()
    got = run(px, target, filt, True)
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    if d.max() <= 1:
        continue
    found += 1
    exact = run(px, target, filt, False)
    y, x, c = [int(v) for v in np.argwhere(d > 1)[0]]
    print("== %dx%d -> %dx%d %s mode %d: %d samples off, first (y %d, x %d, c %d): got %s want %s exact-mode %s" % (
        cols, rows, target[0], target[1], filt, mode, int((d > 1).sum()), y, x, c, got[y, x], want[y, x], exact[y, x]))
    print("   exact mode identical to the reference:", np.array_equal(exact, want))
    # the reference's intermediate: vertical first when x_factor <= y_factor ... (resize.c:3846-3861)
    xf, yf = target[0] / cols, target[1] / rows
    if xf > yf:
        mid_want = refmod.RefImage(px).resize(target[0], rows, filt).numpy()
        mid_got = run(px, (target[0], rows), filt, True)
        print("   horizontal first; intermediate FAST vs reference: max", np.abs(mid_got.astype(np.int64) - mid_want).max(),
              "differing samples", int((mid_got != mid_want).sum()))
        col = mid_want[:, x, :]
        print("   intermediate column x=%d alpha:" % x, col[max(0, int(y / yf) - 4): int(y / yf) + 6, 3].tolist())
        bad = np.argwhere(mid_got != mid_want)
        for b in bad[:6]:
            print("     intermediate differs at", b.tolist(), "got", mid_got[tuple(b[:2])], "want", mid_want[tuple(b[:2])])
    else:
        mid_want = refmod.RefImage(px).resize(cols, target[1], filt).numpy()
        mid_got = run(px, (cols, target[1]), filt, True)
        print("   vertical first; intermediate FAST vs reference: max", np.abs(mid_got.astype(np.int64) - mid_want).max(),
              "differing samples", int((mid_got != mid_want).sum()))
        row = mid_want[y, :, :]
        lo_c = max(0, int(x / xf) - 4)
        print("   intermediate row y=%d, columns %d..: alpha" % (y, lo_c), row[lo_c: int(x / xf) + 6, 3].tolist(),
              "channel %d" % c, row[lo_c: int(x / xf) + 6, c].tolist())
        print("   FAST intermediate there: alpha", mid_got[y, lo_c: int(x / xf) + 6, 3].tolist(), "channel", mid_got[y, lo_c: int(x / xf) + 6, c].tolist())
        bad = np.argwhere(mid_got != mid_want)
        for b in bad[:6]:
            print("     intermediate differs at", b.tolist(), "got", mid_got[tuple(b[:2])], "want", mid_want[tuple(b[:2])])
    # second pass alone on the reference's intermediate
    second = run(mid_want, target, filt, True)
    print("   second pass alone (FAST) on the reference's intermediate: max diff",
          np.abs(second.astype(np.int64) - want).max())
    if found >= 8:
        break
print("found", found)
