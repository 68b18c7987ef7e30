"""The certificate of the fused fp64 2-D convolve (convolve2d_tie.hip), checked on the CPU.

The kernel forms S = sum k*P with one fused multiply-add per cell and takes the Quantum of S_c/S_alpha
(or S_c) for the reference's unless the value lies within error_unit * max|P| (of the 64 x 16 tile's
window) of a rounding boundary.  This test restates S in extended precision and the host's bound
(launch_conv2d_tie) in NumPy and asserts against the compiled reference that every Q16 sample whose
level differs from the model's lies inside the bound — the device kernel would have recomputed it —
and that such samples are rare."""
import numpy as np
import pytest

ULP = 1.1102230246251565e-16


@pytest.mark.parametrize("alpha", [True, False])
@pytest.mark.parametrize("kernel", ["Disk:4.3", "Gaussian:3x1.7",
                                    "7x5+2+1: 0.11,0.52,0.73,0.14,0.95,0.36,0.27 0.2,nan,0.6,0.8,0.6,nan,0.2 "
                                    "0.31,0.62,0.93,1.3,0.9,0.6,0.2 0.2,0.4,0.6,0.8,0.6,0.4,0.2 0.1,0.2,0.3,0.4,0.3,0.2,0.1"])
def test_fused_sums_decide_the_level_outside_the_bound(im, refmod, kernel, alpha):
    rng = np.random.default_rng(len(kernel) + 1)
    rows, cols = 37, 150
    px = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    px[5:15, 10:30, 3] = rng.integers(0, 4, (10, 20))          # tiny alpha
    px[20:26, 35:50, 3] = 0                                      # transparent
    px[25:36, 80:140, 3] = 65535
    values, kx, ky, _ = im.kernel_to_numpy(kernel, scale=(1.0, 1))
    kh, kw = values.shape
    window = np.where(np.isnan(values), 0.0, values)[::-1, ::-1]          # the reflected walk, NaN = no cell
    shiftx, shifty = kw - 1 - kx, kh - 1 - ky
    # the host's bound: the reference's walk from the last cell backwards
    walk = values.ravel()[::-1]
    walk = walk[~np.isnan(walk)]
    magnitude = float(np.abs(walk).sum())
    partials = float(np.cumsum(np.abs(walk)).sum())
    error_unit = 2.0 * ULP * (partials + 4.0 * magnitude + (walk.size + 8.0) * magnitude)
    p = px.astype(np.longdouble)
    if alpha:
        want = refmod.RefImage(px).set_artifact("convolve:scale", "!").morphology("Convolve", 1, kernel).numpy()
        planes = [p[:, :, 3] * p[:, :, c] for c in range(3)] + [p[:, :, 3]]
    else:
        want = np.stack([refmod.RefImage(px[:, :, c].copy()).set_artifact("convolve:scale", "!")
                         .morphology("Convolve", 1, kernel).numpy().reshape(rows, cols) for c in range(4)], axis=2)
        planes = [p[:, :, c] for c in range(4)]
    pad = ((shifty, kh - 1 - shifty), (shiftx, kw - 1 - shiftx))
    padded = [np.pad(q, pad, mode="edge") for q in planes]
    sums = []
    for q in padded:
        s = np.zeros((rows, cols), dtype=np.longdouble)
        for v in range(kh):
            for u in range(kw):
                if window[v, u] != 0.0:
                    s += np.longdouble(window[v, u]) * q[v:v + rows, u:u + cols]
        sums.append(s)
    # the largest |P| of each 64 x 16 tile's window, as the kernel's staging finds it
    most = []
    for q in padded:
        m = np.zeros((rows, cols))
        for y0 in range(0, rows, 16):
            for x0 in range(0, cols, 64):
                region = q[y0:y0 + 16 + kh - 1, x0:x0 + 64 + kw - 1]
                m[y0:y0 + 16, x0:x0 + 64] = float(np.abs(region).max())
        most.append(m)
    undecided = total = 0
    for c in range(4):
        weighted = alpha and c != 3
        error = error_unit * most[c]
        if weighted:
            sa = sums[3].astype(np.float64)
            ea = error_unit * most[3]
            with np.errstate(divide="ignore", invalid="ignore"):
                inverse = np.where(sa != 0.0, 1.0 / sa, 0.0)
            value = sums[c].astype(np.float64) * inverse
            bound = (np.abs(value) * ea + error) * np.abs(inverse) + np.abs(value) * 1.0e-15
            # tie_check.hpp: an alpha sum below PerceptibleReciprocal's clamp or down at its own error
            unsure = (sa != 0.0) & (~(np.abs(sa / 65535.0) >= 1.000001e-12) | ~(np.abs(sa) > 8.0 * ea))
        else:
            value = sums[c].astype(np.float64)
            bound = error
            unsure = np.zeros((rows, cols), dtype=bool)
        shifted = value + 0.5
        level = np.minimum(np.floor(np.maximum(shifted, 0.0)), 65535.0).astype(np.int64)
        fraction = shifted - np.floor(shifted)
        distance = np.minimum(fraction, 1.0 - fraction)
        doubtful = ((value > -1.0) & (value < 65536.0) & ~(distance > bound + 1.0e-9)) | unsure
        differs = level != want[:, :, c].astype(np.int64)
        assert not (differs & ~doubtful).any(), "%s channel %d: %d samples differ outside the bound" % (
            kernel[:20], c, int((differs & ~doubtful).sum()))
        undecided += int(doubtful.sum())
        total += doubtful.size
    # (the windows over the tiny-alpha block are undecided by design: their alpha sums are a few levels)
    assert undecided <= 0.08 * total, (undecided, total)
