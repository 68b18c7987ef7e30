"""The certificate of the exact-integer 2-D convolve (convolve2d_exact.hip), checked on the CPU.

The kernel forms sum m*P exactly (integers) and takes the Quantum of unit*sum (or of the quotient of two
such sums) for the reference's — unless the value lies within a bound of a rounding boundary, in which
case it recomputes the sample in the reference's order.  The bound is host arithmetic
(launch_conv2d_exact: `relative`); this test restates the integer sums and the bound in NumPy from the
product's own host functions (MhKernelIntegerCells, the kernel builder) and asserts against the
compiled reference that EVERY sample whose level differs from the model's lies inside the bound — i.e.
that the device kernel would have recomputed it — and that such samples are rare."""
import numpy as np
import pytest

ULP = 1.1102230246251565e-16


def integer_sums(planes, mref, shiftx, shifty):
    """sum over the reflected window of mref[v][u] * plane[y-shifty+v][x-shiftx+u], edge-clamped."""
    kh, kw = mref.shape
    out = []
    for p in planes:
        padded = np.pad(p, ((shifty, kh - 1 - shifty), (shiftx, kw - 1 - shiftx)), mode="edge")
        rows, cols = p.shape
        s = np.zeros((rows, cols), dtype=np.int64)
        for v in range(kh):
            for u in range(kw):
                if mref[v, u] != 0:
                    s += int(mref[v, u]) * padded[v:v + rows, u:u + cols]
        out.append(s)
    return out


@pytest.mark.parametrize("alpha", [True, False])
@pytest.mark.parametrize("kernel", ["Disk:4.3", "Octagon:3", "Rectangle:6x4", "Plus:3",
                                    "5x5+1+3: 1,2,nan,2,1 2,4,6,4,2 3,6,9,6,3 nan,4,6,4,2 1,2,3,2,1"])
def test_integer_sums_decide_the_level_outside_the_bound(im, refmod, kernel, alpha):
    rng = np.random.default_rng(len(kernel))
    rows, cols = 41, 57
    px = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    px[5:15, 10:30, 3] = rng.integers(0, 4, (10, 20))          # tiny alpha
    px[20:26, 35:50, 3] = 0                                      # transparent
    px[30:40, 5:25, 3] = 65535
    values, kx, ky, _ = im.kernel_to_numpy(kernel, scale=(1.0, 1))
    got = im.kernel_integer_cells(kernel, scale=(1.0, 1))
    assert got is not None
    m, unit = got
    kh, kw = m.shape
    mref = m[::-1, ::-1]                                         # the reflected walk, morphology.c:2925
    shiftx, shifty = kw - 1 - kx, kh - 1 - ky
    # the host's bound (launch_conv2d_exact)
    cells = int((~np.isnan(values)).sum())
    nonzero = ~np.isnan(values) & (values != 0.0)
    worst = float(np.max(np.abs(values[nonzero] - m[nonzero] * unit) / np.abs(values[nonzero]))) + 4.440892098500626e-16
    relative = 2.0 * (worst + ULP * (cells + 12.0))
    p = px.astype(np.int64)
    if alpha:
        want = refmod.RefImage(px).set_artifact("convolve:scale", "!").morphology("Convolve", 1, kernel).numpy()
        planes = [p[:, :, 3] * p[:, :, c] for c in range(3)] + [p[:, :, 3]]
    else:
        want = np.stack([refmod.RefImage(px[:, :, c].copy()).set_artifact("convolve:scale", "!")
                         .morphology("Convolve", 1, kernel).numpy().reshape(rows, cols) for c in range(4)], axis=2)
        planes = [p[:, :, c] for c in range(4)]
    sums = integer_sums(planes, mref, shiftx, shifty)
    undecided = total = 0
    for c in range(4):
        weighted = alpha and c != 3
        if weighted:
            d = sums[3].astype(np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                value = np.where(d > 0, sums[c].astype(np.float64) / d, 0.0)
            bound = value * 2.0 * relative + 1.0e-9
        else:
            value = unit * sums[c].astype(np.float64)
            bound = value * relative + 1.0e-9
        shifted = value + 0.5
        level = np.minimum(np.floor(np.maximum(shifted, 0.0)), 65535.0).astype(np.int64)
        fraction = shifted - np.floor(shifted)
        distance = np.minimum(fraction, 1.0 - fraction)
        doubtful = (value < 65536.0) & ~(distance > bound)
        differs = level != want[:, :, c].astype(np.int64)
        assert not (differs & ~doubtful).any(), "%s channel %d: %d samples differ outside the bound" % (
            kernel, c, int((differs & ~doubtful).sum()))
        undecided += int(doubtful.sum())
        total += doubtful.size
    # an even cell sum puts one sample in sum(m) on a true tie; otherwise the bound is a few 1e-9 level wide
    even = int(m.sum()) % 2 == 0
    assert undecided <= (0.2 if even else 0.002) * total, (undecided, total)
