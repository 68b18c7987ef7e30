"""The certificate of the separable EXACT 2-D convolve (convolve_separable.hip), checked on the CPU.

GaussianBlurImage's kernel is an outer product column x row to within rounding; the library runs it as
two fp64 passes over alpha-premultiplied doubles and takes the Quantum of the result for the
reference's unless it lies within error_unit * max|P| of a rounding boundary (launch_separable_exact).
This test restates the two passes in extended precision and that bound in NumPy, from the product's
own factorisation (MhKernelOuterProductFactors), and asserts against the compiled reference that every
Q16 sample whose level differs from the model's lies inside the bound."""
import numpy as np
import pytest

ULP = 1.1102230246251565e-16


@pytest.mark.parametrize("alpha", [True, False])
@pytest.mark.parametrize("kernel", ["Gaussian:0x2", "Gaussian:4x1.3", "Square:2",
                                    "5x3+1+2: 0.01,0.02,0.03,0.02,0.01 0.02,0.04,0.06,0.04,0.02 0.03,0.06,0.09,0.06,0.03"])
def test_two_pass_sums_decide_the_level_outside_the_bound(im, refmod, kernel, alpha):
    rng = np.random.default_rng(len(kernel) + 2)
    rows, cols = 45, 61
    px = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    px[5:15, 10:30, 3] = rng.integers(0, 4, (10, 20))
    px[20:26, 35:50, 3] = 0
    values, kx, ky, _ = im.kernel_to_numpy(kernel)
    kh, kw = values.shape
    factors = im.kernel_outer_product_factors(kernel)
    assert factors is not None
    row, column = factors
    shiftx, shifty = kw - 1 - kx, kh - 1 - ky
    # the host's bound, launch_separable_exact: the reference's walk from the last cell backwards
    residual = magnitude = running = partials = 0.0
    for i in range(kw * kh - 1, -1, -1):
        cell = values.ravel()[i]
        product = column[i // kw] * row[i % kw]
        residual += abs(cell - product) + abs(product) * 2.220446049250313e-16
        magnitude += abs(cell)
        running += abs(cell)
        partials += running
    error_unit = 2.0 * (residual + ULP * (partials + 2.0 * magnitude) + ULP * ((kw + kh) + 8.0) * magnitude)
    p = px.astype(np.longdouble)
    if alpha:
        want = refmod.RefImage(px).morphology("Convolve", 1, kernel).numpy()
        planes = [p[:, :, 3] * p[:, :, c] for c in range(3)] + [p[:, :, 3]]
        fixed = [65535.0 * 65535.0] * 3 + [65535.0]
    else:
        want = np.stack([refmod.RefImage(px[:, :, c].copy()).morphology("Convolve", 1, kernel).numpy()
                         .reshape(rows, cols) for c in range(4)], axis=2)
        planes = [p[:, :, c] for c in range(4)]
        fixed = [65535.0] * 4
    rrow, rcol = row[::-1], column[::-1]                         # the reflected walk
    sums = []
    for q in planes:
        padded = np.pad(q, ((shifty, kh - 1 - shifty), (shiftx, kw - 1 - shiftx)), mode="edge")
        h = np.zeros((rows + kh - 1, cols), dtype=np.longdouble)
        for u in range(kw):
            h += np.longdouble(rrow[u]) * padded[:, u:u + cols]
        s = np.zeros((rows, cols), dtype=np.longdouble)
        for v in range(kh):
            s += np.longdouble(rcol[v]) * h[v:v + rows]
        sums.append(s)
    undecided = total = 0
    for c in range(4):
        weighted = alpha and c != 3
        error = error_unit * fixed[c]
        if weighted:
            sa = sums[3].astype(np.float64)
            ea = error_unit * fixed[3]
            with np.errstate(divide="ignore", invalid="ignore"):
                inverse = np.where(sa != 0.0, 1.0 / sa, 0.0)
            value = sums[c].astype(np.float64) * inverse
            bound = (np.abs(value) * ea + error) * np.abs(inverse) + np.abs(value) * 1.0e-15
            unsure = (sa != 0.0) & (~(np.abs(sa / 65535.0) >= 1.000001e-12) | ~(np.abs(sa) > 8.0 * ea))
        else:
            value = sums[c].astype(np.float64)
            bound = np.full((rows, cols), error)
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
    assert undecided <= 0.12 * total, (undecided, total)
