"""The certificate of the float-Quantum (HDRI) 1-D convolve with the float-rounding tie check
(convolve.hip, Accum::finish, Tie64 on float), checked on the CPU.

The kernel forms S = sum k*P with fused multiply-adds over alpha-premultiplied doubles and takes the
float nearest to S_c/S_alpha (or S_c) for the reference's unless that value lies closer to the midpoint
of two floats than (2K+6)*2^-53 * max|P| of the window (plus the quotient's share).  This test
restates the sums in extended precision and that bound in NumPy and asserts against the compiled HDRI
reference that every sample whose float differs from the model's lies inside the bound."""
import numpy as np
import pytest

ULP = 1.1102230246251565e-16


def half_ulp_of(nearest):
    """Half an ulp of a float32 (a quarter just below a power of two), as Accum::finish derives it."""
    bits = nearest.view(np.uint32)
    exponent = ((bits >> 23) & 0xff).astype(np.int64)
    power_of_two = (bits & 0x7fffff) == 0
    ordinary = (exponent != 0xff) & ((exponent != 0) | ((bits & 0x7fffffff) == 0))
    half_exponent = np.where(exponent > 0, exponent, 1) - 151 - power_of_two.astype(np.int64)
    return np.ldexp(1.0, half_exponent.astype(np.int32)), ordinary


@pytest.mark.parametrize("alpha", [True, False])
@pytest.mark.parametrize("sigma", [3.0, 5.5])
def test_fused_sums_decide_the_float_outside_the_bound(im, refmod, sigma, alpha):
    rng = np.random.default_rng(int(sigma * 10) + (1 if alpha else 0))
    rows, cols = 23, 211
    px = (rng.random((rows, cols, 4)) * 65535.0).astype(np.float32)
    px[3:9, 20:80, 3] = (10.0 ** rng.uniform(-6, 0, (6, 60))).astype(np.float32)      # tiny alpha
    px[12:16, 100:160, 3] = 0.0
    px[18:, :, :3] = (rng.random((rows - 18, cols, 3)) * 90000.0 - 12000.0).astype(np.float32)   # beyond the range
    kernel = "Blur:0x%g" % sigma
    values, kx, ky, _ = im.kernel_to_numpy(kernel)
    assert values.shape[0] == 1
    taps = values[0]
    K = taps.size
    assert K >= 16 and (taps >= 0).all() and taps.sum() <= 1.0 + 1e-9       # what the launcher requires
    error_unit = (2 * K + 6) * ULP
    window = taps[::-1]
    shift = K - 1 - kx
    p = px.astype(np.longdouble)
    if alpha:
        want = refmod.RefImage(px).morphology("Convolve", 1, kernel).numpy()
        planes = [p[:, :, 3] * p[:, :, c] for c in range(3)] + [p[:, :, 3]]
    else:
        want = np.stack([refmod.RefImage(px[:, :, c].copy()).morphology("Convolve", 1, kernel).numpy()
                         .reshape(rows, cols) for c in range(4)], axis=2)
        planes = [p[:, :, c] for c in range(4)]
    sums, most = [], []
    for q in planes:
        padded = np.pad(q, ((0, 0), (shift, K - 1 - shift)), mode="edge")
        s = np.zeros((rows, cols), dtype=np.longdouble)
        m = np.zeros((rows, cols))
        for u in range(K):
            s += np.longdouble(window[u]) * padded[:, u:u + cols]
            m = np.maximum(m, np.abs(padded[:, u:u + cols]).astype(np.float64))
        sums.append(s)
        most.append(m)
    undecided = total = 0
    for c in range(4):
        weighted = alpha and c != 3
        error = error_unit * most[c]
        if weighted:
            sa = sums[3].astype(np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                inverse = np.where(sa != 0.0, 1.0 / sa, 0.0)
            value = sums[c].astype(np.float64) * inverse
            error = (np.abs(value) * (error_unit * most[3]) + error) * np.abs(inverse) + np.abs(value) * 1.0e-15
            unsure = (sa != 0.0) & ~(np.abs(sa / 65535.0) >= 1.000001e-12)
        else:
            value = sums[c].astype(np.float64)
            unsure = np.zeros((rows, cols), dtype=bool)
        nearest = value.astype(np.float32)
        half_ulp, ordinary = half_ulp_of(nearest)
        distance = np.abs(value - nearest.astype(np.float64))
        decided = ordinary & (half_ulp - distance > error) & ~unsure
        differs = nearest.view(np.uint32) != want[:, :, c].view(np.uint32)
        assert not (differs & decided).any(), "sigma %g channel %d: %d floats differ outside the bound" % (
            sigma, c, int((differs & decided).sum()))
        undecided += int((~decided).sum())
        total += decided.size
    assert undecided <= 0.02 * total, (undecided, total)
