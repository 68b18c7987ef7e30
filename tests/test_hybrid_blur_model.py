"""The argument behind the FAST BlurImage kernel (convolve_fused_hybrid.hip), checked on the CPU
against the compiled reference: the row pass hands the column pass the reference's OWN alpha level
(exact integer sums on the device) and the UNROUNDED colour quotient (f16 sums on the device, here:
extended precision plus an injected error of the size the f16 path makes), the column pass rounds
once.  Claim: every sample within +-1 level of the reference's two-pass result, on any content —
random and tiny alpha, transparent bands, checkerboards whose row-pass values sit on exact rounding
ties (the frames that gave round 2's rounded f16 intermediate its +-2)."""
import numpy as np
import pytest

F16_ERROR = 0.04        # levels: what the hi/lo f16 operands + f32 accumulation may contribute per pass


def frames(rng, rows, cols):
    px = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    yield "random", px
    tiny = px.copy()
    tiny[:, :, 3] = rng.integers(0, 4, (rows, cols))
    tiny[4:9, 30:90, 3] = 0
    yield "alpha of 0..3 levels", tiny
    board = np.zeros((rows, cols, 4), dtype=np.uint16)
    yy, xx = np.mgrid[0:rows, 0:cols]
    even = ((xx + yy) & 1) == 0
    board[:, :, 0] = np.where(even, 1000, 1001)          # every row-pass value on a .5 tie
    board[:, :, 1] = np.where(even, 40000, 40003)
    board[:, :, 2] = np.where(xx & 1, 7, 8)
    board[:, :, 3] = np.where(even, 65535, 65534)
    yield "checkerboard of adjacent levels", board
    half = board.copy()
    half[:, :, 3] = np.where(xx & 1, 0, 2)               # zero centre alpha beside tiny alpha
    yield "checkerboard under alpha 0 / 2", half
    sparse = px.copy()
    sparse[:, :, 3] = 0
    sparse[::5, ::7, 3] = rng.integers(1, 65536, sparse[::5, ::7, 3].shape)
    yield "sparse alpha", sparse


@pytest.mark.parametrize("sigma", [2.0, 10.0])
def test_exact_alpha_and_unrounded_colour_stay_within_one_level(im, refmod, sigma):
    rng = np.random.default_rng(int(10 * sigma))
    rows, cols = 37, 181
    values, kx, ky, _ = im.kernel_to_numpy("Blur:0x%g" % sigma)
    taps = values[0].astype(np.longdouble)
    K = taps.size
    window = taps[::-1]
    shift = K - 1 - kx

    def along(plane, axis):
        pad = [(0, 0), (0, 0)]
        pad[axis] = (shift, K - 1 - shift)
        padded = np.pad(plane, pad, mode="edge")
        out = np.zeros(plane.shape, dtype=np.longdouble)
        for u in range(K):
            out += window[u] * (padded[:, u:u + cols] if axis == 1 else padded[u:u + rows, :])
        return out

    for name, px in frames(rng, rows, cols):
        want = refmod.RefImage(px).blur(0.0, sigma).numpy().astype(np.int64)
        row_pass = refmod.RefImage(px).morphology("Convolve", 1, "Blur:0x%g" % sigma).numpy()
        p = px.astype(np.longdouble)
        alpha_level = row_pass[:, :, 3].astype(np.longdouble)       # the reference's own level: exact on the device
        total = along(p[:, :, 3], 1)
        got = np.zeros_like(want)
        weight_sum = along(alpha_level, 0)
        got[:, :, 3] = np.clip(np.floor(weight_sum + 0.5 + rng.uniform(-F16_ERROR, F16_ERROR, weight_sum.shape)),
                               0, 65535).astype(np.int64)
        for c in range(3):
            numerator = along(p[:, :, 3] * p[:, :, c], 1)
            with np.errstate(divide="ignore", invalid="ignore"):
                quotient = np.where(total > 0, numerator / np.where(total > 0, total, 1), 0)
            quotient = quotient + rng.uniform(-F16_ERROR, F16_ERROR, quotient.shape)      # the row pass's f16 error
            column = along(alpha_level * quotient, 0)
            with np.errstate(divide="ignore", invalid="ignore"):
                value = np.where(weight_sum > 0, column / np.where(weight_sum > 0, weight_sum, 1), 0)
            value = value + rng.uniform(-F16_ERROR, F16_ERROR, value.shape)               # the column pass's
            got[:, :, c] = np.clip(np.floor(value + 0.5), 0, 65535).astype(np.int64)
        diff = np.abs(got - want)
        assert diff.max() <= 1, "%s, sigma %g: max |diff| = %d at %s" % (
            name, sigma, diff.max(), np.argwhere(diff > 1)[:3].tolist())
