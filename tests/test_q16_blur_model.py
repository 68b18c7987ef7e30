"""The certificate of the Q16 EXACT 1-D convolve on the fused fp64 kernels (convolve.hip, Tie64: the
path of kernels beyond the matrix-core blur's 81 taps and of gray layouts), checked on the CPU: fused
sums over alpha-premultiplied samples, the Quantum of S_c/S_alpha (or S_c) — and every result within
kTieMargin = 1e-6 level of a rounding tie recomputed in the reference's order.  Restated in extended
precision, against the compiled reference: no level may differ outside that margin."""
import numpy as np
import pytest

TIE_MARGIN = 1.0e-6


@pytest.mark.parametrize("alpha", [True, False])
@pytest.mark.parametrize("sigma", [4.0, 14.0])
def test_fused_sums_decide_the_level_outside_the_tie_margin(im, refmod, sigma, alpha):
    rng = np.random.default_rng(int(sigma) + (7 if alpha else 0))
    rows, cols = 19, 257
    px = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    px[3:9, 20:120, 3] = rng.integers(0, 4, (6, 100))            # tiny alpha
    px[12:16, 100:200, 3] = 0
    kernel = "Blur:0x%g" % sigma
    values, kx, ky, _ = im.kernel_to_numpy(kernel)
    taps = values[0]
    K = taps.size
    assert K >= 16 and (taps >= 0).all()
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
    sums = []
    for q in planes:
        padded = np.pad(q, ((0, 0), (shift, K - 1 - shift)), mode="edge")
        s = np.zeros((rows, cols), dtype=np.longdouble)
        for u in range(K):
            s += np.longdouble(window[u]) * padded[:, u:u + cols]
        sums.append(s)
    undecided = total = 0
    for c in range(4):
        weighted = alpha and c != 3
        if weighted:
            sa = sums[3].astype(np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                value = np.where(sa != 0.0, sums[c].astype(np.float64) / sa, 0.0)
            unsure = (sa != 0.0) & ~(sa / 65535.0 >= 1.0e-12)
        else:
            value = sums[c].astype(np.float64)
            unsure = np.zeros((rows, cols), dtype=bool)
        shifted = value + 0.5
        fraction = shifted - np.floor(shifted)
        tie = (fraction < TIE_MARGIN) | (fraction > 1.0 - TIE_MARGIN) | unsure
        level = np.minimum(np.floor(np.maximum(shifted, 0.0)), 65535.0).astype(np.int64)
        differs = level != want[:, :, c].astype(np.int64)
        assert not (differs & ~tie).any(), "sigma %g channel %d: %d levels differ outside the margin" % (
            sigma, c, int((differs & ~tie).sum()))
        undecided += int(tie.sum())
        total += tie.size
    assert undecided <= 0.001 * total, (undecided, total)
