#!/usr/bin/env python3
"""Numerical model (CPU, NumPy) of the exact integer matrix-core blur pass sketched in DESIGN.md
section 8: Q16 samples as bytes, the taps as four balanced signed 8-bit digits, every digit product
accumulated exactly in integers (what v_mfma_i32_16x16x64_i8 does), the weight classes combined in
fp64, and a tie window that scales with 1/alpha deciding which results must be recomputed in the
reference's order.  It checks, on random and on adversarial rows, that every result outside the
window rounds to the level the reference's own fp64 loop (morphology.c:2746-2764, restated here
operation by operation) produces, and prints how many samples fall inside the window.

    python tools/model_exact_i8.py [sigma] [rows]
"""
import math
import sys

import numpy as np

sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = 2048
QS = 1.0 / 65535.0


def blur_taps(sigma):
    """GetOptimalKernelWidth1D + the normalised 1-D Gaussian of AcquireKernelBuiltIn (gem.c:262-345,
    morphology.c:1113-1160), enough of it for this model."""
    width = 5
    while True:
        j = (width - 1) // 2
        k = np.exp(-(np.arange(-j, j + 1, dtype=np.float64) ** 2) / (2.0 * sigma * sigma)) / (math.sqrt(2 * math.pi) * sigma)
        if int(65535.0 * (k[0] / k.sum())) <= 0:
            break
        width += 2
    width -= 2
    j = (width - 1) // 2
    k = np.exp(-(np.arange(-j, j + 1, dtype=np.float64) ** 2) / (2.0 * sigma * sigma)) / (math.sqrt(2 * math.pi) * sigma)
    return k / k.sum()


def reference_pass(px, taps):
    """The reference's row loop for alpha-weighted RGBA: every multiply and add rounded separately."""
    K = len(taps)
    pad = np.pad(px, ((0, 0), (K // 2, K // 2), (0, 0)), mode="edge").astype(np.float64)
    n = px.shape[1]
    colour = np.zeros(px.shape[:2] + (3,))
    gamma = np.zeros(px.shape[:2])
    plain_alpha = np.zeros(px.shape[:2])
    for v in range(K):
        a = pad[:, v:v + n, 3]
        alpha = QS * a
        w = alpha * taps[v]
        for c in range(3):
            colour[:, :, c] = colour[:, :, c] + w * pad[:, v:v + n, c]
        gamma = gamma + w
        plain_alpha = plain_alpha + taps[v] * a
    g = np.where(np.abs(gamma) >= 1e-12, 1.0 / np.where(gamma == 0, 1, gamma), np.sign(gamma + 1e-300) / 1e-12)
    out = np.empty(px.shape, dtype=np.float64)
    for c in range(3):
        out[:, :, c] = g * colour[:, :, c]
    out[:, :, 3] = plain_alpha
    return np.clip(np.floor(out + 0.5), 0, 65535), out


def integer_pass(px, taps):
    """Digit products accumulated exactly (Python/NumPy int64 stands in for the i32 tiles: the
    bound 79*255*127 < 2^22 per tile is asserted), then the classes combined in fp64."""
    K = len(taps)
    # the finest fixed point whose largest tap still fits four balanced digits (-2^31 .. 2^31-129)
    frac_bits = int(math.floor(math.log2((2.0 ** 31 - 129.0) / taps.max())))
    q = np.rint(taps * (1 << frac_bits)).astype(np.int64)            # fixed-point taps
    digits = []
    rest = q.copy()
    for _ in range(4):                                                # balanced digits -128..127
        d = ((rest + 128) % 256) - 128
        digits.append(d)
        rest = (rest - d) >> 8
    assert (rest == 0).all(), "taps need more than four digits"
    pad = np.pad(px, ((0, 0), (K // 2, K // 2), (0, 0)), mode="edge").astype(np.int64)
    n = px.shape[1]
    prod = pad[:, :, :3] * pad[:, :, 3:4]                             # alpha*p, 32 bits
    sums = []
    for values, nbytes in ((prod, 4), (pad[:, :, 3:4], 2)):
        total = np.zeros(values.shape[:1] + (n, values.shape[2]), dtype=object)
        for i in range(nbytes):
            byte = (values >> (8 * i)) & 0xff
            signed = byte - 128                                       # x ^ 0x80 as a signed byte
            for j, d in enumerate(digits):
                tile = np.zeros(values.shape[:1] + (n, values.shape[2]), dtype=np.int64)
                for v in range(K):
                    tile += signed[:, v:v + n, :] * d[v]
                assert np.abs(tile).max() < (1 << 22)
                tile += 128 * int(d.sum())                            # the offset: a constant per digit
                total = total + tile.astype(object) * (1 << (8 * (i + j)))
        sums.append(total)
    scale = float(1 << frac_bits)
    s_colour = np.array(sums[0], dtype=np.float64) / scale           # sum k*alpha*p
    s_alpha = np.array(sums[1], dtype=np.float64)[:, :, 0] / scale   # sum k*alpha
    value = np.empty(px.shape, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        for c in range(3):
            value[:, :, c] = np.where(s_alpha > 0, s_colour[:, :, c] / s_alpha, 0.0)
    value[:, :, 3] = s_alpha
    # tap quantisation: |error of sum k*x| <= K * 2^-(frac_bits+1) * max x
    err_colour = K * 2.0 ** -(frac_bits + 1) * 65535.0 * 65535.0
    err_alpha = K * 2.0 ** -(frac_bits + 1) * 65535.0
    with np.errstate(divide="ignore"):
        window = np.empty(px.shape, dtype=np.float64)
        bound = np.where(s_alpha > 0, (err_colour + 65535.0 * err_alpha) / np.maximum(s_alpha, 1e-30), np.inf)
        for c in range(3):
            window[:, :, c] = bound + 1e-9
        window[:, :, 3] = err_alpha + 1e-9
    return value, window


def run(name, px, taps):
    want, exact_value = reference_pass(px, taps)
    value, window = integer_pass(px, taps)
    level = np.clip(np.floor(value + 0.5), 0, 65535)
    frac = value + 0.5 - np.floor(value + 0.5)
    doubtful = np.minimum(frac, 1.0 - frac) <= window                # within the window of a rounding tie
    wrong = (level != want) & ~doubtful
    print("%-28s samples %8d   doubtful %7d (%.4f %%)   wrong outside the window %d   max |value - reference| %.2e" %
          (name, px.size, int(doubtful.sum()), 100.0 * doubtful.mean(), int(wrong.sum()),
           float(np.nanmax(np.abs(np.where(np.isfinite(value), value, 0) - np.where(np.isfinite(exact_value), exact_value, 0))))))
    return int(wrong.sum())


taps = blur_taps(sigma)
print("sigma %g: %d taps" % (sigma, len(taps)))
rng = np.random.default_rng(5)
bad = 0
bad += run("random RGBA", rng.integers(0, 65536, (rows, W, 4)), taps)
opaque = rng.integers(0, 65536, (rows, W, 4)); opaque[:, :, 3] = 65535
bad += run("opaque", opaque, taps)
tiny = rng.integers(0, 65536, (rows, W, 4)); tiny[:, :, 3] = rng.integers(0, 4, (rows, W))
bad += run("alpha 0..3", tiny, taps)
checker = np.empty((rows, W, 4), dtype=np.int64)
checker[:] = ((np.add.outer(np.arange(rows), np.arange(W)) % 2) * 40000 + 100)[:, :, None]
bad += run("checkerboard (exact ties)", checker, taps)
sys.exit(1 if bad else 0)
