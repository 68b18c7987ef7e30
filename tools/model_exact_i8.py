#!/usr/bin/env python3
"""Numerical model (CPU, NumPy) of the exact-integer matrix-core blur pass of
imagemagick_amd/csrc/convolve_fused_exact.hip, with the same constants as its host side
(plan_exact_taps): Q16 samples alpha*p (32 bits) and alpha*2^16 as four signed bytes, the taps as
five balanced signed 8-bit digits of rint(k*2^F), the digit products of weight class i+j >= 3
accumulated exactly in integers (what v_mfma_i32_16x16x64_i8 does), the classes combined in fp64,
and an ambiguity window 65536*(E_N+E_D)/D + 4e-9 that decides which results must be recomputed in
the reference's operation order.  It checks, on random and on adversarial rows, that every result
OUTSIDE the window rounds to the level the reference's own fp64 loop (morphology.c:2746-2764,
restated here operation by operation) produces, and prints how many samples fall inside.

    python tools/model_exact_i8.py [sigma] [rows]
"""
import math
import sys

import numpy as np

sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 64
W = 2048
QS = 1.0 / 65535.0
DIGITS = 5


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


def plan(taps):
    """plan_exact_taps of convolve_fused_exact.hip."""
    K = len(taps)
    F = math.frexp(5.4e11 / taps.max())[1] - 1
    scaled = np.ldexp(taps, F)
    nearest = np.rint(scaled)
    quantisation = float(np.abs(scaled - nearest).sum())
    rest = nearest.astype(np.int64)
    digits = []
    for _ in range(DIGITS):
        d = ((rest + 128) & 255) - 128
        digits.append(d)
        rest = (rest - d) >> 8
    assert (rest == 0).all()
    magnitude = [float(np.abs(d).sum()) for d in digits]
    signed = [float(d.sum()) for d in digits]
    unit = math.ldexp(1.0, -F)

    def offset_of(i0):
        return sum(128.0 * signed[j] * math.ldexp(1.0, 8 * (i + j - 3))
                   for i in range(i0, 4) for j in range(DIGITS) if i + j >= 3)

    def dropped_of(i0):
        return unit * sum(255.0 * magnitude[j] * math.ldexp(1.0, 8 * (i + j))
                          for i in range(i0, 4) for j in range(DIGITS) if i + j <= 2)

    e_colour = quantisation * unit * 65535.0 * 65535.0 + dropped_of(0)
    e_shifted = quantisation * unit * 65535.0 * 65536.0 + dropped_of(2)
    m_unit = math.ldexp(1.0, 24 - F)
    return dict(F=F, digits=digits, offset=offset_of(0), alpha_scale=math.ldexp(1.0, 8 - F),
                alpha_window=e_shifted / 65536.0 + 4.0e-9,
                colour_window=1.002 * 65536.0 * (e_colour + e_shifted) / m_unit,
                alpha_floor=1024.0 * e_shifted / m_unit, e_colour=e_colour, e_shifted=e_shifted,
                smallest_ok=taps.min() * 65536.0 > 8.0 * e_shifted)


def integer_pass(px, taps, p):
    """The kernel's arithmetic: signed bytes x balanced digits, classes i+j >= 3, exact."""
    K = len(taps)
    pad = np.pad(px, ((0, 0), (K // 2, K // 2), (0, 0)), mode="edge").astype(np.int64)
    n = px.shape[1]
    samples = np.empty(pad.shape, dtype=np.int64)
    samples[:, :, :3] = pad[:, :, :3] * pad[:, :, 3:4]                # alpha*p
    samples[:, :, 3] = pad[:, :, 3] << 16                             # alpha*2^16
    M = np.zeros(px.shape, dtype=object)
    for i in range(4):
        signed_byte = ((samples >> (8 * i)) & 0xff) - 128             # b ^ 0x80 as a signed byte
        for j, d in enumerate(p["digits"]):
            if i + j < 3:
                continue
            tile = np.zeros(px.shape, dtype=np.int64)
            for v in range(K):
                tile += signed_byte[:, v:v + n, :] * d[v]
            assert np.abs(tile).max() < (1 << 23)
            M = M + tile.astype(object) * (1 << (8 * (i + j - 3)))
    M = np.array(M, dtype=np.float64) + p["offset"]                   # exact: |M| < 2^53
    Ma = M[:, :, 3]
    value = np.empty(px.shape, dtype=np.float64)
    window = np.empty(px.shape, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(Ma > 0, 1.0 / np.where(Ma > 0, Ma, 1.0), 0.0)
        for c in range(3):
            value[:, :, c] = M[:, :, c] * (65536.0 * r)
            window[:, :, c] = np.where(Ma >= p["alpha_floor"], p["colour_window"] * r + 4.0e-9, np.inf)
            window[:, :, c] = np.where(Ma == 0, 0.0, window[:, :, c])  # all-transparent: level 0, certain
    value[:, :, 3] = Ma * p["alpha_scale"]
    window[:, :, 3] = np.where(Ma == 0, 0.0, np.where(Ma >= p["alpha_floor"], p["alpha_window"], np.inf))
    return value, window


def run(name, px, taps, p):
    want, exact_value = reference_pass(px, taps)
    value, window = integer_pass(px, taps, p)
    shifted = value + 0.5
    level = np.clip(np.floor(shifted), 0, 65535)
    frac = shifted - np.floor(shifted)
    doubtful = (frac < window) | (frac > 1.0 - window)
    wrong = (level != want) & ~doubtful
    print("%-28s samples %8d   doubtful %7d (%.5f %%)   wrong outside the window %d   max |value - reference| %.2e" %
          (name, px.size, int(doubtful.sum()), 100.0 * doubtful.mean(), int(wrong.sum()),
           float(np.nanmax(np.abs(np.where(np.isfinite(value), value, 0) - np.where(np.isfinite(exact_value), exact_value, 0))))))
    return int(wrong.sum())


taps = blur_taps(sigma)
p = plan(taps)
print("sigma %g: %d taps, F = %d, E_colour %.4f, E_shifted %.4f (sample units), opaque colour window %.2e level, eligible %s" %
      (sigma, len(taps), p["F"], p["e_colour"], p["e_shifted"],
       p["colour_window"] / (65535.0 * 65536.0 / math.ldexp(1.0, 24 - p["F"])) + 4e-9, p["smallest_ok"]))
rng = np.random.default_rng(5)
bad = 0
bad += run("random RGBA", rng.integers(0, 65536, (rows, W, 4)), taps, p)
opaque = rng.integers(0, 65536, (rows, W, 4)); opaque[:, :, 3] = 65535
bad += run("opaque", opaque, taps, p)
tiny = rng.integers(0, 65536, (rows, W, 4)); tiny[:, :, 3] = rng.integers(0, 4, (rows, W))
bad += run("alpha 0..3", tiny, taps, p)
sparse = rng.integers(0, 65536, (rows, W, 4)); sparse[:, :, 3] = np.where(rng.random((rows, W)) < 0.01, 65535, 0)
bad += run("alpha sparse (1 %)", sparse, taps, p)
band = rng.integers(0, 65536, (rows, W, 4)); band[:, : W // 2, 3] = 0
bad += run("half transparent", band, taps, p)
checker = np.empty((rows, W, 4), dtype=np.int64)
checker[:] = ((np.add.outer(np.arange(rows), np.arange(W)) % 2) * 40000 + 100)[:, :, None]
bad += run("checkerboard (exact ties)", checker, taps, p)
sys.exit(1 if bad else 0)
