#!/usr/bin/env python3
"""Functional model of convolve_fused.hip's index maps (no GPU needed): LDS planes, the lane ->
operand-line / accumulator-register maps of v_mfma_f32_32x32x16_f16, the ring-group schedule
and the work-item decomposition, executed lane by lane in NumPy with float64 "matrix cores"
(operands unsplit), and compared with the oracle's BlurImage.  Index bugs show up here; the
f16 operand split and the f32 accumulation are the GPU tests' business.

    python tools/model_blur_fused.py [rows cols sigma]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def layout(extent, units, channel_major):
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]

    def ok(S, PAD):
        CH = units * S + PAD
        for g in groups:
            seen = set()
            for lane in g:
                ch, unit = (lane >> 3, lane & 7) if channel_major else (lane & 3, lane >> 2)
                slot = (((ch * CH + unit * S) * 2) % 256) // 16
                if slot in seen:
                    return False
                seen.add(slot)
        return True
    for S in range(extent, extent + 65, 8):
        for PAD in range(8, 65, 8):
            if ok(S, PAD):
                return S, PAD
    return extent, 8


def mfma(acc, a_lines, b_lines):
    """acc[lane][reg] += A x B for one 32x32x16 product.  a_lines[lane] / b_lines[lane]: the 8
    operand values of a lane.  A: row = lane&31, k = 8*(lane>>5)+i; B: column = lane&31, same k;
    D: column = lane&31, row = (reg&3)+8*(reg>>2)+4*(lane>>5)."""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    for lane in range(64):
        A[lane & 31, 8 * (lane >> 5):8 * (lane >> 5) + 8] = a_lines[lane]
        B[8 * (lane >> 5):8 * (lane >> 5) + 8, lane & 31] = b_lines[lane]
    D = A @ B
    for lane in range(64):
        for reg in range(16):
            acc[lane, reg] += D[(reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), lane & 31]


def quantize(v):
    v = np.where(np.isnan(v) | (v <= 0), 0.0, v)
    return np.minimum(np.floor(v + 0.5), 65535.0)


def run(px, taps, origin, blend=True, segments=1):
    H, W, _ = px.shape
    K = len(taps)
    shift = K - 1 - origin
    rev = taps[::-1].copy()                       # taps[v] multiplies input o-shift+v
    NQ = max(3, (K + 31 + 15) // 16)
    assert NQ <= 7
    COLS, GROUP, BLOCK = 64, 16, 32
    RC, XS = 16 * NQ, 16 * NQ + 32
    SR, PADR = layout(XS, GROUP, True)
    SC, PADC = layout(RC, COLS, False)
    CHR, CHC = GROUP * SR + PADR, COLS * SC + PADC
    lds = 2 * 2 * 4 * (CHR + CHC)
    GPR = XS // 4
    FETCH_GROUPS = GROUP * GPR
    strips = (W + COLS - 1) // COLS
    blocks = (H + BLOCK - 1) // BLOCK
    bps = (blocks + segments - 1) // segments
    segments = (blocks + bps - 1) // bps
    out = np.zeros((H, W, 4))
    lanes = np.arange(64)
    n_of, half_of = lanes & 31, lanes >> 5
    # Toeplitz operands per chunk and lane
    T = np.zeros((NQ, 64, 8))
    for q in range(NQ):
        for lane in range(64):
            for i in range(8):
                j = 16 * q + 8 * half_of[lane] + i - n_of[lane]
                T[q, lane, i] = rev[j] if 0 <= j < K else 0.0

    def samples(p):                               # [.., 4] Quantum -> sample values
        p = p.astype(np.float64)
        if blend:
            v = p.copy()
            v[..., :3] = p[..., :3] * p[..., 3:4]     # alpha*p (scale factors dropped)
            return v
        return p

    def epilogue(s):                              # [.., 4] sums -> Quantum
        if blend:
            with np.errstate(divide="ignore", invalid="ignore"):
                c = s[..., :3] / s[..., 3:4]
            return quantize(np.concatenate([c, s[..., 3:4]], axis=-1))
        return quantize(s)

    for item in range(strips * segments):
        segment, strip = divmod(item, strips)
        x0 = COLS * strip
        block_begin = segment * bps
        block_end = min(block_begin + bps, blocks)
        nblocks = block_end - block_begin
        out_begin = BLOCK * block_begin
        in0, xin0 = out_begin - shift, x0 - shift
        ngroups = 2 * (nblocks - 1) + NQ
        ring = np.full(4 * CHC, np.nan)
        stage = np.full(4 * CHR, np.nan)
        for g in range(ngroups):
            stage[:] = np.nan
            for idx in range(FETCH_GROUPS):
                row, xg = divmod(idx, GPR)
                y = min(max(in0 + GROUP * g + row, 0), H - 1)
                for i in range(4):
                    x = min(max(xin0 + 4 * xg + i, 0), W - 1)
                    v = samples(px[y, x])
                    for c in range(4):
                        stage[c * CHR + row * SR + 4 * xg + i] = v[c]
            for wave in range(4):
                rmg, rng = wave & 1, wave >> 1
                acc = np.zeros((64, 16))
                for q in range(NQ):
                    a = np.zeros((64, 8))
                    for lane in range(64):
                        n, half = n_of[lane], half_of[lane]
                        entry = (n >> 3) * CHR + (8 * rmg + (n & 7)) * SR + 32 * rng + 8 * half + 16 * q
                        a[lane] = stage[entry:entry + 8]
                    assert not np.isnan(a).any()
                    mfma(acc, a, T[q])
                for lane in range(64):
                    n, half = n_of[lane], half_of[lane]
                    sums = np.stack([acc[lane, 4 * c + np.arange(4)] for c in range(4)], axis=-1)   # [i][c]
                    v = samples(epilogue(sums))
                    slot = (g % NQ) * GROUP + 8 * rmg + 4 * half
                    at0 = (32 * rng + n) * SC + slot
                    for c in range(4):
                        ring[c * CHC + at0:c * CHC + at0 + 4] = v[:, c]
            if g >= NQ - 1 and ((g - (NQ - 1)) & 1) == 0:
                block = (g - (NQ - 1)) >> 1
                for wave in range(4):
                    accs = []
                    for t in range(2):
                        cg = 2 * wave + t
                        acc = np.zeros((64, 16))
                        group = (2 * block) % NQ
                        for q in range(NQ):
                            a = np.zeros((64, 8))
                            for lane in range(64):
                                n, half = n_of[lane], half_of[lane]
                                at = (n & 3) * CHC + (n >> 2) * SC + 8 * half + 8 * cg * SC + GROUP * group
                                a[lane] = ring[at:at + 8]
                            assert not np.isnan(a).any()
                            mfma(acc, a, T[q])
                            group = 0 if group + 1 == NQ else group + 1
                        accs.append(acc)
                    for t in range(2):
                        result = np.stack([epilogue(accs[t][:, 4 * pg:4 * pg + 4]) for pg in range(4)], axis=1)  # [lane][pg][c]
                        for pair in range(2):
                            # v_permlane32_swap(vdst = result[pair], src0 = result[pair+2]): lanes 32..63 of
                            # vdst <-> lanes 0..31 of src0
                            vdst, src0 = result[:, pair].copy(), result[:, pair + 2].copy()
                            new_vdst, new_src0 = vdst.copy(), src0.copy()
                            new_vdst[32:] = src0[:32]
                            new_src0[:32] = vdst[32:]
                            for lane in range(64):
                                n, half = n_of[lane], half_of[lane]
                                x = x0 + 8 * (2 * wave + t) + 4 * half + 2 * pair
                                y = out_begin + BLOCK * block + n
                                if y < H:
                                    if x < W:
                                        out[y, x] = new_vdst[lane]
                                    if x + 1 < W:
                                        out[y, x + 1] = new_src0[lane]
    return out.astype(np.uint16), lds


def mfma16(acc, a_lines, b_lines):
    """acc[lane][reg] += A x B for one 16x16x32 product.  A: row = lane&15, k = 8*(lane>>4)+i;
    B: column = lane&15, same k; D: column = lane&15, row = 4*(lane>>4)+reg."""
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for lane in range(64):
        A[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = a_lines[lane]
        B[8 * (lane >> 4):8 * (lane >> 4) + 8, lane & 15] = b_lines[lane]
    D = A @ B
    for lane in range(64):
        for reg in range(4):
            acc[lane, reg] += D[4 * (lane >> 4) + reg, lane & 15]


def run16(px, taps, origin, blend=True, segments=1):
    """blur_fused16_kernel: 16 waves, 16x16x32 tiles; a column tile reads NG = 2*NC ring groups, the
    ring holds NR = NG+1; iteration g: stage g, column block g-NG, row group g."""
    H, W, _ = px.shape
    K = len(taps)
    shift = K - 1 - origin
    rev = taps[::-1].copy()
    NC = (K + 15 + 31) // 32
    assert NC <= 3
    COLS, GROUP = 64, 16
    NG = 2 * NC
    NR = NG + 1
    RC, XS = GROUP * NR, 32 * NC + 48
    SR, PADR, SC, PADC = XS, 64, RC, 32            # what fused16_layout picks
    CHR, CHC = GROUP * SR + PADR, COLS * SC + PADC
    lds = 2 * 2 * 4 * (CHR + CHC)
    GPR = XS // 4
    FETCH_GROUPS = GROUP * GPR
    strips = (W + COLS - 1) // COLS
    blocks = (H + GROUP - 1) // GROUP
    bps = (blocks + segments - 1) // segments
    segments = (blocks + bps - 1) // bps
    out = np.zeros((H, W, 4))
    lanes = np.arange(64)
    n_of, kq_of = lanes & 15, lanes >> 4
    T = np.zeros((NC, 64, 8))
    for c in range(NC):
        for lane in range(64):
            for i in range(8):
                j = 32 * c + 8 * kq_of[lane] + i - n_of[lane]
                T[c, lane, i] = rev[j] if 0 <= j < K else 0.0

    def samples(p):
        p = p.astype(np.float64)
        if blend:
            v = p.copy()
            v[..., :3] = p[..., :3] * p[..., 3:4]
            return v
        return p

    for item in range(strips * segments):
        segment, strip = divmod(item, strips)
        x0 = COLS * strip
        block_begin = segment * bps
        block_end = min(block_begin + bps, blocks)
        nblocks = block_end - block_begin
        out_begin = GROUP * block_begin
        in0, xin0 = out_begin - shift, x0 - shift
        ngroups = nblocks + NG - 1
        ring = np.full(4 * CHC, np.nan)
        for g in range(ngroups + 1):
            if g < ngroups:
                stage = np.full(4 * CHR, np.nan)
                for tid in range(FETCH_GROUPS):
                    row, xg = divmod(tid, GPR)
                    y = min(max(in0 + GROUP * g + row, 0), H - 1)
                    for i in range(4):
                        x = min(max(xin0 + 4 * xg + i, 0), W - 1)
                        v = samples(px[y, x])
                        for c in range(4):
                            stage[c * CHR + row * SR + 4 * xg + i] = v[c]
            if g >= NG:
                block = g - NG
                for wave in range(16):
                    acc = np.zeros((64, 4))
                    for c in range(NC):
                        a = np.zeros((64, 8))
                        for lane in range(64):
                            n, kq = n_of[lane], kq_of[lane]
                            wide = ((g % NR) + 1) % NR + 2 * c + (kq >> 1)      # block mod NR = (g+1) mod NR
                            group = wide if wide < NR else wide - NR
                            at = (n & 3) * CHC + (4 * wave + (n >> 2)) * SC + 8 * (kq & 1) + GROUP * group
                            a[lane] = ring[at:at + 8]
                        assert not np.isnan(a).any()
                        mfma16(acc, a, T[c])
                    for lane in range(64):
                        n, kq = n_of[lane], kq_of[lane]
                        s4 = acc[lane]
                        if blend:
                            with np.errstate(divide="ignore", invalid="ignore"):
                                r = quantize(np.concatenate([s4[:3] / s4[3], s4[3:4]]))
                        else:
                            r = quantize(s4)
                        x, y = x0 + 4 * wave + kq, out_begin + GROUP * block + n
                        if x < W and y < H:
                            out[y, x] = r
            if g == ngroups:
                break
            for wave in range(16):
                rq, ot = wave & 3, wave >> 2
                acc = np.zeros((64, 4))
                for c in range(NC):
                    a = np.zeros((64, 8))
                    for lane in range(64):
                        n, kq = n_of[lane], kq_of[lane]
                        entry = (n >> 2) * CHR + (4 * rq + (n & 3)) * SR + 16 * ot + 8 * kq + 32 * c
                        a[lane] = stage[entry:entry + 8]
                    assert not np.isnan(a).any()
                    mfma16(acc, a, T[c])
                for lane in range(64):
                    n, kq = n_of[lane], kq_of[lane]
                    if blend:
                        sa = acc[48 + n]                      # __shfl(acc[r], 48+n)
                        level = quantize(sa)
                        if kq == 3:
                            v = level
                        else:
                            with np.errstate(divide="ignore", invalid="ignore"):
                                v = quantize(acc[lane] / sa) * level
                    else:
                        v = quantize(acc[lane])
                    at = kq * CHC + (16 * ot + n) * SC + (g % NR) * GROUP + 4 * rq
                    ring[at:at + 4] = v
    return out.astype(np.uint16), lds


def main():
    from oracle import restate
    rows, cols, sigma = 70, 100, 2.0
    if len(sys.argv) > 3:
        rows, cols, sigma = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
    rng = np.random.default_rng(5)
    px = rng.integers(0, 65536, (rows, cols, 4), dtype=np.uint16)
    taps = restate.blur_kernel(0.0, sigma)
    K = len(taps)
    for form, segments in ((run, 1), (run, 2), (run16, 1), (run16, 2)):
        got, lds = form(px, np.asarray(taps, dtype=np.float64), (K - 1) // 2, True, segments)
        want = restate.blur_image(px, 0.0, sigma)
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        print("%s K=%d segments=%d LDS=%d bytes: max |model - oracle| = %d, identical %.4f"
              % (form.__name__, K, segments, lds, d.max(), (d == 0).mean()))
        assert d.max() <= 1


if __name__ == "__main__":
    main()
