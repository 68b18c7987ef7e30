"""The geometry of the four-row-band form (operators.cpp fused_blur_gray_bands, morphology.hip try_rects_gray_bands,
pointwise.hip gray_bands_pack_kernel / gray_bands_unpack_kernel), restated in NumPy: a one-channel frame's rows cut into
four bands = the four channels of a frame a quarter as tall, `halo` extra rows either side of every band, the frame's
edge rows repeated above the first and below the last band.  Any translation-invariant stencil that clamps at the edges
and treats channels one by one (morphology.c:2892-2979 Convolve, :3060-3199 Erode / Dilate) gives, on the packed frame's
rows [halo, halo + band), exactly what it gives on the frame itself.  No GPU."""
import numpy as np
import pytest


def pack(frame, halo):
    rows = frame.shape[0]
    band = (rows + 3) // 4
    packed = np.empty((band + 2 * halo, frame.shape[1], 4), dtype=frame.dtype)
    for c in range(4):
        source = np.clip(c * band + np.arange(band + 2 * halo) - halo, 0, rows - 1)
        packed[:, :, c] = frame[source]
    return packed, band


def unpack(packed, rows, band, halo):
    out = np.empty((rows, packed.shape[1]), dtype=packed.dtype)
    for c in range(4):
        count = min(band, rows - c * band)
        if count > 0:
            out[c * band: c * band + count] = packed[halo: halo + count, :, c]
    return out


def stencil(plane, cells, origin, reduce):
    """cells: (h, w) weights (NaN = not part of the kernel); origin (y, x); edge clamp per axis."""
    rows, cols = plane.shape
    h, w = cells.shape
    terms = []
    for v in range(h):
        for u in range(w):
            if np.isnan(cells[v, u]):
                continue
            ys = np.clip(np.arange(rows) + v - origin[0], 0, rows - 1)
            xs = np.clip(np.arange(cols) + u - origin[1], 0, cols - 1)
            terms.append(cells[v, u] * plane[np.ix_(ys, xs)].astype(np.float64))
    return reduce(np.stack(terms), axis=0)


@pytest.mark.parametrize("rows", [8, 37, 64, 131])
@pytest.mark.parametrize("shape,origin", [((5, 3), (2, 1)), ((7, 7), (3, 3)), ((5, 5), (1, 3)), ((1, 9), (0, 4)), ((9, 1), (6, 0))])
@pytest.mark.parametrize("reduce", [np.sum, np.max, np.min])
def test_four_row_bands_reproduce_the_frame(rows, shape, origin, reduce):
    rng = np.random.default_rng(rows * 100 + shape[0] * 10 + shape[1])
    frame = rng.integers(0, 65536, (rows, 23)).astype(np.float64)
    cells = rng.random(shape)
    cells[rng.random(shape) < 0.2] = np.nan
    cells[origin] = 1.0
    halo = max(origin[0], shape[0] - 1 - origin[0])
    packed, band = pack(frame, halo)          # (bands shorter than 2 * halo too: the library declines those for their cost only)
    want = stencil(frame, cells, origin, reduce)
    result = np.stack([stencil(packed[:, :, c], cells, origin, reduce) for c in range(4)], axis=2)
    got = unpack(result, rows, band, halo)
    assert np.array_equal(got, want)


def test_pack_and_unpack_are_inverse_on_the_frames_own_rows():
    frame = np.arange(11 * 5, dtype=np.uint16).reshape(11, 5)
    packed, band = pack(frame, 2)
    assert packed.shape == (band + 4, 5, 4) and band == 3
    assert np.array_equal(unpack(packed, 11, band, 2), frame)
    assert np.array_equal(packed[0, :, 0], frame[0]) and np.array_equal(packed[-1, :, 3], frame[10])   # clamped
    assert np.array_equal(packed[0, :, 1], frame[1])                                                  # the neighbouring band's rows
