"""Multi-GPU host logic.  One process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

* Independent images (BASELINE config C4, and the benchmark): `shard_range` gives
  each rank a contiguous slice of the batch.  No data-path collective.
* One image row-sharded across ranks, global-histogram operators
  (EqualizeImage / ContrastStretchImage): every rank bins its own band on its GPU,
  the (MaxMap+1) x channels table is summed with ONE all-reduce — the only
  collective of the design — every rank builds the identical LUT and applies it to its
  band: on the device for device images (MagickHipApplyHistogram; the table never leaves
  HBM), with the library's host LUT builders for host arrays (the CPU tests).
* Stencil operators on a row-sharded image need halo rows: `band_with_halo` gives
  the row range a rank must hold (own rows + kernel reach), the caller uploads
  overlapping bands (the pixel cache lives on the host).

Nothing here computes pixels on the CPU: histogram and LUT application go through
libmagickhip.so (`imagemagick_amd.histogram` / `apply_lut`).
"""
import numpy as np

from . import _lib


def shard_range(items, rank, world):
    """Contiguous, balanced [begin, end) slice of `items` work units for `rank`."""
    base, extra = divmod(int(items), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def band_with_halo(rows, rank, world, reach_above, reach_below):
    """Rows [own_begin, own_end) owned by `rank` and the rows [lo, hi) it must hold to
    evaluate a stencil that reads `reach_above` rows above / `reach_below` rows below
    (BlurImage sigma=10: 39/39; Disk:15: 15/15).  Edge bands clamp at the image border,
    where the kernels' own edge clamp (cache.c:2663-2679) takes over."""
    begin, end = shard_range(rows, rank, world)
    return (begin, end), (max(0, begin - reach_above), min(rows, end + reach_below))


def stencil_reach(operator, **kw):
    """Rows a stencil operator reads above / below an output row: (reach_above, reach_below).
    blur / unsharp: both separable passes reach (K-1)/2 rows (only the column pass crosses
    bands, but it consumes row-pass output, which is computed band-locally from the same
    rows — so a band needs exactly the column reach); morphology: the kernel's extent around
    its origin, times the number of primitive applications."""
    import imagemagick_amd as im
    if operator in ("blur", "unsharp"):
        width = im.optimal_kernel_width_1d(kw.get("radius", 0.0), kw["sigma"])
        return (width - 1) // 2, (width - 1) // 2
    if operator == "morphology":
        values, kx, ky, count = im.kernel_to_numpy(kw["kernel"])
        if count != 1:
            raise ValueError("row-sharded morphology takes a single kernel")
        height = values.shape[0]
        stages = {"erode": 1, "dilate": 1, "convolve": 1, "open": 2, "close": 2, "smooth": 4, "edgein": 1,
                  "edgeout": 1, "edge": 1, "tophat": 2, "bottomhat": 2}[kw["method"].lower()]
        n = stages * max(1, int(kw.get("iterations", 1)))
        # the reflected kernel of Dilate / Convolve swaps the two reaches: take the larger for both
        reach = max(ky, height - 1 - ky)
        return n * reach, n * reach
    raise ValueError(operator)


def run_on_band(pixels, rank, world, reach, operator):
    """One rank's share of a row-sharded stencil operator (BASELINE config C5 on 8 GPUs): take
    the rows this rank owns plus its halo out of the host image `pixels` ([rows, cols, ch] NumPy),
    upload, run `operator(image) -> image` on the GPU, and return (own_begin, own_end, rows) —
    the owned output rows as a NumPy array.  Bands at the image border rely on the kernels' own
    edge clamp; interior bands never see a clamped row inside their owned range because the halo
    covers the operator's full reach."""
    import torch
    import imagemagick_amd as im
    rows = pixels.shape[0]
    (begin, end), (lo, hi) = band_with_halo(rows, rank, world, reach[0], reach[1])
    band = np.ascontiguousarray(pixels[lo:hi])
    if band.dtype == np.uint16:
        dev = torch.from_numpy(band.view(np.int16)).cuda().view(torch.uint16)
    else:
        dev = torch.from_numpy(band).cuda()
    out = operator(im.Image(dev)).numpy()
    return begin, end, out[begin - lo:end - lo]


def all_reduce_histogram(histogram, group=None):
    """Sum a [65536, channels] count table over the process group, in place.
    Accepts a CUDA int64 tensor (RCCL) or a NumPy uint64 array (gloo, CPU tests)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return histogram                                             # a single rank: nothing to sum
    if isinstance(histogram, np.ndarray):
        t = torch.from_numpy(histogram.view(np.int64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return histogram
    if histogram.is_cuda and dist.get_backend(group) == "gloo":     # CPU-only collective backend
        host = histogram.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        histogram.copy_(host)
        return histogram
    dist.all_reduce(histogram, op=dist.ReduceOp.SUM, group=group)
    return histogram


def equalize_band(image, group=None, total_rows=0):
    """EqualizeImage on this rank's band of a row-sharded image (enhance.c:2040-2280):
    local histogram on the GPU -> all-reduce -> identical LUT on every rank -> apply.
    Device images never leave the device: the table is all-reduced where it lies (RCCL) and
    the LUT is built and applied by the library's device kernels."""
    import imagemagick_amd as im
    sync = (image.channel_mask & _lib.SYNC_CHANNELS) != 0
    hist = im.histogram(image, sync)
    all_reduce_histogram(hist, group)
    if not isinstance(hist, np.ndarray):
        return im.apply_histogram(image, hist, sync, True, image_rows=total_rows)
    lut, mask = im.equalize_lut(hist, image.quantum)
    return im.apply_lut(image, lut, mask)


def contrast_stretch_band(image, total_columns, total_rows, black_point, white_point, group=None):
    """ContrastStretchImage on this rank's band (enhance.c:1544-1818); black/white points
    are pixel counts of the WHOLE image, as the MagickCore API defines them."""
    import imagemagick_amd as im
    mode = image.channel_mask == _lib.ALL_CHANNELS
    hist = im.histogram(image, mode)
    all_reduce_histogram(hist, group)
    if not isinstance(hist, np.ndarray):
        return im.apply_histogram(image, hist, mode, False, black_point, white_point, image_rows=total_rows)
    lut, mask = im.contrast_stretch_lut(hist, total_columns, total_rows, black_point, white_point,
                                        image.quantum)
    return im.apply_lut(image, lut, mask)
