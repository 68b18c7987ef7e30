"""Multi-GPU host logic.  One process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

* Independent images (BASELINE config C4, and the benchmark): `shard_range` gives
  each rank a contiguous slice of the batch.  No data-path collective.
* One image row-sharded across ranks, global-histogram operators
  (EqualizeImage / ContrastStretchImage): every rank bins its own band on its GPU,
  the (MaxMap+1) x channels table is summed with ONE all-reduce — the only
  collective of the design — every rank builds the identical LUT on the host with
  the library's LUT builders and applies it to its band.
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


def all_reduce_histogram(histogram, group=None):
    """Sum a [65536, channels] count table over the process group, in place.
    Accepts a CUDA int64 tensor (RCCL) or a NumPy uint64 array (gloo, CPU tests)."""
    import torch
    import torch.distributed as dist
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


def equalize_band(image, group=None):
    """EqualizeImage on this rank's band of a row-sharded image (enhance.c:2040-2280):
    local histogram on the GPU -> all-reduce -> identical LUT on every rank -> apply."""
    import imagemagick_amd as im
    sync = (image.channel_mask & _lib.SYNC_CHANNELS) != 0
    hist = im.histogram(image, sync)
    all_reduce_histogram(hist, group)
    host = hist.cpu().numpy().view(np.uint64) if not isinstance(hist, np.ndarray) else hist
    lut, mask = im.equalize_lut(host, image.quantum)
    return im.apply_lut(image, lut, mask)


def contrast_stretch_band(image, total_columns, total_rows, black_point, white_point, group=None):
    """ContrastStretchImage on this rank's band (enhance.c:1544-1818); black/white points
    are pixel counts of the WHOLE image, as the MagickCore API defines them."""
    import imagemagick_amd as im
    hist = im.histogram(image, image.channel_mask == _lib.ALL_CHANNELS)
    all_reduce_histogram(hist, group)
    host = hist.cpu().numpy().view(np.uint64) if not isinstance(hist, np.ndarray) else hist
    lut, mask = im.contrast_stretch_lut(host, total_columns, total_rows, black_point, white_point,
                                        image.quantum)
    return im.apply_lut(image, lut, mask)
