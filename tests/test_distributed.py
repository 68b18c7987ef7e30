"""N > 1 path on CPU: two `gloo` ranks row-shard one image and run the global-histogram
operators with the single all-reduce of the design.  The device steps
(binning, LUT application) need a GPU, so here each rank bins and applies with NumPy —
test-side only — while everything that is multi-GPU *logic* is the product's:
`shard_range`, `all_reduce_histogram` (torch.distributed) and the C LUT builders of
libmagickhip.so.  The merged result must equal the single-image oracle bit for bit."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, hdri, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import imagemagick_amd as im
    from imagemagick_amd import distributed as D
    from oracle import restate as R
    vectors = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
    px = vectors[("hdri" if hdri else "q16") + "_smooth_in"]
    rows, cols, ch = px.shape
    begin, end = D.shard_range(rows, rank, world)
    band = px[begin:end]
    quantum = 1 if hdri else 0

    def histogram_of(band_pixels):          # stands in for the HIP histogram kernel of this rank
        inten = R.pixel_intensity(band_pixels)
        idx = R.scale_quantum_to_map(R.clamp_to_quantum(inten, hdri), hdri)
        h = np.zeros((65536, ch), dtype=np.uint64)
        for c in range(ch):
            h[:, c] = np.bincount(idx.ravel(), minlength=65536)
        return h

    def apply(band_pixels, lut, mask):      # stands in for MagickHipApplyLUT
        own = R.scale_quantum_to_map(band_pixels, hdri)
        out = band_pixels.copy()
        for c in range(ch):
            if (mask >> c) & 1:
                out[:, :, c] = R.clamp_to_quantum(lut[own[:, :, c], c], hdri)
        return out

    hist = D.all_reduce_histogram(histogram_of(band))
    assert int(hist[:, 0].sum()) == rows * cols          # every rank holds the global table
    lut, mask = im.equalize_lut(hist, quantum)
    np.save(os.path.join(outdir, "eq_%d.npy" % rank), apply(band, lut, mask))
    n = rows * cols
    lut, mask = im.contrast_stretch_lut(hist, cols, rows, 0.02 * n, n - 0.01 * n, quantum)
    np.save(os.path.join(outdir, "cs_%d.npy" % rank), apply(band, lut, mask))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("hdri", [False, True])
def test_row_sharded_histogram_operators_two_ranks(hdri):
    import torch.multiprocessing as mp
    world = 2
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(world, _free_port(), hdri, outdir), nprocs=world, join=True)
        vectors = np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))
        tag = "hdri" if hdri else "q16"
        for key, name in (("eq", "_smooth_equalize"), ("cs", "_smooth_cstretch")):
            merged = np.concatenate([np.load(os.path.join(outdir, "%s_%d.npy" % (key, r)))
                                     for r in range(world)], axis=0)
            assert np.array_equal(merged, vectors[tag + name]), key


def test_shard_range_and_halo():
    from imagemagick_amd import distributed as D
    for items, world in ((512, 8), (10, 4), (3, 8), (8192, 3)):
        spans = [D.shard_range(items, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == items
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [e - b for b, e in spans]
        assert max(sizes) - min(sizes) <= 1
    own, held = D.band_with_halo(16384, 3, 8, 15, 15)        # C5: Disk:15 on 8 GPUs
    assert own == (6144, 8192) and held == (6129, 8207)
    own, held = D.band_with_halo(8192, 0, 8, 39, 39)         # sigma=10 column pass, top band
    assert own == (0, 1024) and held == (0, 1063)


@pytest.mark.gpu
def test_equalize_band_single_rank_on_gpu(im, refmod):
    """The same functions on the GPU path (world_size 1, gloo): device histogram and apply."""
    import torch.distributed as dist
    from conftest import make_pixels, to_device
    from imagemagick_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        px = make_pixels(64, 80, 4, np.uint16, kind="smooth")
        dev = im.Image(to_device(px))
        D.equalize_band(dev)
        assert np.array_equal(dev.numpy(), refmod.RefImage(px).equalize().numpy())
        dev = im.Image(to_device(px))
        n = 64 * 80
        D.contrast_stretch_band(dev, 80, 64, 0.02 * n, n - 0.01 * n)
        assert np.array_equal(dev.numpy(), refmod.RefImage(px).contrast_stretch(0.02 * n, n - 0.01 * n).numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_row_sharded_stencils_match_whole_image(im, refmod, world):
    """BASELINE config C5 on N GPUs, rehearsed on one: every rank's band (owned rows + halo,
    `distributed.run_on_band`) goes through the GPU separately and the stitched result must be
    the whole-image result bit for bit — blur (EXACT), Dilate with a disk, a compound method and
    UnsharpMask; odd row counts so that bands are uneven."""
    from conftest import make_pixels
    from imagemagick_amd import distributed as D
    px = make_pixels(173, 96, 4, np.uint16)
    cases = [
        ("blur 0x3", D.stencil_reach("blur", sigma=3.0), lambda i: im.blur_image(i, 0.0, 3.0),
         lambda r: r.blur(0.0, 3.0)),
        ("dilate disk:5", D.stencil_reach("morphology", method="Dilate", kernel="Disk:5"),
         lambda i: im.morphology_image(i, "Dilate", 1, "Disk:5"), lambda r: r.morphology("Dilate", 1, "Disk:5")),
        ("smooth octagon:2", D.stencil_reach("morphology", method="Smooth", kernel="Octagon:2"),
         lambda i: im.morphology_image(i, "Smooth", 1, "Octagon:2"), lambda r: r.morphology("Smooth", 1, "Octagon:2")),
        ("unsharp 0x2", D.stencil_reach("unsharp", sigma=2.0), lambda i: im.unsharp_mask_image(i, 0.0, 2.0, 1.0, 0.02),
         lambda r: r.unsharp(0.0, 2.0, 1.0, 0.02)),
    ]
    for name, reach, gpu_op, ref_op in cases:
        merged = np.empty_like(px)
        for rank in range(world):
            begin, end, rows = D.run_on_band(px, rank, world, reach, gpu_op)
            merged[begin:end] = rows
        assert np.array_equal(merged, ref_op(refmod.RefImage(px)).numpy()), "%s over %d bands" % (name, world)
