"""MagickHipBatchImages / MagickHipShardedImage (SURVEY section 8e from plain C): operator
chains over independent images and over the row bands of one image.  A single GPU rehearses
several logical devices (logical d runs on physical d mod MhDeviceCount()): worker threads,
streams, halo exchange with hipMemcpyPeerAsync and the table reduction are the real code paths,
only the peers happen to be the same GPU."""
import numpy as np
import pytest

from conftest import make_pixels, to_device, assert_parity

pytestmark = pytest.mark.gpu

Q16, HDRI = np.uint16, np.float32


def test_batch_lab_contrast_stretch_host_images(im, refmod):
    """BASELINE config C4 in miniature: independent host images, sRGB->Lab + ContrastStretch,
    two logical devices x two streams; every result equals the reference's."""
    rows, cols, count = 150, 170, 7
    pixels = [make_pixels(rows, cols, 4, Q16, seed=300 + i) for i in range(count)]
    images = [im.Image(p.copy()) for p in pixels]
    n = rows * cols
    report = im.batch_images([("colorspace", "Lab"), ("contraststretch", 0.02 * n, n - 0.01 * n)], images,
                             devices=2, streams_per_device=2)
    assert report["devices"] == 2 and report["workers"] == 4
    assert sum(report["images_per_device"]) == count
    for p, image in zip(pixels, images):
        want = refmod.RefImage(p).colorspace("Lab").contrast_stretch(0.02 * n, n - 0.01 * n).numpy()
        assert image.colorspace == "lab"
        assert_parity(image.numpy(), want, True, "batch Lab + ContrastStretch")


@pytest.mark.parametrize("memory", ["host", "device"])
def test_batch_new_image_chain(im, refmod, memory):
    """A chain of new-image operators (blur, resize) with result descriptors of the final
    geometry; inputs stay untouched."""
    rows, cols, count = 120, 96, 5
    pixels = [make_pixels(rows, cols, 4, Q16, seed=400 + i) for i in range(count)]
    if memory == "host":
        images = [im.Image(p.copy()) for p in pixels]
        results = [im.Image(np.zeros((75, 131, 4), dtype=np.uint16)) for _ in range(count)]
    else:
        images = [im.Image(to_device(p)) for p in pixels]
        results = [images[0].like(rows=75, columns=131) for _ in range(count)]
    report = im.batch_images([("blur", 0.0, 2.0), ("resize", 131, 75, "Lanczos")], images, results,
                             devices=3, streams_per_device=1)
    assert sum(report["images_per_device"]) == count
    for p, image, result in zip(pixels, images, results):
        assert np.array_equal(image.numpy(), p)
        want = refmod.RefImage(p).blur(0.0, 2.0).resize(131, 75, "Lanczos").numpy()
        assert_parity(result.numpy(), want, True, "batch blur + resize (%s)" % memory)


def test_batch_reports_the_first_failure(im):
    images = [im.Image(make_pixels(40, 50, 4, Q16, seed=i)) for i in range(3)]
    with pytest.raises(im.MagickHipError):
        im.batch_images([("morphology", "Dilate", 1, "NoSuchKernel:3")], images)


@pytest.mark.parametrize("devices", [1, 2, 3, 5])
@pytest.mark.parametrize("dtype", [Q16, HDRI])
def test_sharded_stencil_chain_matches_reference(im, refmod, devices, dtype):
    """One image in row bands: Dilate Disk:5 then UnsharpMask (BASELINE config C5's operators) —
    halo rows are exchanged between the bands before the second operator.  EXACT precision:
    bit-identical to the reference whatever the number of bands."""
    px = make_pixels(230, 150, 4, dtype, seed=77)
    image = im.Image(px.copy())
    result, report = im.sharded_image([("morphology", "Dilate", 1, "Disk:5"), ("unsharpmask", 0.0, 2.0, 1.0, 0.02)],
                                      image, devices=devices)
    want = refmod.RefImage(px).morphology("Dilate", 1, "Disk:5").unsharp(0.0, 2.0, 1.0, 0.02).numpy()
    assert report["devices"] == devices
    assert report["halo_exchanges"] == 2 * (devices - 1)
    assert_parity(result.numpy(), want, True, "sharded dilate + unsharp, %d bands" % devices)


@pytest.mark.parametrize("devices", [2, 4])
@pytest.mark.parametrize("operator", ["equalize", "contraststretch"])
@pytest.mark.parametrize("dtype", [Q16, HDRI])
def test_sharded_histogram_operators(im, refmod, devices, operator, dtype):
    """Global-histogram operators on a row-sharded image: local tables over the owned rows, one
    all-reduce of the 65536 x channels table, the identical LUT on every band."""
    px = make_pixels(300, 210, 4, dtype, seed=91, kind="smooth")
    n = px.shape[0] * px.shape[1]
    chain = [("equalize",)] if operator == "equalize" else [("contraststretch", 0.03 * n, n - 0.02 * n)]
    result, report = im.sharded_image(chain, im.Image(px.copy()), devices=devices)
    ref = refmod.RefImage(px)
    want = (ref.equalize() if operator == "equalize" else ref.contrast_stretch(0.03 * n, n - 0.02 * n)).numpy()
    assert report["devices"] == devices
    assert_parity(result.numpy(), want, True, "sharded %s, %d bands" % (operator, devices))


def test_sharded_chain_with_colourspace_and_blur_device_memory(im, refmod):
    """Device-resident source and result; pointwise operator between two stencils."""
    px = make_pixels(260, 140, 4, Q16, seed=5)
    image = im.Image(to_device(px), colorspace="sRGB")
    result, report = im.sharded_image([("blur", 0.0, 3.0), ("colorspace", "RGB"), ("blur", 0.0, 1.5)], image,
                                      devices=3)
    want = refmod.RefImage(px).blur(0.0, 3.0).colorspace("RGB").blur(0.0, 1.5).numpy()
    assert result.colorspace == "rgb"
    assert_parity(result.numpy(), want, True, "sharded blur, sRGB->RGB, blur")


def test_sharded_fast_blur_within_one_level(im, refmod):
    """FAST precision (the fused matrix-core blur) on bands: within +-1 level of the reference."""
    px = make_pixels(400, 330, 4, Q16, seed=15)
    im.set_precision(im.PRECISION_FAST)
    try:
        result, _ = im.sharded_image([("blur", 0.0, 6.0)], im.Image(px.copy()), devices=3)
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert_parity(result.numpy(), refmod.RefImage(px).blur(0.0, 6.0).numpy(), False, "sharded FAST blur")


# ---------------------------------------------------------------- two physical GPUs
# The tests above rehearse several logical devices on one GPU (the table reduction then runs
# through peer copies + an add kernel, the halo copies stay on the device).  These run the same
# entry points over two PHYSICAL devices — RCCL all-reduce, real hipMemcpyPeerAsync, kernels on
# the GPU that owns the memory — and skip cleanly on a one-GPU box.
def _two_gpus(im):
    lib = im.load()
    if lib.MhDeviceCount() < 2:
        pytest.skip("needs two physical GPUs (MhDeviceCount() = %d)" % lib.MhDeviceCount())


@pytest.mark.parametrize("operator", ["equalize", "contraststretch"])
def test_two_gpus_sharded_histogram_uses_rccl(im, refmod, operator):
    _two_gpus(im)
    px = make_pixels(512, 384, 4, Q16, seed=191, kind="smooth")
    n = px.shape[0] * px.shape[1]
    chain = [("equalize",)] if operator == "equalize" else [("contraststretch", 0.03 * n, n - 0.02 * n)]
    for _ in range(2):                       # the second call reuses the cached communicator
        result, report = im.sharded_image(chain, im.Image(px.copy()), devices=2)
        assert report["devices"] == 2 and report["used_rccl"] == 1, report
    ref = refmod.RefImage(px)
    want = (ref.equalize() if operator == "equalize" else ref.contrast_stretch(0.03 * n, n - 0.02 * n)).numpy()
    assert_parity(result.numpy(), want, True, "two GPUs, sharded %s" % operator)


def test_two_gpus_sharded_stencil_chain_peer_halos(im, refmod):
    _two_gpus(im)
    px = make_pixels(460, 300, 4, Q16, seed=177)
    result, report = im.sharded_image([("morphology", "Dilate", 1, "Disk:5"), ("unsharpmask", 0.0, 2.0, 1.0, 0.02)],
                                      im.Image(px.copy()), devices=2)
    want = refmod.RefImage(px).morphology("Dilate", 1, "Disk:5").unsharp(0.0, 2.0, 1.0, 0.02).numpy()
    assert report["devices"] == 2 and report["halo_exchanges"] == 2
    assert_parity(result.numpy(), want, True, "two GPUs, sharded dilate + unsharp")


def test_two_gpus_batch_in_place_images_stay_on_their_gpu(im, refmod):
    """Device-resident images on BOTH GPUs, processed in place (results = None): every chain must
    run on the GPU that owns the pixels, whichever worker picks the image up."""
    import torch
    _two_gpus(im)
    rows, cols, count = 150, 170, 8
    pixels = [make_pixels(rows, cols, 4, Q16, seed=500 + i) for i in range(count)]
    tensors = [torch.from_numpy(p.view(np.int16)).to("cuda:%d" % (i % 2)).view(torch.uint16) for i, p in enumerate(pixels)]
    images = [im.Image(t) for t in tensors]
    n = rows * cols
    report = im.batch_images([("colorspace", "Lab"), ("contraststretch", 0.02 * n, n - 0.01 * n)], images,
                             devices=2, streams_per_device=2)
    assert sum(report["images_per_device"]) == count
    for i, (p, image) in enumerate(zip(pixels, images)):
        assert image.pixels.device.index == i % 2
        want = refmod.RefImage(p).colorspace("Lab").contrast_stretch(0.02 * n, n - 0.01 * n).numpy()
        assert_parity(image.numpy(), want, True, "two GPUs, in-place batch, image %d" % i)


def test_two_gpus_batch_copies_across_gpus(im, refmod):
    """Device-resident inputs on GPU 0 with result descriptors on GPU 1: the working copies and
    the delivery cross the GPUs with peer copies."""
    import torch
    _two_gpus(im)
    rows, cols, count = 96, 130, 4
    pixels = [make_pixels(rows, cols, 4, Q16, seed=600 + i) for i in range(count)]
    images = [im.Image(torch.from_numpy(p.view(np.int16)).to("cuda:0").view(torch.uint16)) for p in pixels]
    results = [im.Image(torch.zeros((rows, cols, 4), dtype=torch.int16, device="cuda:1").view(torch.uint16))
               for _ in range(count)]
    im.batch_images([("blur", 0.0, 2.0)], images, results, devices=2, streams_per_device=1)
    for p, result in zip(pixels, results):
        assert_parity(result.numpy(), refmod.RefImage(p).blur(0.0, 2.0).numpy(), True, "two GPUs, cross-GPU batch")


def test_batch_waits_for_the_callers_stream(im, refmod):
    """A device image whose pixels are still being produced on the caller's stream when
    MagickHipBatchImages is called (a long fill kernel, then the copy-in): the batch's worker
    streams must order themselves behind that stream (wait_for_caller, batch.cpp)."""
    import torch
    px = make_pixels(400, 600, 4, Q16, seed=71)
    source = to_device(px)
    stream = torch.cuda.Stream()
    target = torch.zeros_like(source)
    burn = torch.empty((64 << 20,), dtype=torch.float32, device="cuda")
    with torch.cuda.stream(stream):
        for _ in range(6):                   # tens of milliseconds of work ahead of the copy
            burn.normal_()
        target.copy_(source)
    image = im.Image(target, stream=stream.cuda_stream)
    result = image.like()
    im.batch_images([("blur", 0.0, 1.5)], [image], [result], devices=1, streams_per_device=1)
    assert_parity(result.numpy(), refmod.RefImage(px).blur(0.0, 1.5).numpy(), True, "batch behind the caller's stream")
