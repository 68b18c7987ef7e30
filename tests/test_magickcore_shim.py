"""Drop-in check on the GPU box: MagickCore itself (compiled from the reference with
shim/accelerate_hip.c + shim/opencl_hip.c in place of accelerate.c / opencl.c) runs
BlurImage / ResizeImage / EqualizeImage through its own unchanged call sites
(effect.c:783-787, resize.c:3818-3826, enhance.c:2072-2075), lands in
libmagickhip.so, and returns what the pure-CPU MagickCore returns."""
import ctypes
import os

import numpy as np
import pytest

from conftest import make_pixels, assert_parity

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def shim(refmod, im):
    if not (os.path.exists(refmod.shim_lib_path(False)) and os.path.exists(refmod.shim_lib_path(True))):
        pytest.skip("shim/_build is not built (make -C shim, build container only)")
    os.environ["MAGICK_HIP_LIBRARY"] = os.path.join(ROOT, "imagemagick_amd", "lib", "libmagickhip.so")
    return refmod


def accelerated_calls(refmod, hdri):
    lib = refmod._load(hdri, True)
    lib.GetMagickHipAcceleratedCalls.restype = ctypes.c_size_t
    return lib.GetMagickHipAcceleratedCalls()


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_blur_resize_equalize_through_magickcore(shim, dtype):
    hdri = dtype == np.float32
    px = make_pixels(70, 90, 4, dtype)
    before = accelerated_calls(shim, hdri)
    gpu = shim.RefImage(px, shim=True)
    cpu = shim.RefImage(px)
    assert_parity(gpu.blur(0.0, 3.0).numpy(), cpu.blur(0.0, 3.0).numpy(), True, "BlurImage via MagickCore")
    assert accelerated_calls(shim, hdri) == before + 1, "BlurImage did not take the accelerated path"
    assert_parity(gpu.resize(200, 131, "Lanczos").numpy(), cpu.resize(200, 131, "Lanczos").numpy(), True,
                  "ResizeImage via MagickCore")
    assert_parity(gpu.resize(41, 33, "Mitchell").numpy(), cpu.resize(41, 33, "Mitchell").numpy(), True,
                  "ResizeImage (Mitchell) via MagickCore")
    assert accelerated_calls(shim, hdri) == before + 3
    smooth = make_pixels(64, 64, 3, dtype, kind="smooth")
    g, c = shim.RefImage(smooth, shim=True), shim.RefImage(smooth)
    assert_parity(g.equalize().numpy(), c.equalize().numpy(), True, "EqualizeImage via MagickCore")
    assert accelerated_calls(shim, hdri) == before + 4


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_large_frame_moves_through_the_staged_transfers(shim, dtype):
    """A frame of several 4 MiB pieces (1100x1300 RGBA: 11 MB as Q16, 23 MB as float) goes up
    and comes down through MhUpload / MhDownload's threaded staging; ragged last piece included."""
    hdri = dtype == np.float32
    px = make_pixels(1100, 1301, 4, dtype, seed=3)
    before = accelerated_calls(shim, hdri)
    gpu = shim.RefImage(px, shim=True)
    cpu = shim.RefImage(px)
    assert_parity(gpu.blur(0.0, 1.5).numpy(), cpu.blur(0.0, 1.5).numpy(), True, "BlurImage, large frame")
    assert accelerated_calls(shim, hdri) == before + 1


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_pixel_caches_are_page_locked_once_acceleration_is_switched_on(shim, dtype):
    """SetOpenCLEnabled(MagickTrue) installs the library's page-locked allocator behind
    AcquireAlignedMemory (SetMagickAlignedMemoryMethods, memory.c:1541): the pixel cache of an image
    created afterwards (4 MiB and more) is hipHostMalloc memory, moves with one DMA transfer per
    direction, and is given back when the image is destroyed; smaller caches and blocks that predate
    the switch keep using the C allocator.  Same pixels as the CPU MagickCore."""
    import gc
    hdri = dtype == np.float32
    lib = shim._load(hdri, True)
    lib.GetMagickHipPinnedCacheExtent.restype = ctypes.c_size_t
    lib.SetOpenCLEnabled.argtypes = [ctypes.c_int]
    old = shim.RefImage(make_pixels(600, 700, 4, dtype, seed=5), shim=True)      # predates the switch (or not: both fine)
    assert lib.SetOpenCLEnabled(1) == 1
    gc.collect()
    base = lib.GetMagickHipPinnedCacheExtent()
    px = make_pixels(1100, 1301, 4, dtype, seed=4)
    before = accelerated_calls(shim, hdri)
    gpu = shim.RefImage(px, shim=True)
    held = lib.GetMagickHipPinnedCacheExtent()
    assert held >= base + px.nbytes, (base, held, px.nbytes)
    small = shim.RefImage(make_pixels(40, 50, 4, dtype), shim=True)
    assert lib.GetMagickHipPinnedCacheExtent() == held, "a 16 KB cache must not be page-locked"
    cpu = shim.RefImage(px)
    assert_parity(gpu.blur(0.0, 1.5).numpy(), cpu.blur(0.0, 1.5).numpy(), True, "BlurImage, page-locked cache")
    assert accelerated_calls(shim, hdri) == before + 1
    assert_parity(old.blur(0.0, 1.5).numpy(), shim.RefImage(make_pixels(600, 700, 4, dtype, seed=5)).blur(0.0, 1.5).numpy(),
                  True, "BlurImage, cache from before the switch")
    del gpu, small, old
    gc.collect()
    assert lib.GetMagickHipPinnedCacheExtent() <= held - px.nbytes


def test_fast_precision_through_magickcore(shim, im):
    """MAGICK_HIP_PRECISION=fast (here: MhSetPrecision on the library instance the shim loaded):
    MagickCore's own BlurImage, GaussianBlurImage (a 2-D kernel, separated by the library) and
    UnsharpMaskImage (fused column pass) stay within the FAST contract of the CPU MagickCore."""
    px = make_pixels(96, 120, 4, np.uint16, seed=21)
    before = accelerated_calls(shim, False)
    gpu = shim.RefImage(px, shim=True)
    cpu = shim.RefImage(px)
    im.set_precision(im.PRECISION_FAST)
    try:
        blur = gpu.blur(0.0, 3.0).numpy()
        gauss = gpu.gaussian_blur(0.0, 2.0).numpy()
        unsharp = gpu.unsharp(0.0, 2.0, 1.0, 0.02).numpy()
    finally:
        im.set_precision(im.PRECISION_EXACT)
    assert accelerated_calls(shim, False) >= before + 3
    assert_parity(blur, cpu.blur(0.0, 3.0).numpy(), False, "FAST BlurImage via MagickCore")
    assert_parity(gauss, cpu.gaussian_blur(0.0, 2.0).numpy(), False, "FAST GaussianBlurImage via MagickCore")
    # FAST UnsharpMaskImage runs on the reference's own blur (both passes exact): bit-identical on this layout
    assert_parity(unsharp, cpu.unsharp(0.0, 2.0, 1.0, 0.02).numpy(), True, "FAST UnsharpMaskImage via MagickCore")


def test_default_mode_through_magickcore(shim):
    """What an UNCHANGED caller gets: a fresh process with no MAGICK_HIP_PRECISION in its environment and no
    MhSetPrecision call (tests/shim_default_mode_child.py) runs MagickCore's own BlurImage, GaussianBlurImage,
    UnsharpMaskImage, ResizeImage x4 and /4, ConvolveImage, MorphologyImage(Convolve), TransformImageColorspace(Lab)
    + ContrastStretchImage on Q16 and float frames through the binding; every result against the CPU MagickCore
    to the documented contract of the default (FAST) mode."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if not k.startswith("MAGICKHIP_") and k != "MAGICK_HIP_PRECISION"}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shim_default_mode_child.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["precision"] == 1, "the library's default is FAST"
    assert r["accelerated_calls_q16"] >= 10 and r["accelerated_calls_float"] >= 4, r
    for name in ("blur", "gaussian_blur", "resize_x4", "resize_div4", "convolve", "convolve_disk", "lab"):
        assert r[name] <= 1, (name, r)                   # within one Quantum level
    # bit-identical in the default mode too: UnsharpMaskImage (the reference's own blur feeds the epilogue), a
    # blur whose outer taps are tiny on an alpha-weighted frame (the exact kernels), the table operators
    for name in ("unsharp", "unsharp_radius", "blur_radius", "contrast_stretch_of_that_lab_frame"):
        assert r[name] == 0, (name, r)
    for name in ("float_resize_x4", "float_resize_div4"):
        assert r[name] <= 1, (name, r)                   # one float ULP
    for name in ("float_blur", "float_unsharp"):
        assert r[name] == 0, (name, r)                   # float Quantum blurs are bit-identical in either mode


def test_gate_falls_back_to_cpu(shim):
    """An image the gate rejects (a colourspace the backend does not take) silently runs the
    CPU path — the reference's NULL-return convention."""
    px = make_pixels(40, 50, 4, np.uint16)
    before = accelerated_calls(shim, False)
    gpu = shim.RefImage(px, "Lab", shim=True)
    cpu = shim.RefImage(px, "Lab")
    assert_parity(gpu.blur(0.0, 2.0).numpy(), cpu.blur(0.0, 2.0).numpy(), True, "Lab blur (CPU fallback)")
    assert accelerated_calls(shim, False) == before


def transfers(refmod, hdri):
    lib = refmod._load(hdri, True)
    up, down = ctypes.c_size_t(0), ctypes.c_size_t(0)
    lib.GetMagickHipTransfers(ctypes.byref(up), ctypes.byref(down))
    return up.value, down.value


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_chained_operators_stay_on_the_device(shim, dtype):
    """BlurImage -> ResizeImage -> EqualizeImage through MagickCore: the image is uploaded
    once, intermediate results are never downloaded, and the host pixel cache is brought up to
    date lazily when the CPU finally reads it (the reference's CopyOpenCLBuffer protocol,
    cache.c:5341-5353) — and the pixels still equal the CPU MagickCore's."""
    hdri = dtype == np.float32
    px = make_pixels(80, 100, 4, dtype, kind="smooth")
    up0, down0 = transfers(shim, hdri)
    g = shim.RefImage(px, shim=True)
    blurred = g.blur(0.0, 2.5)
    resized = blurred.resize(150, 120, "Lanczos")
    resized.equalize()
    assert transfers(shim, hdri) == (up0 + 1, down0), "chain must not move pixels over PCIe"
    got = resized.numpy()                       # first CPU access: the one download
    assert transfers(shim, hdri) == (up0 + 1, down0 + 1)
    c = shim.RefImage(px).blur(0.0, 2.5).resize(150, 120, "Lanczos")
    c.equalize()
    assert_parity(got, c.numpy(), True, "blur -> resize -> equalize chain")
    # a CPU operator in the middle of a chain sees current pixels (download), then the GPU
    # path re-uploads: Lab is rejected by the gate, so this blur runs on the CPU
    mixed = g.blur(0.0, 1.5)
    mixed.colorspace("Lab")
    ref_mixed = shim.RefImage(px).blur(0.0, 1.5)
    ref_mixed.colorspace("Lab")
    assert_parity(mixed.blur(0.0, 2.0).numpy(), ref_mixed.blur(0.0, 2.0).numpy(), True, "GPU -> CPU -> CPU")


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_grayscale_and_function_through_magickcore(shim, dtype):
    """AccelerateGrayscaleImage / AccelerateFunctionImage have live call sites in the reference
    (enhance.c:2502-2510, statistic.c:1101-1105)."""
    hdri = dtype == np.float32
    px = make_pixels(45, 60, 4, dtype)
    before = accelerated_calls(shim, hdri)
    g, c = shim.RefImage(px, shim=True), shim.RefImage(px)
    assert_parity(g.function("Polynomial", (0.5, -0.2, 0.6)).numpy(), c.function("Polynomial", (0.5, -0.2, 0.6)).numpy(),
                  True, "FunctionImage via MagickCore")
    assert accelerated_calls(shim, hdri) == before + 1
    assert_parity(g.grayscale("Rec709Luma").numpy(), c.grayscale("Rec709Luma").numpy(), True,
                  "GrayscaleImage via MagickCore")
    assert accelerated_calls(shim, hdri) == before + 2
    assert g.info()["colorspace"].lower() == "gray" and g.info()["channels"] == c.info()["channels"]


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_morphology_hook_covers_convolve_callers(shim, dtype):
    """The hook shim/patch_hooks.py adds at the top of MorphologyApply puts every
    caller of MorphologyImage / ConvolveImage on the device: -morphology, GaussianBlurImage,
    SharpenImage, EdgeImage, EmbossImage (+ its EqualizeImage), and the re-enabled
    UnsharpMaskImage stanza (effect.c:4287-4294)."""
    hdri = dtype == np.float32
    px = make_pixels(52, 64, 4, dtype, kind="smooth")
    g, c = shim.RefImage(px, shim=True), shim.RefImage(px)
    steps = [
        ("EdgeOut Disk:2.5", lambda i: i.morphology("EdgeOut", 1, "Disk:2.5"), 1),
        ("Smooth Octagon:2", lambda i: i.morphology("Smooth", 1, "Octagon:2"), 1),
        ("GaussianBlur 0x1.5", lambda i: i.gaussian_blur(0.0, 1.5), 1),
        ("Sharpen 0x1", lambda i: i.sharpen(0.0, 1.0), 1),
        ("Edge 1", lambda i: i.edge(1.0), 1),
        ("Emboss 0x1", lambda i: i.emboss(0.0, 1.0), 2),
        ("UnsharpMask 0x2+1+0.02", lambda i: i.unsharp(0.0, 2.0, 1.0, 0.02), 1),
        ("MotionBlur 0x3+40", lambda i: i.motion_blur(0.0, 3.0, 40.0), 1),
        ("RotationalBlur 15", lambda i: i.rotational_blur(15.0), 1),
        ("LocalContrast 80x50", lambda i: i.local_contrast(80.0, 50.0), 1),
        ("Despeckle", lambda i: i.despeckle(), 1),
        ("WaveletDenoise 4000x0.3", lambda i: i.wavelet_denoise(4000.0, 0.3), 1),
    ]
    for name, op, calls in steps:
        before = accelerated_calls(shim, hdri)
        got = op(g).numpy()
        assert accelerated_calls(shim, hdri) == before + calls, name + " did not take the accelerated path"
        assert_parity(got, op(c).numpy(), True, name + " via MagickCore")


def test_hit_and_miss_list_through_magickcore(shim):
    px = make_pixels(40, 48, 3, np.uint16, kind="binary")
    before = accelerated_calls(shim, False)
    g, c = shim.RefImage(px, shim=True), shim.RefImage(px)
    assert_parity(g.morphology("HitAndMiss", 1, "LineEnds").numpy(), c.morphology("HitAndMiss", 1, "LineEnds").numpy(),
                  True, "HitAndMiss LineEnds via MagickCore")
    assert_parity(g.morphology("Thinning", -1, "Skeleton").numpy(), c.morphology("Thinning", -1, "Skeleton").numpy(),
                  True, "Thinning Skeleton via MagickCore")
    assert accelerated_calls(shim, False) == before + 2
    # Distance is sequential: the library declines, MagickCore runs its CPU code
    assert_parity(g.morphology("Distance", 1, "Euclidean:1").numpy(), c.morphology("Distance", 1, "Euclidean:1").numpy(),
                  True, "Distance (CPU fallback)")
    assert accelerated_calls(shim, False) == before + 2


@pytest.mark.parametrize("compose,accelerated", [("Plus", True), ("Darken", True), ("Screen", True), ("MinusSrc", True), ("Overlay", False)])
def test_morphology_compose_through_magickcore(shim, compose, accelerated):
    """`-define morphology:compose=` (morphology.c:4206-4215) reaches the hook as MorphologyApply's
    compose argument: Plus / Darken / Multiply / Screen (round 4) compose on the device, any other
    operator is MagickCore's CPU path."""
    px = make_pixels(44, 60, 4, np.uint16, kind="smooth")
    g, c = shim.RefImage(px, shim=True), shim.RefImage(px)
    before = accelerated_calls(shim, False)
    for image in (g, c):
        image.set_artifact("morphology:compose", compose)
    got = g.morphology("Convolve", 1, "Sobel:>").numpy()
    want = c.morphology("Convolve", 1, "Sobel:>").numpy()
    assert accelerated_calls(shim, False) == before + (1 if accelerated else 0)
    assert_parity(got, want, True, "Convolve Sobel:> compose %s via MagickCore" % compose)


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_colorspace_and_contrast_stretch_chain(shim, dtype):
    """BASELINE config C4 through MagickCore: TransformImageColorspace(Lab) (new hook,
    colorspace.c:1751) then ContrastStretchImage (hook added to the caller-less
    AccelerateContrastStretchImage): one upload, no download until the CPU reads the result."""
    hdri = dtype == np.float32
    px = make_pixels(64, 96, 4, dtype, kind="smooth")
    n = 64 * 96
    before = accelerated_calls(shim, hdri)
    up0, down0 = transfers(shim, hdri)
    g, c = shim.RefImage(px, shim=True), shim.RefImage(px)
    for image in (g, c):
        image.colorspace("Lab")
        image.contrast_stretch(0.02 * n, n - 0.01 * n)
    assert accelerated_calls(shim, hdri) == before + 2
    assert transfers(shim, hdri) == (up0 + 1, down0)
    assert g.info()["colorspace"] == c.info()["colorspace"] and "lab" in g.info()["colorspace"].lower()
    assert_parity(g.numpy(), c.numpy(), True, "sRGB -> Lab + ContrastStretch via MagickCore", max_ulp=1)
    assert transfers(shim, hdri) == (up0 + 1, down0 + 1)
    # a tour of the accelerated colourspaces, two-step transforms included (X -> sRGB -> Y)
    g, c = shim.RefImage(px, shim=True), shim.RefImage(px)
    for target in ("XYZ", "Lab", "RGB", "sRGB", "Lab", "XYZ", "sRGB"):
        g.colorspace(target)
        c.colorspace(target)
        assert g.info()["colorspace"] == c.info()["colorspace"], target
        assert_parity(g.numpy(), c.numpy(), True, "-> %s via MagickCore" % target, max_ulp=1)
    # ... and of the table-driven ones (round 4): accelerated in both directions, except YCC -> sRGB,
    # which the library declines (YCCMap) — MagickCore's CPU path runs and the bits are the same
    g, c = shim.RefImage(px, shim=True), shim.RefImage(px)
    before = accelerated_calls(shim, hdri)
    for target in ("OHTA", "Rec709YCbCr", "sRGB", "Rec601YCbCr", "scRGB", "YCC", "sRGB"):
        g.colorspace(target)
        c.colorspace(target)
        assert g.info()["colorspace"] == c.info()["colorspace"], target
        assert_parity(g.numpy(), c.numpy(), True, "-> %s via MagickCore" % target, max_ulp=1)
    assert accelerated_calls(shim, hdri) == before + 6


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
def test_contrast_and_modulate_through_magickcore(shim, dtype):
    """AccelerateContrastImage / AccelerateModulateImage have live call sites in the reference
    (enhance.c:1412-1415, :3770-3774); ModulateImage's other colour models (here HWB,
    enhance.c:3826-3890) are accelerated too since round 2."""
    hdri = dtype == np.float32
    px = make_pixels(50, 66, 4, dtype)
    before = accelerated_calls(shim, hdri)
    g, c = shim.RefImage(px, shim=True), shim.RefImage(px)
    assert_parity(g.contrast(True).numpy(), c.contrast(True).numpy(), True, "ContrastImage via MagickCore", max_ulp=1)
    assert accelerated_calls(shim, hdri) == before + 1
    assert_parity(g.modulate(115.0, 85.0, 140.0).numpy(), c.modulate(115.0, 85.0, 140.0).numpy(), True,
                  "ModulateImage via MagickCore", max_ulp=1)
    assert accelerated_calls(shim, hdri) == before + 2
    assert_parity(g.modulate(90.0, 120.0, 70.0, "HSB").numpy(), c.modulate(90.0, 120.0, 70.0, "HSB").numpy(), True,
                  "ModulateImage HSB via MagickCore", max_ulp=1)
    assert accelerated_calls(shim, hdri) == before + 3
    assert_parity(g.modulate(90.0, 120.0, 70.0, "HWB").numpy(), c.modulate(90.0, 120.0, 70.0, "HWB").numpy(), True,
                  "ModulateImage HWB via MagickCore", max_ulp=1)
    assert accelerated_calls(shim, hdri) == before + 4


def test_accelerate_events_are_logged(shim, tmp_path):
    """`-debug accelerate` (LogMagickEvent(AccelerateEvent, ...), MagickCore/log.h:37-57): the shim
    says which path an operator took — accepted on the HIP backend, or declined and why."""
    import sys
    lib = shim._load(False, True)
    if not hasattr(lib, "SetLogEventMask"):
        pytest.skip("the shim's MagickCore does not export SetLogEventMask")
    lib.SetLogEventMask.restype = ctypes.c_int
    lib.SetLogEventMask.argtypes = [ctypes.c_char_p]
    px = make_pixels(48, 60, 4, np.uint16)
    capture = tmp_path / "accelerate.log"
    sys.stderr.flush()
    saved = os.dup(2)
    fd = os.open(str(capture), os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    try:
        os.dup2(fd, 2)
        lib.SetLogEventMask(b"Accelerate")
        image = shim.RefImage(px, shim=True)
        image.blur(0.0, 2.0)                                           # accepted
        image.set_artifact("convolve:scale", "2").blur(0.0, 2.0)       # declined: an artifact is set
    finally:
        lib.SetLogEventMask(b"None")
        os.dup2(saved, 2)
        os.close(fd)
        os.close(saved)
    text = capture.read_text(errors="replace")
    assert "AccelerateBlurImage: accelerated on the HIP backend" in text, text[-2000:]
    assert "AccelerateBlurImage: not accelerated, the CPU path runs" in text, text[-2000:]
    assert "artifact is set" in text


@pytest.mark.parametrize("dtype", [np.uint16, np.float32])
@pytest.mark.parametrize("target", ["Gray", "LinearGray"])
def test_srgb_to_gray_through_magickcore(shim, dtype, target):
    """TransformImageColorspace(GRAY / LinearGRAY) (colorspace.c:843-957): the gray values are
    formed on the device, MagickCore's own SetImageColorspace re-lays the cache out as one gray
    (+ alpha) channel and finds the device copy first."""
    hdri = dtype == np.float32
    px = make_pixels(50, 66, 4, dtype, seed=12)
    before = accelerated_calls(shim, hdri)
    g, c = shim.RefImage(px, shim=True), shim.RefImage(px)
    g.colorspace(target)
    c.colorspace(target)
    assert accelerated_calls(shim, hdri) == before + 1, "the GRAY transform did not take the accelerated path"
    assert g.info()["colorspace"] == c.info()["colorspace"]
    assert g.numpy().shape == c.numpy().shape
    assert_parity(g.numpy(), c.numpy(), True, "sRGB -> %s via MagickCore" % target, max_ulp=1)
    # and the gray image goes on through the accelerated operators (the gate admits GRAY)
    assert_parity(g.blur(0.0, 2.0).numpy(), c.blur(0.0, 2.0).numpy(), True, "BlurImage of the gray image")


# ----------------------------------------------------------------------- devices and queues
class _KernelProfileRecord(ctypes.Structure):          # MagickCore/opencl.h:33-43
    _fields_ = [("kernel_name", ctypes.c_char_p), ("count", ctypes.c_ulong), ("max", ctypes.c_ulong),
                ("min", ctypes.c_ulong), ("total", ctypes.c_ulong)]


def _device_api(lib):
    vp = ctypes.c_void_p
    lib.GetOpenCLDevices.restype = ctypes.POINTER(vp)
    lib.GetOpenCLDevices.argtypes = [ctypes.POINTER(ctypes.c_size_t), vp]
    for name in ("GetOpenCLDeviceName", "GetOpenCLDeviceVendorName", "GetOpenCLDeviceVersion"):
        getattr(lib, name).restype = ctypes.c_char_p
        getattr(lib, name).argtypes = [vp]
    lib.GetOpenCLDeviceType.restype = ctypes.c_int
    lib.GetOpenCLDeviceType.argtypes = [vp]
    lib.GetOpenCLDeviceEnabled.restype = ctypes.c_int
    lib.GetOpenCLDeviceEnabled.argtypes = [vp]
    lib.GetOpenCLDeviceBenchmarkScore.restype = ctypes.c_double
    lib.GetOpenCLDeviceBenchmarkScore.argtypes = [vp]
    lib.SetOpenCLDeviceEnabled.restype = None
    lib.SetOpenCLDeviceEnabled.argtypes = [vp, ctypes.c_int]
    lib.SetOpenCLKernelProfileEnabled.restype = None
    lib.SetOpenCLKernelProfileEnabled.argtypes = [vp, ctypes.c_int]
    lib.GetOpenCLKernelProfileRecords.restype = ctypes.POINTER(ctypes.POINTER(_KernelProfileRecord))
    lib.GetOpenCLKernelProfileRecords.argtypes = [vp, ctypes.POINTER(ctypes.c_size_t)]
    n = ctypes.c_size_t(0)
    devices = lib.GetOpenCLDevices(ctypes.byref(n), None)
    return [devices[i] for i in range(n.value)]


def test_public_device_api_lists_the_gpus_and_masks_them(shim, im):
    """MagickCore/opencl.h's device API on the HIP backend (the reference: opencl.c:1823-2130,
    :3127-3168): GetOpenCLDevices returns one MagickCLDevice per GPU with name / vendor / version /
    type / score / enabled; SetOpenCLDeviceEnabled is the arbitration's device mask — with every
    device off BlurImage runs on the CPU, back on it is accelerated again."""
    lib = shim._load(False, True)
    devices = _device_api(lib)
    assert len(devices) == im.logical_device_count() >= 1
    info = im.device_info(0)
    for d in devices:
        assert lib.GetOpenCLDeviceName(d).decode() == info["name"] or len(devices) > im.device_count()
        assert b"Advanced Micro Devices" in lib.GetOpenCLDeviceVendorName(d)
        assert lib.GetOpenCLDeviceVersion(d).decode().startswith("HIP gfx")
        assert lib.GetOpenCLDeviceType(d) == 2                       # GpuCLDeviceType
        assert lib.GetOpenCLDeviceEnabled(d) == 1
        assert lib.GetOpenCLDeviceBenchmarkScore(d) > 0.0
    assert lib.GetOpenCLDeviceName(None) is None and lib.GetOpenCLDeviceEnabled(None) == 0
    px = make_pixels(60, 70, 4, np.uint16, seed=31)
    want = shim.RefImage(px).blur(0.0, 1.5).numpy()
    try:
        for d in devices:
            lib.SetOpenCLDeviceEnabled(d, 0)
        before = accelerated_calls(shim, False)
        assert_parity(shim.RefImage(px, shim=True).blur(0.0, 1.5).numpy(), want, True, "every device off: CPU path")
        assert accelerated_calls(shim, False) == before
    finally:
        for d in devices:
            lib.SetOpenCLDeviceEnabled(d, 1)
    before = accelerated_calls(shim, False)
    resident = shim.RefImage(px, shim=True)
    first = resident.blur(0.0, 1.5)
    assert accelerated_calls(shim, False) == before + 1
    # the source is resident on a device that is then switched off: the next operator brings the
    # pixels back and runs elsewhere (here: the CPU, or another logical device)
    try:
        for d in devices:
            lib.SetOpenCLDeviceEnabled(d, 0)
        assert_parity(resident.blur(0.0, 1.5).numpy(), want, True, "resident on a disabled device")
    finally:
        for d in devices:
            lib.SetOpenCLDeviceEnabled(d, 1)
    assert_parity(first.numpy(), want, True, "BlurImage, devices back on")


def test_kernel_profile_records_through_the_device_api(shim):
    """SetOpenCLKernelProfileEnabled + GetOpenCLKernelProfileRecords (opencl.c:3162, :2081): the
    library's hipEvent records of the device, one per kernel, microseconds."""
    lib = shim._load(False, True)
    devices = _device_api(lib)
    px = make_pixels(300, 280, 4, np.uint16, seed=32)
    for d in devices:
        lib.SetOpenCLKernelProfileEnabled(d, 1)
    try:
        image = shim.RefImage(px, shim=True)
        image.blur(0.0, 2.0).numpy()
        image.equalize().numpy()
    finally:
        for d in devices:
            lib.SetOpenCLKernelProfileEnabled(d, 0)
    names, total = set(), 0
    for d in devices:
        n = ctypes.c_size_t(0)
        records = lib.GetOpenCLKernelProfileRecords(d, ctypes.byref(n))
        for i in range(n.value):
            r = records[i].contents
            assert r.count >= 1 and r.min <= r.max <= r.total
            names.add(r.kernel_name.decode())
            total += r.count
        if n.value:
            assert not records[n.value]                  # NULL-terminated like the reference's array
    assert total >= 2 and any("blur" in name or "conv" in name for name in names), names


def _run_threads_worker(threads, per_thread, env_extra, hdri=False):
    import json
    import subprocess
    import sys
    env = dict(os.environ, **env_extra)
    env["MAGICK_HIP_LIBRARY"] = os.path.join(ROOT, "imagemagick_amd", "lib", "libmagickhip.so")
    env["MAGICK_HIP_PRECISION"] = "exact"          # (the worker compares bit for bit; the default is FAST)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "helpers", "shim_threads.py"), str(threads), str(per_thread)]
    if hdri:
        cmd.append("hdri")
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("hdri", [False, True])
def test_eight_threads_spread_over_devices_and_streams(shim, hdri):
    """an unchanged MagickCore caller that converts a batch from 8 host threads
    reaches several GPUs.  MAGICKHIP_LOGICAL_DEVICES=4 maps four logical devices onto the GPUs
    present; every operator call is arbitrated (least-busy enabled device, RequestOpenCLDevice
    opencl.c:3056-3102) and gets one of the device's streams round-robin
    (AcquireOpenCLCommandQueue, opencl.c:656); chained operators stay on the device and stream of
    their image.  Bit-identical to the CPU MagickCore."""
    report = _run_threads_worker(8, 3, {"MAGICKHIP_LOGICAL_DEVICES": "4"}, hdri)
    assert report["errors"] == [] and report["mismatches"] == 0 and report["images"] == 24, report
    assert report["devices"] == 4
    assert report["accelerated"] == 48                               # 24 x (BlurImage + EqualizeImage)
    assert sum(report["calls"]) == 48
    assert sum(1 for c in report["calls"] if c > 0) >= 2, report     # landed on >= 2 logical devices
    assert all(s >= 2 for c, s in zip(report["calls"], report["streams"]) if c >= 4), report


def _run_spread_worker(edge, extra_env, hdri=False):
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.update(extra_env)
    env["MAGICK_HIP_PRECISION"] = "exact"
    env["MAGICK_HIP_LIBRARY"] = os.path.join(ROOT, "imagemagick_amd", "lib", "libmagickhip.so")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "helpers", "shim_spread.py"), str(edge)]
    if hdri:
        cmd.append("hdri")
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize("hdri", [False, True])
def test_one_big_host_image_goes_over_every_device(shim, hdri):
    """BASELINE configs[4] through the boundary: an unchanged caller's MorphologyImage (Dilate Disk:15),
    BlurImage and EqualizeImage on ONE big host-resident image.  The reference hands an operator one
    device (opencl.c:3056-3102) and that device's host link then carries the whole frame both ways; with
    more than one device enabled and a frame of MAGICK_HIP_SPREAD_BYTES or more the binding runs the
    stencil operators on the host blocks with MH_DEVICE_ALL (row bands round all devices) and the
    histogram operator through MagickHipShardedImage (one band per device, the table all-reduced).
    Four logical devices on the GPU present; bit-identical to the CPU MagickCore."""
    report = _run_spread_worker(3072 if hdri else 4096, {"MAGICKHIP_LOGICAL_DEVICES": "4",
                                                         "MAGICK_HIP_SPREAD_BYTES": str(100 << 20)}, hdri)
    assert report["mismatches"] == [], report
    assert report["accelerated"] == 3 and report["devices"] == 4, report
    assert all(c == 3 for c in report["calls"]), report              # every device took part in every call
    assert sum(1 for b in report["bands"] if b > 0) >= 2, report     # the stencil bands went round the devices


def test_one_big_host_image_on_two_physical_gpus(shim, im):
    """The same over the physical devices of a multi-GPU node (no logical mapping)."""
    if im.device_count() < 2:
        pytest.skip("needs two physical GPUs")
    report = _run_spread_worker(8192, {"MAGICK_HIP_SPREAD_BYTES": str(256 << 20)})
    assert report["mismatches"] == [], report
    assert report["accelerated"] == 3 and report["devices"] == im.device_count(), report
    assert sum(1 for b in report["bands"] if b > 0) >= 2, report


def test_eight_threads_on_two_physical_gpus(shim, im):
    """The same through the physical devices of a multi-GPU node (no logical mapping)."""
    if im.device_count() < 2:
        pytest.skip("needs two physical GPUs")
    report = _run_threads_worker(8, 3, {})
    assert report["errors"] == [] and report["mismatches"] == 0 and report["images"] == 24, report
    assert report["devices"] == im.device_count()
    assert sum(1 for c in report["calls"] if c > 0) >= 2, report
