"""imagemagick_amd — Python binding of libmagickhip.so, the MI355X-native
MagickCore accelerate backend (see include/magickhip.h).

The functions here mirror the reference's operator names (BlurImage,
ConvolveImage, MorphologyImage, UnsharpMaskImage, ResizeImage,
ContrastStretchImage, EqualizeImage, TransformImageColorspace) and do nothing
but marshal buffers into the C ABI: pixels may be a CUDA/HIP ``torch.Tensor``
(used in place on the device, on torch's current stream) or a NumPy array
(host memory, staged by the library).  Layout is the pixel cache's:
``[rows, columns, channels]``, ``uint16`` (Q16) or ``float32`` (Q16-HDRI).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import (MagickHipError, MhImage, COLORSPACES, MORPHOLOGY, FILTERS,  # noqa: F401
                   PRECISION_EXACT, PRECISION_FAST, TRAIT_COPY, TRAIT_UPDATE, TRAIT_BLEND,
                   ALL_CHANNELS, SYNC_CHANNELS)

__all__ = ["Image", "blur_image", "convolve_image", "morphology_image", "morphology_primitive",
           "unsharp_mask_image", "resize_image", "contrast_stretch_image", "equalize_image",
           "transform_image_colorspace", "wavelet_denoise_image", "despeckle_image", "local_contrast_image", "rotational_blur_image", "motion_blur_image", "gaussian_blur_image", "sharpen_image", "edge_image",
           "emboss_image", "import_image_pixels", "export_image_pixels", "contrast_image", "modulate_image", "grayscale_image", "function_image", "histogram", "apply_lut", "contrast_stretch_lut",
           "equalize_lut", "is_image_gray", "set_precision", "get_precision", "set_option", "get_option", "option",
           "logical_device_count", "device_info", "device_count",
           "build", "load", "MagickHipError"]

build = _lib.build
load = _lib.load


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Image:
    """A pixel buffer plus the per-image facts the operators need (what the
    MagickCore shim reads from ``Image``): channel traits, alpha, colourspace."""

    def __init__(self, pixels, colorspace="srgb", has_alpha=None, channel_mask=ALL_CHANNELS,
                 copy_channels=(), intensity=0, stream=None, precision=None):
        # stream: the raw HIP stream handle the pixels belong to (device memory; default: torch's
        # current stream of the tensor's device when a descriptor is made)
        self.stream = stream
        # precision: PRECISION_EXACT / PRECISION_FAST for the operators called on THIS image
        # (MhImage::precision), None = the library default (set_precision)
        self.precision = precision
        if pixels.ndim == 2:
            pixels = pixels.reshape(pixels.shape[0], pixels.shape[1], 1)
        if pixels.ndim != 3:
            raise ValueError("pixels must be [rows, columns, channels]")
        self.pixels = pixels
        self.rows, self.columns, self.channels = (int(s) for s in pixels.shape)
        if has_alpha is None:
            has_alpha = self.channels in (2, 4)
        self.has_alpha = bool(has_alpha)
        self.colorspace = colorspace.lower()
        self.channel_mask = channel_mask
        self.copy_channels = tuple(copy_channels)
        self.intensity = intensity
        if _is_torch(pixels):
            import torch
            if not pixels.is_contiguous():
                raise ValueError("pixel tensor must be contiguous")
            if pixels.dtype == torch.uint16:
                self.quantum = _lib.QUANTUM_U16
            elif pixels.dtype == torch.float32:
                self.quantum = _lib.QUANTUM_F32
            else:
                raise ValueError("pixels must be uint16 (Q16) or float32 (Q16-HDRI)")
            if not pixels.is_cuda:
                raise ValueError("torch pixels must live on the GPU; pass a NumPy array for host memory")
            self.memory = _lib.MEMORY_DEVICE
        else:
            if not pixels.flags["C_CONTIGUOUS"]:
                raise ValueError("pixel array must be C-contiguous")
            if pixels.dtype == np.uint16:
                self.quantum = _lib.QUANTUM_U16
            elif pixels.dtype == np.float32:
                self.quantum = _lib.QUANTUM_F32
            else:
                raise ValueError("pixels must be uint16 (Q16) or float32 (Q16-HDRI)")
            self.memory = _lib.MEMORY_HOST

    # ------------------------------------------------------------------ helpers
    def _pointer(self):
        if self.memory == _lib.MEMORY_DEVICE:
            return self.pixels.data_ptr()
        return self.pixels.ctypes.data

    def descriptor(self):
        lib = _lib.load()
        d = MhImage()
        lib.MhInitImage(ctypes.byref(d), self._pointer(), self.columns, self.rows, self.channels,
                        1 if self.has_alpha else 0, self.quantum, self.memory)
        d.colorspace = COLORSPACES[self.colorspace]
        d.intensity = self.intensity
        d.channel_mask = self.channel_mask
        d.precision = 0 if self.precision is None else int(self.precision) + 1
        for c in self.copy_channels:
            d.channel_traits[c] = TRAIT_COPY
        if self.memory == _lib.MEMORY_DEVICE:
            import torch
            d.device = self.pixels.device.index if self.pixels.device.index is not None else 0
            d.stream = self.stream if self.stream is not None else \
                torch.cuda.current_stream(self.pixels.device).cuda_stream
        return d

    def like(self, rows=None, columns=None):
        rows = self.rows if rows is None else rows
        columns = self.columns if columns is None else columns
        if self.memory == _lib.MEMORY_DEVICE:
            import torch
            px = torch.empty((rows, columns, self.channels), dtype=self.pixels.dtype,
                             device=self.pixels.device)
        else:
            px = np.empty((rows, columns, self.channels), dtype=self.pixels.dtype)
        return Image(px, self.colorspace, self.has_alpha, self.channel_mask, self.copy_channels,
                     self.intensity, precision=self.precision)

    def numpy(self):
        if self.memory == _lib.MEMORY_DEVICE:
            import torch
            t = self.pixels
            if t.dtype == torch.uint16:
                return t.view(torch.int16).cpu().numpy().view(np.uint16)
            return t.cpu().numpy()
        return self.pixels


class _Kernel:
    def __init__(self, kernel, scale=None):
        """`scale`: None, or (factor, flags) for ScaleKernelInfo — ("!" normalise = flag 1,
        "^" correlate-normalise = flag 2): what `-define convolve:scale=...` does to the kernel
        MorphologyImage is about to use (morphology.c:4144-4160)."""
        lib = _lib.load()
        self.owned = isinstance(kernel, (str, bytes))
        if self.owned:
            text = kernel.encode() if isinstance(kernel, str) else kernel
            self.ptr = lib.MhAcquireKernelInfo(text)
            if not self.ptr:
                raise MagickHipError(3, lib.MhGetLastError().decode())
            if scale is not None:
                lib.MhScaleKernelInfo(self.ptr, float(scale[0]), int(scale[1]))
        else:
            self.ptr = kernel

    def __enter__(self):
        return self.ptr

    def __exit__(self, *exc):
        if self.owned and self.ptr:
            _lib.load().MhDestroyKernelInfo(self.ptr)
        return False


def optimal_kernel_width_1d(radius, sigma):
    """GetOptimalKernelWidth1D — MagickCore/gem.c:262 (the tap count BlurImage uses)."""
    return int(_lib.load().MhGetOptimalKernelWidth1D(radius, sigma))


def kernel_outer_product_factors(kernel_string):
    """(row, column) when the first kernel of the string is an outer product column x row —
    what FAST ConvolveImage separates into two 1-D passes — else None."""
    lib = _lib.load()
    ptr = lib.MhAcquireKernelInfo(kernel_string.encode())
    if not ptr:
        raise MagickHipError(3, lib.MhGetLastError().decode())
    try:
        k = ptr.contents
        row = np.empty(k.width, dtype=np.float64)
        column = np.empty(k.height, dtype=np.float64)
        ok = lib.MhKernelOuterProductFactors(ptr, row.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                             column.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        return (row, column) if ok else None
    finally:
        lib.MhDestroyKernelInfo(ptr)


def kernel_outer_product_plus_delta(kernel_string):
    """(kind, row, column, delta) with kind 1 = outer product, 2 = outer product + delta at the
    origin cell (what the EXACT separable path takes), or None."""
    lib = _lib.load()
    ptr = lib.MhAcquireKernelInfo(kernel_string.encode())
    if not ptr:
        raise MagickHipError(3, lib.MhGetLastError().decode())
    try:
        k = ptr.contents
        row = np.empty(k.width, dtype=np.float64)
        column = np.empty(k.height, dtype=np.float64)
        delta = ctypes.c_double(0.0)
        kind = lib.MhKernelOuterProductPlusDelta(ptr, row.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                 column.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                 ctypes.byref(delta))
        return (kind, row, column, delta.value) if kind else None
    finally:
        lib.MhDestroyKernelInfo(ptr)


def kernel_integer_cells(kernel_string, scale=None):
    """(cells, unit) with kernel = cells * unit (integers of at most seven bits; NaN cells as 0) — the
    form the exact-integer 2-D convolve takes — or None.  scale: (factor, normalize flags) applied
    first, as `-define convolve:scale` does."""
    lib = _lib.load()
    ptr = lib.MhAcquireKernelInfo(kernel_string.encode())
    if not ptr:
        raise MagickHipError(3, lib.MhGetLastError().decode())
    try:
        if scale is not None:
            lib.MhScaleKernelInfo(ptr, float(scale[0]), int(scale[1]))
        k = ptr.contents
        cells = np.empty((k.height, k.width), dtype=np.int32)
        unit = ctypes.c_double(0.0)
        ok = lib.MhKernelIntegerCells(ptr, cells.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.byref(unit))
        return (cells, unit.value) if ok else None
    finally:
        lib.MhDestroyKernelInfo(ptr)


def kernel_to_numpy(kernel_string, index=0, scale=None):
    """Build a kernel list with the product's host builder and return kernel
    `index` as (values[h,w], x, y, count).  scale: (factor, normalize flags) applied first, as
    `-define convolve:scale` does."""
    lib = _lib.load()
    ptr = lib.MhAcquireKernelInfo(kernel_string.encode())
    if not ptr:
        raise MagickHipError(3, lib.MhGetLastError().decode())
    try:
        if scale is not None:
            lib.MhScaleKernelInfo(ptr, float(scale[0]), int(scale[1]))
        count = 0
        k = ptr
        chosen = None
        while k:
            if count == index:
                chosen = k.contents
            count += 1
            k = k.contents.next
        if chosen is None:
            raise IndexError(index)
        n = chosen.width * chosen.height
        values = np.ctypeslib.as_array(chosen.values, shape=(n,)).copy()
        return values.reshape(chosen.height, chosen.width), chosen.x, chosen.y, count
    finally:
        lib.MhDestroyKernelInfo(ptr)


# --------------------------------------------------------------------- runtime
def host_alloc(shape, dtype):
    """A NumPy array in page-locked host memory (MhHostAlloc): what a MagickCore pixel cache is
    once the shim's allocator is installed (SetMagickAlignedMemoryMethods).  Operators move such
    a buffer with one DMA transfer per direction instead of through the staging threads.  The
    memory is released when the array (and every view of it) is gone."""
    import weakref
    L = _lib.load()
    dtype = np.dtype(dtype)
    count = int(np.prod(shape))
    block = L.MhHostAlloc(max(count * dtype.itemsize, 1))
    if not block:
        raise MemoryError("MhHostAlloc(%d bytes)" % (count * dtype.itemsize))
    raw = (ctypes.c_char * (count * dtype.itemsize)).from_address(block)
    weakref.finalize(raw, L.MhHostFree, ctypes.c_void_p(block))
    return np.frombuffer(raw, dtype=dtype, count=count).reshape(shape)


def host_allocated_bytes():
    return int(_lib.load().MhHostAllocatedBytes())


def device_count():
    return _lib.load().MhDeviceCount()


def set_precision(precision):
    return _lib.load().MhSetPrecision(precision)


def get_precision():
    return _lib.load().MhGetPrecision()


def set_option(name, value):
    """MhSetOption: the library reads MAGICKHIP_* / MAGICK_HIP_* from the environment once, at
    start-up; this changes the value it holds (None = unset).  Tests and A/B timings only."""
    _lib.check(_lib.load().MhSetOption(name.encode(), None if value is None else str(value).encode()))


def get_option(name):
    v = _lib.load().MhGetOption(name.encode())
    return None if v is None else v.decode()


class option:
    """with option("MAGICKHIP_NO_MFMA", "1"): ...  — a switch for the duration of a block."""

    def __init__(self, name, value="1"):
        self.name, self.value = name, value

    def __enter__(self):
        self.previous = get_option(self.name)
        set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.previous)
        return False


def logical_device_count():
    return _lib.load().MhLogicalDeviceCount()


def device_info(device=0):
    info = _lib.MhDeviceInfo()
    _lib.check(_lib.load().MhGetDeviceInfo(device, ctypes.byref(info)))
    return {"name": info.name.decode(), "architecture": info.architecture.decode(),
            "compute_units": info.compute_units, "clock_mhz": info.clock_mhz,
            "global_memory": info.global_memory, "local_memory": info.local_memory}


# ------------------------------------------------------------------- operators
def blur_image(image, radius, sigma, out=None):
    """BlurImage(image, radius, sigma) — MagickCore/effect.c:765.  `out`: an Image of the same
    geometry to receive the result (what MagickCore does: the destination is a pixel cache that
    already exists), instead of a freshly allocated one."""
    lib = _lib.load()
    if out is None:
        out = image.like()
    _lib.check(lib.MagickHipBlurImage(ctypes.byref(image.descriptor()),
                                      ctypes.byref(out.descriptor()), radius, sigma))
    return out


def convolve_image(image, kernel):
    """ConvolveImage(image, kernel) — MagickCore/effect.c:1170."""
    lib = _lib.load()
    out = image.like()
    with _Kernel(kernel) as k:
        _lib.check(lib.MagickHipConvolveImage(ctypes.byref(image.descriptor()),
                                              ctypes.byref(out.descriptor()), k))
    return out


MORPHOLOGY_COMPOSE = {None: 0, "undefined": 0, "none": 1, "no": 1, "lighten": 2, "difference": 3, "darken": 5,
                      "plus": 6, "multiply": 7, "screen": 8, "exclusion": 9, "minussrc": 10, "minusdst": 11,
                      "lineardodge": 12, "over": 13, "srcover": 13, "dstover": 14}


def morphology_image(image, method, iterations, kernel, bias=0.0, scale=None, compose=None):
    """MorphologyImage(image, method, iterations, kernel) — MagickCore/morphology.c:4129.
    scale=(1.0, 1) is `-define convolve:scale='!'` (the kernel normalised before use);
    compose is `-define morphology:compose=` (None: the method's default; "None", "Lighten",
    "Difference", "Darken", "Plus", "Multiply", "Screen": how the results of a kernel list are merged; any other operator raises
    MagickHipError: the CPU path's business)."""
    lib = _lib.load()
    out = image.like()
    key = compose.lower() if isinstance(compose, str) else compose
    with _Kernel(kernel, scale) as k:
        _lib.check(lib.MagickHipMorphologyImageCompose(ctypes.byref(image.descriptor()),
                                                       ctypes.byref(out.descriptor()),
                                                       MORPHOLOGY[method.lower()], iterations, k, bias,
                                                       MORPHOLOGY_COMPOSE.get(key, 4)))
    return out


def morphology_primitive(image, method, kernel, bias=0.0):
    """One MorphologyPrimitive pass; returns (image, changed)."""
    lib = _lib.load()
    out = image.like()
    changed = ctypes.c_ssize_t(0)
    with _Kernel(kernel) as k:
        _lib.check(lib.MagickHipMorphologyPrimitive(ctypes.byref(image.descriptor()),
                                                    ctypes.byref(out.descriptor()),
                                                    MORPHOLOGY[method.lower()], k, bias,
                                                    ctypes.byref(changed)))
    return out, changed.value


def _pair_operator(name, image, *args):
    lib = _lib.load()
    out = image.like()
    _lib.check(getattr(lib, name)(ctypes.byref(image.descriptor()), ctypes.byref(out.descriptor()), *args))
    return out


def wavelet_denoise_image(image, threshold, softness=0.0):
    """WaveletDenoiseImage(image, threshold, softness) — MagickCore/visual-effects.c:3520."""
    return _pair_operator("MagickHipWaveletDenoiseImage", image, threshold, softness)


def despeckle_image(image):
    """DespeckleImage(image) — MagickCore/effect.c:1308."""
    return _pair_operator("MagickHipDespeckleImage", image)


def local_contrast_image(image, radius, strength):
    """LocalContrastImage(image, radius, strength) — MagickCore/effect.c:1760."""
    return _pair_operator("MagickHipLocalContrastImage", image, radius, strength)


def rotational_blur_image(image, angle):
    """RotationalBlurImage(image, angle) — MagickCore/effect.c:3209."""
    return _pair_operator("MagickHipRotationalBlurImage", image, angle)


def motion_blur_image(image, radius, sigma, angle):
    """MotionBlurImage(image, radius, sigma, angle) — MagickCore/effect.c:2347."""
    return _pair_operator("MagickHipMotionBlurImage", image, radius, sigma, angle)


def gaussian_blur_image(image, radius, sigma):
    """GaussianBlurImage — MagickCore/effect.c:1709 (one 2-D kernel, not the separable blur)."""
    return _pair_operator("MagickHipGaussianBlurImage", image, radius, sigma)


def sharpen_image(image, radius, sigma):
    """SharpenImage — MagickCore/effect.c:3991."""
    return _pair_operator("MagickHipSharpenImage", image, radius, sigma)


def edge_image(image, radius):
    """EdgeImage — MagickCore/effect.c:1523."""
    return _pair_operator("MagickHipEdgeImage", image, radius)


def emboss_image(image, radius, sigma):
    """EmbossImage — MagickCore/effect.c:1600 (convolve, then EqualizeImage of the result)."""
    return _pair_operator("MagickHipEmbossImage", image, radius, sigma)


def unsharp_mask_image(image, radius, sigma, gain, threshold):
    """UnsharpMaskImage — MagickCore/effect.c:4256."""
    lib = _lib.load()
    out = image.like()
    _lib.check(lib.MagickHipUnsharpMaskImage(ctypes.byref(image.descriptor()),
                                             ctypes.byref(out.descriptor()), radius, sigma, gain,
                                             threshold))
    return out


def resize_image(image, columns, rows, filter="lanczos"):
    """ResizeImage(image, columns, rows, filter) — MagickCore/resize.c:3761."""
    lib = _lib.load()
    out = image.like(rows=rows, columns=columns)
    _lib.check(lib.MagickHipResizeImage(ctypes.byref(image.descriptor()),
                                        ctypes.byref(out.descriptor()), FILTERS[filter.lower()]))
    return out


def contrast_stretch_image(image, black_point, white_point):
    """ContrastStretchImage(image, black, white), in place — MagickCore/enhance.c:1544."""
    lib = _lib.load()
    gray = ctypes.c_int(0)
    _lib.check(lib.MagickHipContrastStretchImage(ctypes.byref(image.descriptor()), black_point,
                                                 white_point, ctypes.byref(gray)))
    return image


def equalize_image(image):
    """EqualizeImage(image), in place — MagickCore/enhance.c:2040."""
    lib = _lib.load()
    _lib.check(lib.MagickHipEqualizeImage(ctypes.byref(image.descriptor())))
    return image


def transform_colorspace_contrast_stretch_image(image, colorspace, black_point, white_point):
    """TransformImageColorspace then ContrastStretchImage as one call (the two calls' results; a
    FAST sRGB -> Lab of an RGBA Q16 frame shares its pass over the pixels with the histogram)."""
    lib = _lib.load()
    d = image.descriptor()
    _lib.check(lib.MagickHipTransformColorspaceContrastStretchImage(
        ctypes.byref(d), COLORSPACES[colorspace.lower()], black_point, white_point))
    image.colorspace = colorspace.lower()
    return image


def transform_image_colorspace(image, colorspace):
    """TransformImageColorspace(image, colorspace), in place — MagickCore/colorspace.c:1751."""
    lib = _lib.load()
    d = image.descriptor()
    _lib.check(lib.MagickHipTransformImageColorspace(ctypes.byref(d), COLORSPACES[colorspace.lower()]))
    image.colorspace = colorspace.lower()
    return image


def grayscale_image(image, method="rec709luma"):
    """GrayscaleImage(image, method), in place (first channel) — MagickCore/enhance.c:2476."""
    lib = _lib.load()
    _lib.check(lib.MagickHipGrayscaleImage(ctypes.byref(image.descriptor()), _lib.INTENSITY[method.lower()]))
    return image


# StorageType, MagickCore/pixel.h:146-156
STORAGE = {"uint8": 1, "float64": 2, "float32": 3, "uint32": 4, "uint64": 5, "uint16": 7}
_TORCH_STORAGE = {"torch.uint8": 1, "torch.float64": 2, "torch.float32": 3, "torch.uint32": 4, "torch.uint64": 5,
                  "torch.uint16": 7, "torch.int16": 7, "torch.int32": 4, "torch.int64": 5}


def _component_buffer(data):
    """(pointer, storage, memory kind, height, width, components, keep-alive) of a NumPy array or
    a CUDA tensor shaped [height, width, components]."""
    if isinstance(data, np.ndarray):
        data = np.ascontiguousarray(data)
        return data.ctypes.data, STORAGE[data.dtype.name], _lib.MEMORY_HOST, data
    data = data.contiguous()
    return data.data_ptr(), _TORCH_STORAGE[str(data.dtype)], _lib.MEMORY_DEVICE, data


def import_image_pixels(image, x, y, map, data):
    """ImportImagePixels(image, x, y, width, height, map, type, pixels) — MagickCore/pixel.c:4164.
    data: [height, width, len(map)] NumPy array (host) or CUDA tensor (device)."""
    lib = _lib.load()
    ptr, storage, memory, keep = _component_buffer(data)
    assert keep.shape[2] == len(map)
    _lib.check(lib.MagickHipImportImagePixels(ctypes.byref(image.descriptor()), x, y, keep.shape[1], keep.shape[0],
                                              map.encode(), storage, ptr, memory))
    return image


def export_image_pixels(image, x, y, map, out):
    """ExportImagePixels into `out` ([height, width, len(map)] NumPy array or CUDA tensor) —
    MagickCore/pixel.c:1962."""
    lib = _lib.load()
    if isinstance(out, np.ndarray):
        assert out.flags["C_CONTIGUOUS"]
    else:
        assert out.is_contiguous()
    ptr, storage, memory, keep = _component_buffer(out)
    assert keep.shape[2] == len(map)
    _lib.check(lib.MagickHipExportImagePixels(ctypes.byref(image.descriptor()), x, y, keep.shape[1], keep.shape[0],
                                              map.encode(), storage, ptr, memory))
    return out


def contrast_image(image, sharpen=True):
    """ContrastImage(image, sharpen), in place — MagickCore/enhance.c:1392."""
    lib = _lib.load()
    _lib.check(lib.MagickHipContrastImage(ctypes.byref(image.descriptor()), 1 if sharpen else 0))
    return image


def modulate_image(image, brightness=100.0, saturation=100.0, hue=100.0, colorspace=None):
    """ModulateImage(image, "brightness,saturation,hue"), in place — MagickCore/enhance.c:3665.
    colorspace: None / "HSL" (default model), "HSB", "HCL", "HCLp", "HSI", "HSV", "HWB", "LCH",
    "LCHab" or "LCHuv" (enhance.c:3826-3890)."""
    lib = _lib.load()
    model = 0 if not colorspace else COLORSPACES[colorspace.lower()]
    _lib.check(lib.MagickHipModulateImage(ctypes.byref(image.descriptor()), brightness, saturation, hue, model))
    return image


def function_image(image, function, parameters):
    """FunctionImage(image, function, parameters), in place — MagickCore/statistic.c:1069."""
    lib = _lib.load()
    params = (ctypes.c_double * max(1, len(parameters)))(*parameters)
    _lib.check(lib.MagickHipFunctionImage(ctypes.byref(image.descriptor()), _lib.FUNCTIONS[function.lower()],
                                          len(parameters), params))
    return image


# ------------------------------------------------------------- building blocks
def histogram(image, intensity_mode, out=None):
    """Accumulate the 65536-bin histogram of `image` into `out`
    ([65536, channels] uint64; a CUDA tensor of int64 for device images)."""
    lib = _lib.load()
    if image.memory == _lib.MEMORY_DEVICE:
        import torch
        if out is None:
            out = torch.zeros((65536, image.channels), dtype=torch.int64, device=image.pixels.device)
        ptr = out.data_ptr()
    else:
        if out is None:
            out = np.zeros((65536, image.channels), dtype=np.uint64)
        ptr = out.ctypes.data
    _lib.check(lib.MagickHipHistogram(ctypes.byref(image.descriptor()), 1 if intensity_mode else 0, ptr))
    return out


def contrast_stretch_lut(hist, columns, rows, black_point, white_point, quantum):
    lib = _lib.load()
    hist = np.ascontiguousarray(hist, dtype=np.uint64)
    channels = hist.shape[1]
    lut = np.zeros((65536, channels), dtype=np.float64)
    mask = ctypes.c_uint32(0)
    _lib.check(lib.MhContrastStretchLUT(hist.ctypes.data, channels, columns, rows, black_point,
                                        white_point, quantum, lut.ctypes.data, ctypes.byref(mask)))
    return lut, mask.value


def equalize_lut(hist, quantum):
    lib = _lib.load()
    hist = np.ascontiguousarray(hist, dtype=np.uint64)
    channels = hist.shape[1]
    lut = np.zeros((65536, channels), dtype=np.float64)
    mask = ctypes.c_uint32(0)
    _lib.check(lib.MhEqualizeLUT(hist.ctypes.data, channels, quantum, lut.ctypes.data,
                                 ctypes.byref(mask)))
    return lut, mask.value


def apply_lut(image, lut, mask):
    lib = _lib.load()
    lut = np.ascontiguousarray(lut, dtype=np.float64)
    _lib.check(lib.MagickHipApplyLUT(ctypes.byref(image.descriptor()), lut.ctypes.data, mask))
    return image


def is_image_gray(image):
    lib = _lib.load()
    flag = ctypes.c_int(0)
    _lib.check(lib.MagickHipIsImageGray(ctypes.byref(image.descriptor()), ctypes.byref(flag)))
    return bool(flag.value)


def apply_histogram(image, hist, intensity_mode, equalize, black_point=0.0, white_point=0.0, image_rows=0):
    """MagickHipApplyHistogram: LUT construction + application on the device from a histogram the
    caller holds (a CUDA int64/uint64 tensor for device images, a NumPy uint64 array for host ones)."""
    lib = _lib.load()
    d = image.descriptor()
    pointer = hist.data_ptr() if _is_torch(hist) else hist.ctypes.data
    _lib.check(lib.MagickHipApplyHistogram(ctypes.byref(d), pointer, 1 if intensity_mode else 0,
                                           1 if equalize else 0, float(black_point), float(white_point),
                                           int(image_rows)))
    return image


# ------------------------------------------------------- batches / several GPUs
def _operators(chain):
    """[("colorspace", "Lab"), ("contraststretch", black, white), ("blur", 0, 10), ("morphology",
    "Dilate", 1, "Disk:15"), ("unsharpmask", 0, 10, 1.0, 0.02), ("resize", columns, rows, "Lanczos"),
    ("equalize",)] -> an MhOperator array (and the byte strings it points at)."""
    ops = (_lib.MhOperator * len(chain))()
    keep = []
    for i, step in enumerate(chain):
        name = step[0].lower()
        ops[i].kind = _lib.OPERATORS[name]
        args = list(step[1:])
        if name == "colorspace":
            args = [COLORSPACES[args[0].lower()]]
        elif name == "morphology":
            text = args[2].encode()
            keep.append(text)
            ops[i].text = text
            args = [MORPHOLOGY[args[0].lower()], args[1]]
        elif name == "resize":
            args = [args[0], args[1], FILTERS[args[2].lower()] if len(args) > 2 else FILTERS["lanczos"]]
        for k, a in enumerate(args):
            ops[i].args[k] = float(a)
    return ops, keep


def _report(r):
    return {"devices": int(r.devices), "workers": int(r.workers), "used_rccl": bool(r.used_rccl),
            "halo_exchanges": int(r.halo_exchanges),
            "images_per_device": [int(r.images_per_device[d]) for d in range(int(r.devices))],
            "seconds": float(r.seconds)}


def batch_images(chain, images, results=None, devices=0, streams_per_device=0):
    """MagickHipBatchImages: the operator chain on every image, images spread over `devices`
    logical devices (0 = all GPUs) x `streams_per_device` host threads.  results=None: in place."""
    lib = _lib.load()
    ops, keep = _operators(chain)
    src = (MhImage * len(images))(*[im.descriptor() for im in images])
    dst = None
    if results is not None:
        dst = (MhImage * len(results))(*[im.descriptor() for im in results])
    report = _lib.MhBatchReport()
    _lib.check(lib.MagickHipBatchImages(ops, len(chain), src, dst, len(images), devices,
                                        streams_per_device, ctypes.byref(report)))
    out = results if results is not None else images
    descs = dst if dst is not None else src
    names = {v: k for k, v in COLORSPACES.items()}
    for im, d in zip(out, descs):
        im.colorspace = names.get(int(d.colorspace), im.colorspace)
    return _report(report)


def sharded_image(chain, image, result=None, devices=0):
    """MagickHipShardedImage: the chain on ONE image cut into `devices` row bands (halo exchange
    between bands before every stencil, one all-reduce of the histogram table for
    ContrastStretch / Equalize).  Returns (result image, report)."""
    lib = _lib.load()
    ops, keep = _operators(chain)
    if result is None:
        result = image.like()
    src, dst = image.descriptor(), result.descriptor()
    report = _lib.MhBatchReport()
    _lib.check(lib.MagickHipShardedImage(ops, len(chain), ctypes.byref(src), ctypes.byref(dst), devices,
                                         ctypes.byref(report)))
    names = {v: k for k, v in COLORSPACES.items()}
    result.colorspace = names.get(int(dst.colorspace), result.colorspace)
    return result, _report(report)
