"""ctypes access to the HIP-backed MagickCore build (shim/_build/libMagickCore-hip-*.so): the
reference's MagickCore with shim/accelerate_hip.c + shim/opencl_hip.c slotted in.  bench.py uses it
to time the drop-in boundary itself — MagickCore's BlurImage / TransformImageColorspace /
ContrastStretchImage called as an application would call them, pixel caches page-locked by the
shim's allocator, results brought back by the cache's own lazy sync.

Nothing here touches oracle/: this is the product's binding, driven through the small C driver
linked into the build (ref_driver.c's operator wrappers + shim_driver.c).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
_LIBS = {}


def lib_path(hdri=False):
    return os.path.join(_HERE, "_build", "libMagickCore-hip-%s.so" % ("q16hdri" if hdri else "q16"))


def available(hdri=False):
    return os.path.exists(lib_path(hdri))


def load(hdri=False):
    if hdri in _LIBS:
        return _LIBS[hdri]
    os.environ.setdefault("MAGICK_HIP_LIBRARY", os.path.join(ROOT, "imagemagick_amd", "lib", "libmagickhip.so"))
    os.environ.setdefault("MAGICK_CONFIGURE_PATH", os.path.join(_HERE, "_build", "config"))
    L = ctypes.CDLL(lib_path(hdri), mode=ctypes.RTLD_LOCAL)
    vp, sz, dbl, cp = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_char_p
    pd = ctypes.POINTER(dbl)
    L.ref_init.restype = ctypes.c_int
    L.ref_image_new.restype = vp
    L.ref_image_new.argtypes = [sz, sz, cp, cp, vp]
    L.ref_image_free.argtypes = [vp]
    L.ref_image_get.argtypes = [vp, vp]
    L.ref_blur.restype = vp
    L.ref_blur.argtypes = [vp, dbl, dbl, pd]
    L.ref_colorspace.argtypes = [vp, cp, pd]
    L.ref_contrast_stretch.argtypes = [vp, dbl, dbl, pd]
    L.ref_equalize.argtypes = [vp, pd]
    L.ref_morphology.restype = vp
    L.ref_morphology.argtypes = [vp, cp, ctypes.c_long, cp, pd]
    L.shim_image_sync.argtypes = [vp]
    L.shim_image_touch.argtypes = [vp]
    L.shim_image_pixels.restype = vp
    L.shim_image_pixels.argtypes = [vp, ctypes.POINTER(sz)]
    L.SetOpenCLEnabled.argtypes = [ctypes.c_int]
    L.GetMagickHipAcceleratedCalls.restype = sz
    L.GetMagickHipDeviceStatistics.restype = sz
    L.GetMagickHipDeviceStatistics.argtypes = [sz, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    L.ref_init()
    L.SetOpenCLEnabled(1)          # acceleration on: pixel caches of 4 MiB and more are page-locked from here on
    _LIBS[hdri] = L
    return L


class Image:
    """An image in MagickCore's pixel cache (RGBA / RGB / gray [+ alpha] from a NumPy array)."""

    MAPS = {1: b"GRAY", 2: b"GRAYA", 3: b"RGB", 4: b"RGBA"}

    def __init__(self, pixels=None, handle=None, lib=None):
        if handle is not None:
            self.L, self.handle = lib, handle
            return
        self.L = load(pixels.dtype.name == "float32")
        rows, cols, ch = pixels.shape
        self.handle = self.L.ref_image_new(cols, rows, self.MAPS[ch], b"sRGB", pixels.ctypes.data)
        if not self.handle:
            raise RuntimeError("MagickCore could not create the image")

    def __del__(self):
        if getattr(self, "handle", None):
            self.L.ref_image_free(self.handle)
            self.handle = None

    def blur(self, radius, sigma):
        t = ctypes.c_double(0.0)
        h = self.L.ref_blur(self.handle, radius, sigma, ctypes.byref(t))
        if not h:
            raise RuntimeError("BlurImage failed")
        return Image(handle=h, lib=self.L)

    def morphology(self, method, iterations, kernel):
        t = ctypes.c_double(0.0)
        h = self.L.ref_morphology(self.handle, method.encode(), iterations, kernel.encode(), ctypes.byref(t))
        if not h:
            raise RuntimeError("MorphologyImage failed")
        return Image(handle=h, lib=self.L)

    def equalize(self):
        t = ctypes.c_double(0.0)
        if self.L.ref_equalize(self.handle, ctypes.byref(t)) != 0:
            raise RuntimeError("EqualizeImage failed")
        return self

    def colorspace(self, name):
        t = ctypes.c_double(0.0)
        if self.L.ref_colorspace(self.handle, name.encode(), ctypes.byref(t)) != 0:
            raise RuntimeError("TransformImageColorspace failed")
        return self

    def contrast_stretch(self, black, white):
        t = ctypes.c_double(0.0)
        if self.L.ref_contrast_stretch(self.handle, black, white, ctypes.byref(t)) != 0:
            raise RuntimeError("ContrastStretchImage failed")
        return self

    def sync(self):
        """The CPU reads the pixels: the cache downloads the device copy if it is the newer one."""
        if self.L.shim_image_sync(self.handle) != 0:
            raise RuntimeError("pixel cache sync failed")
        return self

    def touch(self):
        """The CPU writes the pixels: the device copy is dropped, the next operator uploads again."""
        if self.L.shim_image_touch(self.handle) != 0:
            raise RuntimeError("pixel cache touch failed")
        return self


def accelerated_calls(hdri=False):
    return load(hdri).GetMagickHipAcceleratedCalls()


def device_statistics(hdri=False):
    """[(operator calls, streams handed out)] per device the shim arbitrates over."""
    L = load(hdri)
    n = L.GetMagickHipDeviceStatistics(1 << 30, None, None)
    out = []
    for i in range(n):
        c, s = ctypes.c_size_t(0), ctypes.c_size_t(0)
        L.GetMagickHipDeviceStatistics(i, ctypes.byref(c), ctypes.byref(s))
        out.append((c.value, s.value))
    return out
