"""TEST INFRASTRUCTURE — ctypes wrapper around the compiled reference oracle
(oracle/_ref/libmagickref_{q16,q16hdri}.so, built from /root/reference by
oracle/refbuild/Makefile).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this; the product never does.

Both quantum builds can be loaded in one process (their symbols are kept in
separate dlopen namespaces with RTLD_LOCAL).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def lib_path(hdri):
    return os.path.join(_HERE, "_ref", "libmagickref_%s.so" % ("q16hdri" if hdri else "q16"))


def available(hdri=False):
    return os.path.exists(lib_path(hdri))


def shim_lib_path(hdri):
    """MagickCore built with the HIP accelerate backend slotted in (shim/Makefile)."""
    return os.path.join(os.path.dirname(_HERE), "shim", "_build",
                        "libMagickCore-hip-%s.so" % ("q16hdri" if hdri else "q16"))


def _load(hdri, shim=False):
    key = (bool(hdri), bool(shim))
    if key in _LIBS:
        return _LIBS[key]
    path = shim_lib_path(hdri) if shim else lib_path(hdri)
    if not os.path.exists(path):
        raise RuntimeError("compiled reference oracle missing: %s (make -C oracle/refbuild)" % path)
    # the reference looks for its config XML here; nothing is installed (and /root/reference
    # does not exist on the GPU box), so every table falls back to its built-in defaults
    os.environ.setdefault("MAGICK_CONFIGURE_PATH", os.path.join(_HERE, "_ref", "config"))
    L = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    vp, sz, dbl, cp = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_double, ctypes.c_char_p
    pd = ctypes.POINTER(ctypes.c_double)
    L.ref_init.restype = ctypes.c_int
    L.ref_quantum_is_float.restype = ctypes.c_int
    L.ref_thread_limit.restype = ctypes.c_int
    L.ref_set_thread_limit.argtypes = [ctypes.c_int]
    L.ref_last_error.restype = cp
    L.ref_image_new.restype = vp
    L.ref_image_new.argtypes = [sz, sz, cp, cp, vp]
    L.ref_image_free.argtypes = [vp]
    L.ref_image_info.argtypes = [vp, ctypes.POINTER(sz), ctypes.POINTER(sz), ctypes.POINTER(sz),
                                 ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                 ctypes.POINTER(ctypes.c_int)]
    L.ref_colorspace_name.restype = cp
    L.ref_colorspace_name.argtypes = [ctypes.c_int]
    L.ref_image_get.argtypes = [vp, vp]
    L.ref_image_get_rows.argtypes = [vp, ctypes.c_ssize_t, ctypes.c_size_t, vp]
    L.ref_image_set_channel_mask.argtypes = [vp, cp]
    L.ref_image_set_artifact.argtypes = [vp, cp, cp]
    L.ref_image_get_property.restype = cp
    L.ref_image_get_property.argtypes = [vp, cp]
    for name, extra in [("ref_blur", [dbl, dbl]), ("ref_gaussian_blur", [dbl, dbl]),
                        ("ref_sharpen", [dbl, dbl]), ("ref_motion_blur", [dbl, dbl, dbl]), ("ref_rotational_blur", [dbl]), ("ref_local_contrast", [dbl, dbl]), ("ref_despeckle", []), ("ref_wavelet_denoise", [dbl, dbl]), ("ref_emboss", [dbl, dbl]), ("ref_edge", [dbl]),
                        ("ref_unsharp", [dbl, dbl, dbl, dbl]), ("ref_convolve", [cp]),
                        ("ref_morphology", [cp, ctypes.c_ssize_t, cp]),
                        ("ref_resize", [sz, sz, cp])]:
        fn = getattr(L, name)
        fn.restype = vp
        fn.argtypes = [vp] + extra + [pd]
    L.ref_contrast_stretch.argtypes = [vp, dbl, dbl, pd]
    L.ref_equalize.argtypes = [vp, pd]
    L.ref_colorspace.argtypes = [vp, cp, pd]
    L.ref_grayscale.argtypes = [vp, cp, pd]
    L.ref_import_pixels.argtypes = [vp, ctypes.c_ssize_t, ctypes.c_ssize_t, sz, sz, cp, ctypes.c_int, vp]
    L.ref_export_pixels.argtypes = [vp, ctypes.c_ssize_t, ctypes.c_ssize_t, sz, sz, cp, ctypes.c_int, vp]
    L.ref_contrast.argtypes = [vp, ctypes.c_int, pd]
    L.ref_modulate.argtypes = [vp, cp, pd]
    L.ref_function.argtypes = [vp, cp, sz, pd, pd]
    L.ref_kernel.argtypes = [cp, ctypes.c_int, ctypes.POINTER(sz), ctypes.POINTER(sz),
                             ctypes.POINTER(ctypes.c_ssize_t), ctypes.POINTER(ctypes.c_ssize_t),
                             vp, vp]
    L.ref_resize_filter_weights.argtypes = [vp, cp, vp, sz, vp, pd]
    L.ref_pixel_intensity.restype = dbl
    L.ref_pixel_intensity.argtypes = [vp, vp]
    L.ref_decode_gamma.restype = dbl
    L.ref_decode_gamma.argtypes = [dbl]
    L.ref_encode_gamma.restype = dbl
    L.ref_encode_gamma.argtypes = [dbl]
    L.ref_init()
    _LIBS[key] = L
    return L


_MAPS = {1: "GRAY", 2: "GRAYA", 3: "RGB", 4: "RGBA"}


# StorageType, MagickCore/pixel.h:146-156
STORAGE = {"uint8": 1, "float64": 2, "float32": 3, "uint32": 4, "uint64": 5, "uint16": 7}


class RefImage:
    """An image inside the reference's pixel cache."""

    def __init__(self, pixels=None, colorspace="sRGB", handle=None, lib=None, hdri=None, shim=False):
        if handle is not None:
            self.L, self.handle, self.hdri = lib, handle, hdri
            self.last_seconds = 0.0
            return
        pixels = np.ascontiguousarray(pixels)
        if pixels.ndim == 2:
            pixels = pixels[:, :, None]
        self.hdri = pixels.dtype == np.float32
        if not self.hdri and pixels.dtype != np.uint16:
            raise ValueError("uint16 or float32 pixels expected")
        self.L = _load(self.hdri, shim)
        rows, cols, ch = pixels.shape
        self.handle = self.L.ref_image_new(cols, rows, _MAPS[ch].encode(), colorspace.encode(),
                                           pixels.ctypes.data)
        if not self.handle:
            raise RuntimeError("ref_image_new failed: %s" % self.L.ref_last_error().decode())
        self.last_seconds = 0.0

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.L.ref_image_free(self.handle)
                self.handle = None
        except Exception:
            pass

    def info(self):
        c, r, ch = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
        cs, at, ty = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self.L.ref_image_info(self.handle, ctypes.byref(c), ctypes.byref(r), ctypes.byref(ch),
                              ctypes.byref(cs), ctypes.byref(at), ctypes.byref(ty))
        return {"columns": c.value, "rows": r.value, "channels": ch.value,
                "colorspace": self.L.ref_colorspace_name(cs.value).decode(),
                "alpha_trait": at.value, "type": ty.value}

    def numpy(self):
        i = self.info()
        out = np.empty((i["rows"], i["columns"], i["channels"]),
                       dtype=np.float32 if self.hdri else np.uint16)
        if self.L.ref_image_get(self.handle, out.ctypes.data) != 0:
            raise RuntimeError("ref_image_get failed")
        return out

    def numpy_rows(self, y0, rows, out=None):
        """Rows [y0, y0+rows) of the pixel cache (a 17 GB frame is compared a band at a time)."""
        i = self.info()
        if out is None:
            out = np.empty((rows, i["columns"], i["channels"]), dtype=np.float32 if self.hdri else np.uint16)
        if self.L.ref_image_get_rows(self.handle, y0, rows, out.ctypes.data) != 0:
            raise RuntimeError("ref_image_get_rows failed")
        return out

    def set_channel_mask(self, channels):
        self.L.ref_image_set_channel_mask(self.handle, channels.encode())
        return self

    def set_artifact(self, key, value):
        self.L.ref_image_set_artifact(self.handle, key.encode(),
                                      None if value is None else value.encode())
        return self

    def _new(self, fn, *args):
        t = ctypes.c_double(0.0)
        h = fn(self.handle, *args, ctypes.byref(t))
        if not h:
            raise RuntimeError("reference operator failed: %s" % self.L.ref_last_error().decode())
        out = RefImage(handle=h, lib=self.L, hdri=self.hdri)
        out.last_seconds = t.value
        return out

    def _inplace(self, fn, *args):
        t = ctypes.c_double(0.0)
        if fn(self.handle, *args, ctypes.byref(t)) != 0:
            raise RuntimeError("reference operator failed: %s" % self.L.ref_last_error().decode())
        self.last_seconds = t.value
        return self

    def blur(self, radius, sigma):
        return self._new(self.L.ref_blur, radius, sigma)

    def gaussian_blur(self, radius, sigma):
        return self._new(self.L.ref_gaussian_blur, radius, sigma)

    def wavelet_denoise(self, threshold, softness=0.0):
        return self._new(self.L.ref_wavelet_denoise, threshold, softness)

    def despeckle(self):
        return self._new(self.L.ref_despeckle)

    def local_contrast(self, radius, strength):
        return self._new(self.L.ref_local_contrast, radius, strength)

    def rotational_blur(self, angle):
        return self._new(self.L.ref_rotational_blur, angle)

    def motion_blur(self, radius, sigma, angle):
        return self._new(self.L.ref_motion_blur, radius, sigma, angle)

    def sharpen(self, radius, sigma):
        return self._new(self.L.ref_sharpen, radius, sigma)

    def emboss(self, radius, sigma):
        return self._new(self.L.ref_emboss, radius, sigma)

    def edge(self, radius):
        return self._new(self.L.ref_edge, radius)

    def unsharp(self, radius, sigma, gain, threshold):
        return self._new(self.L.ref_unsharp, radius, sigma, gain, threshold)

    def convolve(self, kernel):
        return self._new(self.L.ref_convolve, kernel.encode())

    def morphology(self, method, iterations, kernel):
        return self._new(self.L.ref_morphology, method.encode(), iterations, kernel.encode())

    def resize(self, columns, rows, filter="Lanczos"):
        return self._new(self.L.ref_resize, columns, rows, filter.encode())

    def contrast_stretch(self, black, white):
        return self._inplace(self.L.ref_contrast_stretch, black, white)

    def equalize(self):
        return self._inplace(self.L.ref_equalize)

    def colorspace(self, name):
        return self._inplace(self.L.ref_colorspace, name.encode())

    def grayscale(self, method="Rec709Luma"):
        return self._inplace(self.L.ref_grayscale, method.encode())

    def import_pixels(self, x, y, map, data):
        """ImportImagePixels: data is [height, width, len(map)] of uint8/16/32/64, float32/64."""
        data = np.ascontiguousarray(data)
        if self.L.ref_import_pixels(self.handle, x, y, data.shape[1], data.shape[0], map.encode(),
                                    STORAGE[data.dtype.name], data.ctypes.data) != 0:
            raise RuntimeError("ImportImagePixels failed: %s" % self.L.ref_last_error().decode())
        return self

    def export_pixels(self, x, y, width, height, map, dtype, out=None):
        """ExportImagePixels into a new (or the given) [height, width, len(map)] array."""
        out = np.zeros((height, width, len(map)), dtype=dtype) if out is None else out
        if self.L.ref_export_pixels(self.handle, x, y, width, height, map.encode(),
                                    STORAGE[np.dtype(dtype).name], out.ctypes.data) != 0:
            raise RuntimeError("ExportImagePixels failed: %s" % self.L.ref_last_error().decode())
        return out

    def contrast(self, sharpen=True):
        return self._inplace(self.L.ref_contrast, 1 if sharpen else 0)

    def modulate(self, brightness=100.0, saturation=100.0, hue=100.0, colorspace=None):
        """ModulateImage("brightness,saturation,hue"); colorspace = the modulate:colorspace artifact."""
        if colorspace is not None:
            self.L.ref_image_set_artifact(self.handle, b"modulate:colorspace", colorspace.encode())
        return self._inplace(self.L.ref_modulate, ("%.17g,%.17g,%.17g" % (brightness, saturation, hue)).encode())

    def function(self, function, parameters):
        params = (ctypes.c_double * max(1, len(parameters)))(*parameters)
        return self._inplace(self.L.ref_function, function.encode(), len(parameters), params)

    def intensity(self, pixel):
        px = np.ascontiguousarray(pixel, dtype=np.float32 if self.hdri else np.uint16)
        return self.L.ref_pixel_intensity(self.handle, px.ctypes.data)

    def filter_weights(self, filter, xs):
        xs = np.ascontiguousarray(xs, dtype=np.float64)
        w = np.empty_like(xs)
        support = ctypes.c_double(0.0)
        if self.L.ref_resize_filter_weights(self.handle, filter.encode(), xs.ctypes.data, xs.size,
                                            w.ctypes.data, ctypes.byref(support)) != 0:
            raise RuntimeError("filter %s rejected" % filter)
        return w, support.value


def kernel(kernel_string, index=0, hdri=False):
    """(values[h,w], x, y, count) of kernel `index` as the reference builds it."""
    L = _load(hdri)
    w, h = ctypes.c_size_t(), ctypes.c_size_t()
    x, y = ctypes.c_ssize_t(), ctypes.c_ssize_t()
    n = L.ref_kernel(kernel_string.encode(), index, ctypes.byref(w), ctypes.byref(h),
                     ctypes.byref(x), ctypes.byref(y), None, None)
    if n < 0:
        return None
    values = np.empty(w.value * h.value, dtype=np.float64)
    L.ref_kernel(kernel_string.encode(), index, ctypes.byref(w), ctypes.byref(h), ctypes.byref(x),
                 ctypes.byref(y), values.ctypes.data, None)
    return values.reshape(h.value, w.value), x.value, y.value, n


def thread_limit(hdri=False):
    return _load(hdri).ref_thread_limit()


def set_thread_limit(n, hdri=False):
    _load(hdri).ref_set_thread_limit(n)


def decode_gamma(v, hdri=False):
    return _load(hdri).ref_decode_gamma(v)


def encode_gamma(v, hdri=False):
    return _load(hdri).ref_encode_gamma(v)
