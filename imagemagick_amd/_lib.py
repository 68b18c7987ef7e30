"""ctypes binding of libmagickhip.so (include/magickhip.h).

Only a loader and struct/prototype declarations live here: the product is the
shared library.  Nothing in this package falls back to a CPU implementation —
if the library is missing or no HIP device is usable, calls raise
``MagickHipError``.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# MAGICKHIP_LIBRARY: load another build of the library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("MAGICKHIP_LIBRARY") or os.path.join(_HERE, "lib", "libmagickhip.so")

MH_MAX_CHANNELS = 4
MH_OK = 0
STATUS_NAMES = {0: "MH_OK", 1: "MH_UNSUPPORTED", 2: "MH_NO_DEVICE", 3: "MH_BAD_ARGUMENT",
                4: "MH_OUT_OF_MEMORY", 5: "MH_DEVICE_ERROR", 6: "MH_DISABLED"}

QUANTUM_U16, QUANTUM_F32 = 0, 1
MEMORY_HOST, MEMORY_DEVICE = 0, 1
TRAIT_UNDEFINED, TRAIT_COPY, TRAIT_UPDATE, TRAIT_BLEND = 0, 1, 2, 4
PRECISION_EXACT, PRECISION_FAST = 0, 1
ALL_CHANNELS = 0x7FFFFFF
SYNC_CHANNELS = 0x20000

COLORSPACES = {"undefined": 0, "cmy": 1, "gray": 3, "hcl": 4, "hclp": 5, "hsb": 6, "hsi": 7, "hsl": 8,
               "hsv": 9, "hwb": 10, "lab": 11, "lch": 12, "lchab": 13, "lchuv": 14, "log": 15, "lms": 16, "luv": 17,
               "ohta": 18, "rec601ycbcr": 19, "rec709ycbcr": 20, "rgb": 21, "scrgb": 22, "srgb": 23, "xyy": 25, "xyz": 26, "ycbcr": 27, "ycc": 28, "ydbdr": 29, "yiq": 30,
               "ypbpr": 31, "yuv": 32, "lineargray": 33, "jzazbz": 34, "displayp3": 35, "adobe98": 36,
               "prophoto": 37, "oklab": 38, "oklch": 39, "cat02lms": 40}

MORPHOLOGY = {name: i for i, name in enumerate([
    "undefined", "convolve", "correlate", "erode", "dilate", "erodeintensity",
    "dilateintensity", "iterativedistance", "open", "close", "openintensity",
    "closeintensity", "smooth", "edgein", "edgeout", "edge", "tophat", "bottomhat",
    "hitandmiss", "thinning", "thicken", "distance", "voronoi"])}

FILTERS = {name: i for i, name in enumerate([
    "undefined", "point", "box", "triangle", "hermite", "hann", "hamming", "blackman",
    "gaussian", "quadratic", "cubic", "catrom", "mitchell", "jinc", "sinc", "sincfast",
    "kaiser", "welch", "parzen", "bohman", "bartlett", "lagrange", "lanczos",
    "lanczossharp", "lanczos2", "lanczos2sharp", "robidoux", "robidouxsharp", "cosine",
    "spline", "lanczosradius", "cubicspline", "magickernelsharp2013",
    "magickernelsharp2021"])}


INTENSITY = {"undefined": 0, "average": 1, "brightness": 2, "lightness": 3, "ms": 4, "rec601luma": 5,
             "rec601luminance": 6, "rec709luma": 7, "rec709luminance": 8, "rms": 9}
FUNCTIONS = {"arcsin": 1, "arctan": 2, "polynomial": 3, "sinusoid": 4}


class MagickHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), message))
        self.status = status


class MhImage(ctypes.Structure):
    _fields_ = [
        ("pixels", ctypes.c_void_p),
        ("columns", ctypes.c_size_t),
        ("rows", ctypes.c_size_t),
        ("number_channels", ctypes.c_uint32),
        ("quantum", ctypes.c_uint32),
        ("memory", ctypes.c_uint32),
        ("device", ctypes.c_int32),
        ("channel_traits", ctypes.c_uint32 * MH_MAX_CHANNELS),
        ("alpha_offset", ctypes.c_int32),
        ("alpha_trait", ctypes.c_uint32),
        ("colorspace", ctypes.c_uint32),
        ("intensity", ctypes.c_uint32),
        ("channel_mask", ctypes.c_uint32),
        ("stream", ctypes.c_void_p),
        ("precision", ctypes.c_uint32),      # 0 = library default, else MhPrecision + 1
    ]


class MhDeviceInfo(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 128), ("architecture", ctypes.c_char * 64),
                ("compute_units", ctypes.c_int32), ("clock_mhz", ctypes.c_int32),
                ("global_memory", ctypes.c_uint64), ("local_memory", ctypes.c_uint64)]


class MhKernelInfo(ctypes.Structure):
    pass


MhKernelInfo._fields_ = [
    ("type", ctypes.c_int),
    ("width", ctypes.c_size_t),
    ("height", ctypes.c_size_t),
    ("x", ctypes.c_ssize_t),
    ("y", ctypes.c_ssize_t),
    ("values", ctypes.POINTER(ctypes.c_double)),
    ("minimum", ctypes.c_double),
    ("maximum", ctypes.c_double),
    ("negative_range", ctypes.c_double),
    ("positive_range", ctypes.c_double),
    ("angle", ctypes.c_double),
    ("next", ctypes.POINTER(MhKernelInfo)),
]


class MhKernelProfileRecord(ctypes.Structure):
    _fields_ = [("kernel_name", ctypes.c_char_p), ("count", ctypes.c_ulong),
                ("min_ms", ctypes.c_double), ("max_ms", ctypes.c_double),
                ("total_ms", ctypes.c_double)]


class MhOperator(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("args", ctypes.c_double * 4), ("text", ctypes.c_char_p)]


class MhBatchReport(ctypes.Structure):
    _fields_ = [("devices", ctypes.c_uint32), ("workers", ctypes.c_uint32), ("used_rccl", ctypes.c_uint32),
                ("halo_exchanges", ctypes.c_uint32), ("images_per_device", ctypes.c_uint64 * 16),
                ("seconds", ctypes.c_double)]


OPERATORS = {"blur": 1, "gaussianblur": 2, "unsharpmask": 3, "resize": 4, "morphology": 5,
             "colorspace": 6, "contraststretch": 7, "equalize": 8}


# every symbol include/magickhip.h declares: (name, restype, argtypes)
_P = ctypes.POINTER
PROTOTYPES = [
    ("MhInitImage", None, [_P(MhImage), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                           ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    ("MhInitialize", ctypes.c_int, []),
    ("MhTerminus", None, []),
    ("MhDeviceCount", ctypes.c_int, []),
    ("MhSetDevice", ctypes.c_int, [ctypes.c_int]),
    ("MhGetEnabled", ctypes.c_int, []),
    ("MhSetEnabled", ctypes.c_int, [ctypes.c_int]),
    ("MhGetLastError", ctypes.c_char_p, []),
    ("MhGetVersion", ctypes.c_char_p, []),
    ("MhGetPrecision", ctypes.c_int, []),
    ("MhSetPrecision", ctypes.c_int, [ctypes.c_int]),
    ("MhSetOption", ctypes.c_int, [ctypes.c_char_p, ctypes.c_char_p]),
    ("MhGetOption", ctypes.c_char_p, [ctypes.c_char_p]),
    ("MhLogicalDeviceCount", ctypes.c_int, []),
    ("MhGetDeviceInfo", ctypes.c_int, [ctypes.c_int, _P(MhDeviceInfo)]),
    ("MhStreamCreate", ctypes.c_int, [ctypes.c_int, _P(ctypes.c_void_p)]),
    ("MhStreamDestroy", ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    ("MhDeviceAllocAsync", ctypes.c_int, [ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, _P(ctypes.c_void_p)]),
    ("MhDeviceFreeAsync", ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    ("MhGetDeviceProfileRecords", ctypes.c_size_t, [ctypes.c_int, _P(MhKernelProfileRecord), ctypes.c_size_t]),
    ("MhDeviceAlloc", ctypes.c_int, [ctypes.c_int, ctypes.c_size_t, _P(ctypes.c_void_p)]),
    ("MhDeviceFree", ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    ("MhUpload", ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                ctypes.c_void_p]),
    ("MhDownload", ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_size_t, ctypes.c_void_p]),
    ("MhSynchronize", ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    ("MhSetProfileEnabled", ctypes.c_int, [ctypes.c_int]),
    ("MhGetProfileRecords", ctypes.c_size_t, [_P(MhKernelProfileRecord), ctypes.c_size_t]),
    ("MhResetProfileRecords", None, []),
    ("MhExactBlurRecomputed", ctypes.c_ulonglong, [ctypes.c_int]),
    ("MhSeparableRecomputed", ctypes.c_ulonglong, [ctypes.c_int]),
    ("MhConvolve2DRecomputed", ctypes.c_ulonglong, [ctypes.c_int]),
    ("MhConvolve2DTieRecomputed", ctypes.c_ulonglong, [ctypes.c_int]),
    ("MhKernelIntegerCells", ctypes.c_int, [_P(MhKernelInfo), _P(ctypes.c_int), _P(ctypes.c_double)]),
    ("MhHostAlloc", ctypes.c_void_p, [ctypes.c_size_t]),
    ("MhHostFree", ctypes.c_int, [ctypes.c_void_p]),
    ("MhHostAllocatedBytes", ctypes.c_size_t, []),
    ("MhHostPinnedBytes", ctypes.c_size_t, []),
    ("MhBandedBands", ctypes.c_ulonglong, [ctypes.c_int]),
    ("MhAcquireKernelInfo", _P(MhKernelInfo), [ctypes.c_char_p]),
    ("MhDestroyKernelInfo", _P(MhKernelInfo), [_P(MhKernelInfo)]),
    ("MhCloneKernelInfo", _P(MhKernelInfo), [_P(MhKernelInfo)]),
    ("MhScaleKernelInfo", None, [_P(MhKernelInfo), ctypes.c_double, ctypes.c_uint]),
    ("MhGetOptimalKernelWidth1D", ctypes.c_size_t, [ctypes.c_double, ctypes.c_double]),
    ("MhKernelOuterProductFactors", ctypes.c_int, [_P(MhKernelInfo), _P(ctypes.c_double), _P(ctypes.c_double)]),
    ("MhKernelOuterProductPlusDelta", ctypes.c_int, [_P(MhKernelInfo), _P(ctypes.c_double), _P(ctypes.c_double),
                                                   _P(ctypes.c_double)]),
    ("MhGetOptimalKernelWidth2D", ctypes.c_size_t, [ctypes.c_double, ctypes.c_double]),
    ("MhAcquireResizeFilter", ctypes.c_void_p, [ctypes.c_int, ctypes.c_int]),
    ("MhAcquireResizeFilterFromCallback", ctypes.c_void_p,
     [ctypes.CFUNCTYPE(ctypes.c_double, ctypes.c_void_p, ctypes.c_double), ctypes.c_void_p,
      ctypes.c_double]),
    ("MhDestroyResizeFilter", ctypes.c_void_p, [ctypes.c_void_p]),
    ("MhGetResizeFilterWeight", ctypes.c_double, [ctypes.c_void_p, ctypes.c_double]),
    ("MhGetResizeFilterSupport", ctypes.c_double, [ctypes.c_void_p]),
    ("MagickHipBlurImage", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_double,
                                          ctypes.c_double]),
    ("MagickHipConvolveImage", ctypes.c_int, [_P(MhImage), _P(MhImage), _P(MhKernelInfo)]),
    ("MagickHipMorphologyImage", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_int,
                                                ctypes.c_ssize_t, _P(MhKernelInfo),
                                                ctypes.c_double]),
    ("MagickHipMorphologyImageCompose", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_int,
                                                       ctypes.c_ssize_t, _P(MhKernelInfo),
                                                       ctypes.c_double, ctypes.c_int]),
    ("MagickHipMorphologyPrimitive", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_int,
                                                    _P(MhKernelInfo), ctypes.c_double,
                                                    _P(ctypes.c_ssize_t)]),
    ("MagickHipWaveletDenoiseImage", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_double, ctypes.c_double]),
    ("MagickHipDespeckleImage", ctypes.c_int, [_P(MhImage), _P(MhImage)]),
    ("MagickHipLocalContrastImage", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_double, ctypes.c_double]),
    ("MagickHipRotationalBlurImage", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_double]),
    ("MagickHipMotionBlurImage", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_double, ctypes.c_double,
                                                ctypes.c_double]),
    ("MagickHipMotionBlurImageWithKernel", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.POINTER(ctypes.c_double),
                                                          ctypes.c_size_t, ctypes.POINTER(ctypes.c_ssize_t)]),
    ("MagickHipGaussianBlurImage", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_double, ctypes.c_double]),
    ("MagickHipSharpenImage", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_double, ctypes.c_double]),
    ("MagickHipEdgeImage", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_double]),
    ("MagickHipEmbossImage", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_double, ctypes.c_double]),
    ("MagickHipUnsharpMaskImage", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_double,
                                                 ctypes.c_double, ctypes.c_double,
                                                 ctypes.c_double]),
    ("MagickHipResizeImage", ctypes.c_int, [_P(MhImage), _P(MhImage), ctypes.c_int]),
    ("MagickHipResizeImageWithFilter", ctypes.c_int, [_P(MhImage), _P(MhImage),
                                                      ctypes.c_void_p]),
    ("MagickHipContrastStretchImage", ctypes.c_int, [_P(MhImage), ctypes.c_double,
                                                     ctypes.c_double, _P(ctypes.c_int)]),
    ("MagickHipEqualizeImage", ctypes.c_int, [_P(MhImage)]),
    ("MagickHipTransformImageColorspace", ctypes.c_int, [_P(MhImage), ctypes.c_int]),
    ("MagickHipGrayscaleImage", ctypes.c_int, [_P(MhImage), ctypes.c_int]),
    ("MagickHipImportImagePixels", ctypes.c_int, [_P(MhImage), ctypes.c_ssize_t, ctypes.c_ssize_t, ctypes.c_size_t,
                                                  ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_int]),
    ("MagickHipExportImagePixels", ctypes.c_int, [_P(MhImage), ctypes.c_ssize_t, ctypes.c_ssize_t, ctypes.c_size_t,
                                                  ctypes.c_size_t, ctypes.c_char_p, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_int]),
    ("MagickHipContrastImage", ctypes.c_int, [_P(MhImage), ctypes.c_int]),
    ("MagickHipModulateImage", ctypes.c_int, [_P(MhImage), ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                              ctypes.c_int]),
    ("MagickHipFunctionImage", ctypes.c_int, [_P(MhImage), ctypes.c_int, ctypes.c_size_t,
                                              _P(ctypes.c_double)]),
    ("MagickHipHistogram", ctypes.c_int, [_P(MhImage), ctypes.c_int, ctypes.c_void_p]),
    ("MhContrastStretchLUT", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t,
                                            ctypes.c_size_t, ctypes.c_double, ctypes.c_double,
                                            ctypes.c_int, ctypes.c_void_p,
                                            _P(ctypes.c_uint32)]),
    ("MhEqualizeLUT", ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int,
                                     ctypes.c_void_p, _P(ctypes.c_uint32)]),
    ("MagickHipApplyLUT", ctypes.c_int, [_P(MhImage), ctypes.c_void_p, ctypes.c_uint32]),
    ("MagickHipIsImageGray", ctypes.c_int, [_P(MhImage), _P(ctypes.c_int)]),
    ("MagickHipApplyHistogram", ctypes.c_int, [_P(MhImage), ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_double, ctypes.c_double, ctypes.c_size_t]),
    ("MagickHipTransformColorspaceContrastStretchImage", ctypes.c_int,
     [_P(MhImage), ctypes.c_int, ctypes.c_double, ctypes.c_double]),
    ("MagickHipBatchImages", ctypes.c_int, [_P(MhOperator), ctypes.c_size_t, _P(MhImage), _P(MhImage),
                                            ctypes.c_size_t, ctypes.c_int, ctypes.c_int, _P(MhBatchReport)]),
    ("MagickHipShardedImage", ctypes.c_int, [_P(MhOperator), ctypes.c_size_t, _P(MhImage), _P(MhImage),
                                             ctypes.c_int, _P(MhBatchReport)]),
]

_lib = None


def build(verbose=False):
    """Compile libmagickhip.so for gfx950 in-tree (hipcc cross-compiles on CPU-only hosts)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j%d" % max(1, os.cpu_count() or 1)]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout)
    if out.returncode != 0:
        raise RuntimeError("building libmagickhip.so failed")
    return LIB_PATH


def load():
    """dlopen libmagickhip.so and declare every prototype.  Importing torch first
    makes the process share torch's copy of the HIP runtime (same soname)."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  (loads libamdhip64 before our DT_NEEDED resolves)
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise MagickHipError(5, "libmagickhip.so is not built (%s); run "
                             "`python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, restype, argtypes in PROTOTYPES:
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(status):
    if status != MH_OK:
        lib = load()
        raise MagickHipError(status, lib.MhGetLastError().decode("utf-8", "replace"))
