"""Worker of tests/test_magickcore_shim.py (a subprocess: MAGICKHIP_LOGICAL_DEVICES and
MAGICK_HIP_SPREAD_BYTES must be in the environment when the libraries start): ONE big host-resident
image through MagickCore's own MorphologyImage (Dilate Disk:15), BlurImage and EqualizeImage on the
shim build — the row bands of the frame go round every (logical) device — and the same operators on
the CPU MagickCore; prints one JSON line.

    python tests/helpers/shim_spread.py <edge> [hdri]
"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MAGICK_HIP_LIBRARY", os.path.join(ROOT, "imagemagick_amd", "lib", "libmagickhip.so"))

from oracle import ref  # noqa: E402  (test infrastructure: the compiled reference and the shim build)


def main():
    edge = int(sys.argv[1])
    hdri = len(sys.argv) > 2 and sys.argv[2] == "hdri"
    rng = np.random.default_rng(5)
    px = rng.integers(0, 65536, (edge, edge, 4), dtype=np.uint16)
    if hdri:
        px = px.astype(np.float32)
    lib = ref._load(hdri, True)
    lib.GetMagickHipAcceleratedCalls.restype = ctypes.c_size_t
    lib.GetMagickHipDeviceStatistics.restype = ctypes.c_size_t
    lib.GetMagickHipDeviceStatistics.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                                 ctypes.POINTER(ctypes.c_size_t)]
    ref.set_thread_limit(os.cpu_count() or 1, hdri)
    g = ref.RefImage(px, shim=True)
    c = ref.RefImage(px)
    report = {"mismatches": []}
    before = lib.GetMagickHipAcceleratedCalls()
    gd, cd = g.morphology("Dilate", 1, "Disk:15"), c.morphology("Dilate", 1, "Disk:15")
    if not np.array_equal(gd.numpy(), cd.numpy()):
        report["mismatches"].append("dilate")
    gb, cb = g.blur(0.0, 3.0), c.blur(0.0, 3.0)
    if not np.array_equal(gb.numpy(), cb.numpy()):
        report["mismatches"].append("blur")
    gb.equalize()
    cb.equalize()
    if not np.array_equal(gb.numpy(), cb.numpy()):
        report["mismatches"].append("equalize")
    report["accelerated"] = lib.GetMagickHipAcceleratedCalls() - before
    calls = []
    n = lib.GetMagickHipDeviceStatistics(1 << 30, None, None)
    for i in range(n):
        k, s = ctypes.c_size_t(0), ctypes.c_size_t(0)
        lib.GetMagickHipDeviceStatistics(i, ctypes.byref(k), ctypes.byref(s))
        calls.append(k.value)
    hip = ctypes.CDLL(os.environ["MAGICK_HIP_LIBRARY"])
    hip.MhBandedBands.restype = ctypes.c_ulonglong
    hip.MhBandedBands.argtypes = [ctypes.c_int]
    report.update(devices=n, calls=calls, bands=[int(hip.MhBandedBands(i)) for i in range(n)])
    print(json.dumps(report))


if __name__ == "__main__":
    main()
