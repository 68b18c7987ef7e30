#!/usr/bin/env python3
"""Child process of tests/test_magickcore_shim.py::test_default_mode_through_magickcore: a FRESH process — no
MAGICK_HIP_PRECISION / MAGICKHIP_* in the environment, no MhSetPrecision call — drives MagickCore-with-the-binding
the way an unchanged caller does and compares every result with the CPU MagickCore.  Prints one JSON object."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_pixels, ulp_diff_f32
from oracle import ref as refmod

for name in list(os.environ):
    if name.startswith("MAGICKHIP_") or name == "MAGICK_HIP_PRECISION":
        del os.environ[name]
os.environ["MAGICK_HIP_LIBRARY"] = os.path.join(ROOT, "imagemagick_amd", "lib", "libmagickhip.so")

report = {}


def levels(got, want):
    return int(np.abs(got.astype(np.int64) - want.astype(np.int64)).max())


def ulps(got, want):
    return int(ulp_diff_f32(got, want).max())


def calls(hdri):
    lib = refmod._load(hdri, True)
    lib.GetMagickHipAcceleratedCalls.restype = ctypes.c_size_t
    return lib.GetMagickHipAcceleratedCalls()


px = make_pixels(96, 120, 4, np.uint16, seed=21)
before = calls(False)
gpu, cpu = refmod.RefImage(px, shim=True), refmod.RefImage(px)
report["blur"] = levels(gpu.blur(0.0, 3.0).numpy(), cpu.blur(0.0, 3.0).numpy())
report["gaussian_blur"] = levels(gpu.gaussian_blur(0.0, 2.0).numpy(), cpu.gaussian_blur(0.0, 2.0).numpy())
report["unsharp"] = levels(gpu.unsharp(0.0, 2.0, 1.0, 0.02).numpy(), cpu.unsharp(0.0, 2.0, 1.0, 0.02).numpy())
report["unsharp_radius"] = levels(gpu.unsharp(25.0, 2.0, 1.0, 0.02).numpy(), cpu.unsharp(25.0, 2.0, 1.0, 0.02).numpy())
report["blur_radius"] = levels(gpu.blur(30.0, 2.0).numpy(), cpu.blur(30.0, 2.0).numpy())
report["resize_x4"] = levels(gpu.resize(480, 384, "Lanczos").numpy(), cpu.resize(480, 384, "Lanczos").numpy())
report["resize_div4"] = levels(gpu.resize(30, 24, "Lanczos").numpy(), cpu.resize(30, 24, "Lanczos").numpy())
kernel = "5x5: 1,2,3,2,1 2,4,6,4,2 3,6,9,6,3 2,4,6,4,2 1,2,3,2,1"
report["convolve"] = levels(gpu.set_artifact("convolve:scale", "!").convolve(kernel).numpy(),
                            cpu.set_artifact("convolve:scale", "!").convolve(kernel).numpy())
report["convolve_disk"] = levels(gpu.morphology("Convolve", 1, "Disk:4.3").numpy(),
                                 cpu.morphology("Convolve", 1, "Disk:4.3").numpy())
# sRGB -> Lab (FAST: f32, within one level), then ContrastStretch of THAT frame: integer counts and an fp64 map in
# both modes — bit-identical to the CPU path given the same Lab frame
big = make_pixels(1100, 1200, 4, np.uint16, seed=22)
g2, c2 = refmod.RefImage(big, shim=True), refmod.RefImage(big)
lab_gpu = g2.colorspace("Lab").numpy()
report["lab"] = levels(lab_gpu, c2.colorspace("Lab").numpy())
n = big.shape[0] * big.shape[1]
stretched = g2.contrast_stretch(0.02 * n, n - 0.01 * n).numpy()
same_frame = refmod.RefImage(lab_gpu, "Lab").contrast_stretch(0.02 * n, n - 0.01 * n).numpy()
report["contrast_stretch_of_that_lab_frame"] = levels(stretched, same_frame)
report["accelerated_calls_q16"] = calls(False) - before

fpx = make_pixels(96, 120, 4, np.float32, seed=23)
before = calls(True)
gf, cf = refmod.RefImage(fpx, shim=True), refmod.RefImage(fpx)
report["float_resize_x4"] = ulps(gf.resize(480, 384, "Lanczos").numpy(), cf.resize(480, 384, "Lanczos").numpy())
report["float_resize_div4"] = ulps(gf.resize(30, 24, "Lanczos").numpy(), cf.resize(30, 24, "Lanczos").numpy())
report["float_blur"] = ulps(gf.blur(0.0, 3.0).numpy(), cf.blur(0.0, 3.0).numpy())
report["float_unsharp"] = ulps(gf.unsharp(0.0, 2.0, 1.0, 0.02).numpy(), cf.unsharp(0.0, 2.0, 1.0, 0.02).numpy())
report["accelerated_calls_float"] = calls(True) - before

# what mode was that?  (asked LAST, through the library instance the shim loaded; nobody set it)
hip = ctypes.CDLL(os.environ["MAGICK_HIP_LIBRARY"])
hip.MhGetPrecision.restype = ctypes.c_int
report["precision"] = int(hip.MhGetPrecision())
print(json.dumps(report))
