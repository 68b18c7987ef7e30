#!/usr/bin/env python3
"""Generates the committed golden fixtures of tests/golden/.  Run in the build
container only (needs /root/reference and the compiled reference oracle/_ref):

    make -C oracle/refbuild -j8 && python tests/golden/make_golden.py

Outputs
  perlmagick_filter.npz   the reference's own tolerance goldens: PerlMagick/t/input.miff
                          and PerlMagick/t/reference/filter/{Blur,Convolve,Equalize,Resize,
                          UnsharpMask}.miff decoded to arrays (8-bit RGB, uncompressed MIFF),
                          with the (mean, maximum) error bounds filter.t states for each.
  reference_vectors.npz   seeded inputs and the outputs of the reference's own CPU
                          implementation (oracle/_ref = MagickCore compiled from
                          /root/reference) for every operator on the hot path, Q16 and
                          Q16-HDRI; plus host-side tables (blur taps, resize filter weights).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REFERENCE = "/root/reference"


def read_miff(path):
    """Uncompressed DirectClass depth-8 RGB MIFF (coders/miff.c): a text header
    terminated by ':\\x1a', then rows*columns*3 bytes."""
    data = open(path, "rb").read()
    end = data.index(b":\x1a")
    header = data[:end].decode("latin-1")
    fields = {}
    for token in header.replace("\n", " ").split():
        if "=" in token:
            k, v = token.split("=", 1)
            fields[k] = v
    cols, rows, depth = int(fields["columns"]), int(fields["rows"]), int(fields["depth"])
    assert depth == 8 and fields.get("compression", "None") in ("None", "Undefined"), fields
    body = data[end + 2:]
    assert len(body) == rows * cols * 3, (len(body), rows, cols)
    return np.frombuffer(body, dtype=np.uint8).reshape(rows, cols, 3).copy()


def perlmagick():
    t = os.path.join(REFERENCE, "PerlMagick", "t")
    out = {"input": read_miff(os.path.join(t, "input.miff"))}
    # (method, arguments) and (normalized_mean_error_max, normalized_maximum_error_max): filter.t
    bounds = {"Blur": (0.007, 0.7),          # filter.t:39   Blur('5x2')
              "Convolve": (0.1, 0.7),        # filter.t:63   Convolve([.0625 x4, .5, .0625 x4])
              "Equalize": (0.06, 0.5),       # filter.t:84
              "Resize": (0.00007, 0.07),     # filter.t:156  Resize('60%')
              "UnsharpMask": (0.004, 0.4)}   # filter.t:201  UnsharpMask('5x2+1')
    for name, b in bounds.items():
        out[name] = read_miff(os.path.join(t, "reference", "filter", name + ".miff"))
        out[name + "_bounds"] = np.array(b)
    np.savez_compressed(os.path.join(HERE, "perlmagick_filter.npz"), **out)
    print("perlmagick_filter.npz:", {k: v.shape for k, v in out.items()})


IO_TYPES = [("uint8", np.uint8), ("uint16", np.uint16), ("uint32", np.uint32), ("uint64", np.uint64),
            ("float32", np.float32), ("float64", np.float64)]
KERNEL_LISTS = ["Edges", "Corners", "Diagonals", "Diagonals:1,45", "Diagonals:2", "LineEnds", "LineEnds:3>",
                "LineEnds:4,90", "LineJunctions", "LineJunctions:3@", "LineJunctions:4", "LineJunctions:5",
                "Ridges", "Ridges:2", "ConvexHull", "Skeleton", "Skeleton:2", "Skeleton:3", "ThinSE:41",
                "ThinSE:87x90", "ThinSE:481,180", "ThinSE:423", "FreiChen", "FreiChen:2", "FreiChen:10",
                "FreiChen:13", "FreiChen:11,90", "FreiChen:45", "Laplacian:5", "Laplacian:7", "Laplacian:15",
                "Laplacian:19", "Sobel:>", "Sobel:@", "Kirsch:@", "Compass:90",
                "3x3: 1,2,3 4,5,6 7,8,9 ; 5x1: 1,2,3,2,1", "3x3>: 1,2,3 4,5,6 7,8,9", "3x3@: 0,1,- 0,1,1 -,1,-",
                "3x3<: 0,1,- 0,1,1 -,1,-", "3x1>: 0,1,0"]


def make_pixels(rng, rows, cols, ch, hdri):
    a = rng.integers(0, 65536, (rows, cols, ch), dtype=np.uint16)
    if not hdri:
        return a
    f = a.astype(np.float32) + rng.random((rows, cols, ch), dtype=np.float32)
    return np.minimum(f, np.float32(65535.0))


def reference_vectors():
    from oracle import ref
    out = {}
    rng = np.random.default_rng(20250222)
    for hdri in (False, True):
        tag = "hdri" if hdri else "q16"
        for ch in (1, 3, 4):
            px = make_pixels(rng, 26, 37, ch, hdri)
            key = "%s_c%d" % (tag, ch)
            out[key + "_in"] = px
            out[key + "_blur_0x2"] = ref.RefImage(px).blur(0.0, 2.0).numpy()
            out[key + "_blur_0x10"] = ref.RefImage(px).blur(0.0, 10.0).numpy()
            out[key + "_blur_3x1.5"] = ref.RefImage(px).blur(3.0, 1.5).numpy()
            out[key + "_dilate_disk4"] = ref.RefImage(px).morphology("Dilate", 1, "Disk:4").numpy()
            out[key + "_erode_disk4"] = ref.RefImage(px).morphology("Erode", 1, "Disk:4").numpy()
            for m, k in (("EdgeIn", "Disk:2.5"), ("EdgeOut", "Disk:2.5"), ("Edge", "Disk:2.5"),
                         ("TopHat", "Disk:2.5"), ("BottomHat", "Disk:2.5"), ("Smooth", "Disk:2.5")):
                out[key + "_" + m.lower() + "_disk2.5"] = ref.RefImage(px).morphology(m, 1, k).numpy()
            out[key + "_edge_disk2.5_x2"] = ref.RefImage(px).morphology("Edge", 2, "Disk:2.5").numpy()
            out[key + "_convolve_3x3nan"] = ref.RefImage(px).convolve("3x3: 1,-,1 2,4,2 1,nan,3").numpy()
            out[key + "_resize_lanczos_up"] = ref.RefImage(px).resize(101, 75, "Lanczos").numpy()
            out[key + "_resize_lanczos_down"] = ref.RefImage(px).resize(17, 11, "Lanczos").numpy()
            out[key + "_resize_mitchell"] = ref.RefImage(px).resize(60, 20, "Mitchell").numpy()
            out[key + "_resize_catrom"] = ref.RefImage(px).resize(20, 50, "Catrom").numpy()
            out[key + "_resize_triangle"] = ref.RefImage(px).resize(64, 64, "Triangle").numpy()
            out[key + "_despeckle"] = ref.RefImage(px).despeckle().numpy()
            out[key + "_localcontrast_60x40"] = ref.RefImage(px).local_contrast(60.0, 40.0).numpy()
            out[key + "_localcontrast_30x-25"] = ref.RefImage(px).local_contrast(30.0, -25.0).numpy()
            out[key + "_rotational_12"] = ref.RefImage(px).rotational_blur(12.0).numpy()
            out[key + "_rotational_-40"] = ref.RefImage(px).rotational_blur(-40.0).numpy()
            out[key + "_motion_0x3+30"] = ref.RefImage(px).motion_blur(0.0, 3.0, 30.0).numpy()
            out[key + "_motion_0x1.5-110"] = ref.RefImage(px).motion_blur(0.0, 1.5, -110.0).numpy()
            out[key + "_motion_4x2+90"] = ref.RefImage(px).motion_blur(4.0, 2.0, 90.0).numpy()
            out[key + "_unsharp"] = ref.RefImage(px).unsharp(0.0, 2.0, 1.0, 0.02).numpy()
            n = px.shape[0] * px.shape[1]
            out[key + "_cstretch"] = ref.RefImage(px).contrast_stretch(0.02 * n, n - 0.01 * n).numpy()
            out[key + "_equalize"] = ref.RefImage(px).equalize().numpy()
            for fn, params in (("Polynomial", (0.3, -1.2, 1.5, 0.1)), ("Sinusoid", (3.0, 90.0, 0.4, 0.5)),
                               ("Arcsin", (0.8, 0.45, 1.0, 0.5)), ("Arctan", (4.0, 0.5, 1.0, 0.5))):
                out[key + "_function_" + fn] = ref.RefImage(px).function(fn, params).numpy()
            if ch >= 3:
                out[key + "_contrast_sharpen"] = ref.RefImage(px).contrast(True).numpy()
                out[key + "_contrast_dull"] = ref.RefImage(px).contrast(False).numpy()
                out[key + "_modulate_hsl"] = ref.RefImage(px).modulate(110.0, 80.0, 135.0).numpy()
                out[key + "_modulate_hsl_dim"] = ref.RefImage(px).modulate(60.0, 150.0, 20.0).numpy()
                out[key + "_modulate_hsb"] = ref.RefImage(px).modulate(120.0, 70.0, 160.0, "HSB").numpy()
            if ch >= 3:
                for m in ("Rec709Luma", "Rec601Luma", "Rec709Luminance", "Average", "Brightness", "Lightness",
                          "MS", "RMS"):
                    out[key + "_gray_" + m] = ref.RefImage(px).grayscale(m).numpy()
                out[key + "_gray_linear_Rec709Luma"] = ref.RefImage(px, "RGB").grayscale("Rec709Luma").numpy()
            if ch >= 3:
                for a, b in (("sRGB", "RGB"), ("RGB", "sRGB"), ("sRGB", "Lab"), ("Lab", "sRGB"),
                             ("sRGB", "XYZ"), ("XYZ", "sRGB")):
                    out["%s_%s_to_%s" % (key, a, b)] = ref.RefImage(px, a).colorspace(b).numpy()
        # a smooth (low-entropy) frame for the histogram operators
        y, x = np.mgrid[0:40, 0:48]
        smooth = np.clip((x * 900.0 + y * 500.0)[:, :, None] * np.array([0.6, 0.7, 0.8, 0.9]) +
                         rng.integers(0, 300, (40, 48, 4)), 0, 65535).astype(np.uint16)
        smooth = smooth.astype(np.float32) if hdri else smooth
        out[tag + "_smooth_in"] = smooth
        n = 40 * 48
        out[tag + "_smooth_cstretch"] = ref.RefImage(smooth).contrast_stretch(0.02 * n, n - 0.01 * n).numpy()
        noisy = make_pixels(rng, 45, 50, 4, hdri)
        out[tag + "_wavelet_in"] = noisy
        out[tag + "_wavelet_5000x0"] = ref.RefImage(noisy).wavelet_denoise(5000.0, 0.0).numpy()
        out[tag + "_wavelet_9000x0.4"] = ref.RefImage(noisy).wavelet_denoise(9000.0, 0.4).numpy()
        out[tag + "_smooth_wavelet_800x0.2"] = ref.RefImage(smooth).wavelet_denoise(800.0, 0.2).numpy()
        out[tag + "_smooth_despeckle"] = ref.RefImage(smooth).despeckle().numpy()
        out[tag + "_smooth_equalize"] = ref.RefImage(smooth).equalize().numpy()
        out[tag + "_smooth_lab_cstretch"] = ref.RefImage(smooth).colorspace("Lab").contrast_stretch(
            0.02 * n, n - 0.01 * n).numpy()
    # host-side tables
    for s in ("blur:0x2", "blur:0x10", "blur:0x0.5", "blur:4x1.5", "blur:0x10+90", "Disk:15", "Disk:2.5",
              "Gaussian:0x1.5", "3x3: 1,-,1 2,4,2 1,nan,3"):
        values, x, y, _ = ref.kernel(s)
        out["kernel|" + s] = values
        out["kernel_origin|" + s] = np.array([x, y])
    # ImportImagePixels / ExportImagePixels: every storage type, several maps, a sub-region
    io = np.random.default_rng(77)
    for hdri in (False, True):
        tag = "hdri" if hdri else "q16"
        base = make_pixels(io, 19, 23, 4, hdri)
        out[tag + "_io_base"] = base
        gray = make_pixels(io, 19, 23, 2, hdri)
        out[tag + "_io_gray_base"] = gray
        for name, dt in IO_TYPES:
            for m in ("RGBA", "BGRA", "RGB", "ARGB", "BGRP", "RAB"):
                if dt in (np.float32, np.float64):
                    data = (io.random((7, 9, len(m))) * 1.3 - 0.15).astype(dt)
                else:
                    data = io.integers(0, np.iinfo(dt).max, (7, 9, len(m)), dtype=dt, endpoint=True)
                out["%s_import|%s|%s|data" % (tag, name, m)] = data
                out["%s_import|%s|%s" % (tag, name, m)] = ref.RefImage(base).import_pixels(5, 3, m, data).numpy()
            data = (io.random((7, 9, 2)).astype(dt) if dt in (np.float32, np.float64) else
                    io.integers(0, np.iinfo(dt).max, (7, 9, 2), dtype=dt, endpoint=True))
            out["%s_import|%s|IA|data" % (tag, name)] = data
            out["%s_import|%s|IA" % (tag, name)] = ref.RefImage(gray).import_pixels(5, 3, "IA", data).numpy()
            for m in ("RGBA", "BGRA", "RGB", "ARGB", "BGRP", "RGBP", "I", "IA", "RPPA"):
                out["%s_export|%s|%s" % (tag, name, m)] = ref.RefImage(base).export_pixels(4, 2, 11, 8, m, dt)
            out["%s_export_gray|%s|IA" % (tag, name)] = ref.RefImage(gray).export_pixels(4, 2, 11, 8, "IA", dt)
        out["%s_export_noalpha|uint16|RGBA" % tag] = ref.RefImage(base[:, :, :3].copy()).export_pixels(
            0, 0, 23, 19, "RGBA", np.uint16)
    # kernel lists: every kernel of the named hit-and-miss sets, rotation / mirror expansions
    for s in KERNEL_LISTS:
        first = ref.kernel(s)
        count = first[3]
        out["kernellist|%s|count" % s] = np.array([count])
        for i in range(count):
            values, x, y, _ = ref.kernel(s, i)
            out["kernellist|%s|%d" % (s, i)] = values
            out["kernellist_origin|%s|%d" % (s, i)] = np.array([x, y])
    xs = np.linspace(-4.5, 4.5, 181)
    img = ref.RefImage(np.zeros((2, 2, 4), np.uint16))
    for f in ("Lanczos", "Mitchell", "Catrom", "Triangle", "Box", "Gaussian", "Hann", "Spline", "Cubic",
              "Hermite", "Lanczos2", "LanczosSharp", "Robidoux", "Sinc", "Hamming", "Blackman", "Quadratic"):
        w, support = img.filter_weights(f, xs)
        out["filter|" + f] = w
        out["filter_support|" + f] = np.array([support])
    out["filter_xs"] = xs
    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    print("reference_vectors.npz: %d arrays" % len(out))


if __name__ == "__main__":
    if not os.path.isdir(REFERENCE):
        raise SystemExit("needs %s" % REFERENCE)
    perlmagick()
    reference_vectors()
