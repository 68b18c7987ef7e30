"""The hot kernels' registers and scratch, read from the built library's own code-object metadata
(tools/kernel_resources.py): no GPU needed.  A fused blur kernel sits at the 128 registers of four waves a SIMD; one
register more and the compiler spills into the walk, where a reload waits for every store in flight — that cost the
exact kernel 6 % in round 6 and no parity test can see it."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def kernels():
    import kernel_resources
    if not os.path.exists(kernel_resources.DEFAULT_LIBRARY) or not os.path.exists(kernel_resources.OBJCOPY):
        pytest.skip("library or llvm-objcopy not present")
    rows = kernel_resources.kernel_resources()
    assert len(rows) > 500, "the library's code objects were not found"
    return rows


def named(kernels, *parts):
    rows = [k for k in kernels if all(p in k["name"] for p in parts)]
    assert rows, "no kernel named %s" % (parts,)
    return rows


def test_fused_blur_kernels_keep_four_waves_a_simd_and_do_not_spill(kernels):
    for k in named(kernels, "blur_fused_hybrid_kernel<"):
        assert k["vgprs"] <= 128 and k["scratch"] == 0, k
    for k in named(kernels, "blur_fused_exact_kernel<"):
        assert k["vgprs"] <= 128, k
        unsharp = k["name"].split("<")[1].split(">")[0].split(",")[2].strip() == "true"
        # (UnsharpMask's 79-tap instantiations keep two registers in scratch, read only where a wave raises the
        # give-up word: convolve_fused_exact.hip)
        assert k["scratch"] <= (8 if unsharp else 0), k


def test_streaming_kernels_of_the_bench_configurations_do_not_spill(kernels):
    for parts in (("resize_stream_kernel<",), ("lab_histogram_fast_kernel<true>",), ("morph_rects_kernel<",),
                  ("conv2d_exact_kernel<",), ("stretch_apply_kernel<",), ("resize_vertical_kernel<",),
                  ("resize_horizontal_kernel<",)):
        for k in named(kernels, *parts):
            # (the alpha-weighted seven-neighbour 3x enlargement of a Q16 frame keeps three dwords in scratch at three
            # waves a SIMD; two waves without the spill measured slower, 3.73 against 3.3 ms per 8192^2)
            allowed = 16 if "resize_stream_kernel<unsigned short, true, 3, 7, 6>" in k["name"] else 0
            assert k["scratch"] <= allowed, k


def test_resize_stream_fits_three_waves_a_simd_where_it_says_so(kernels):
    # 4x Lanczos on float RGBA (C3): 512 registers a SIMD lane / 3 waves
    for k in named(kernels, "resize_stream_kernel<float, true, 4, 7, 6>"):
        assert k["vgprs"] <= 168 and k["lds"] * 3 <= 160 * 1024, k
