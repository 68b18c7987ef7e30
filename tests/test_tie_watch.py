"""FAST ResizeImage's one-launch kernels test every value of the rounded intermediate against the
nearest rounding boundary (imagemagick_amd/csrc/tie_watch.hpp: three integer instructions on the low
word of a double).  tests/cpu/tie_watch_test.cpp runs the same header on the host: values inside the
window are reported, values well outside are not, the window of an alpha-weighted colour widens with
the reciprocal of the alpha sum, and a window as wide as the tail's range reports everything.  No GPU."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tie_watch_windows_on_the_host():
    exe = os.path.join(tempfile.mkdtemp(prefix="mh_tie_"), "tie_watch_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "imagemagick_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpu", "tie_watch_test.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout


def test_tie_window_is_wide_enough_on_adversarial_frames():
    """tests/cpu/tie_window_sufficiency_test.cpp: the first filter over frames of tiny, binary and random
    alpha (Triangle, Catrom, Lanczos, Box; Q16 and float), in the one-launch kernels' order and in the
    reference's, 18 million pixels: every pixel whose rounded intermediate differs between the two
    orders was reported by the kernels' test."""
    exe = os.path.join(tempfile.mkdtemp(prefix="mh_tie_"), "tie_window_sufficiency_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "imagemagick_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpu", "tie_window_sufficiency_test.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout
