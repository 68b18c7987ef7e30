"""The one-launch matrix-pipe resize (imagemagick_amd/csrc/resize_mfma.hip) walks tables built on
the host (resize_mfma_plan.hpp).  tests/cpu/resize_mfma_plan_test.cpp emulates that walk — strips,
16-column blocks, the ring of K-blocks, the f64 MFMA lane layouts — on the CPU and compares it with
the plain two-pass evaluation of the same contribution lists.  No GPU."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plan_walk_matches_two_passes():
    exe = os.path.join(tempfile.mkdtemp(prefix="mh_plan_"), "plan_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "imagemagick_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpu", "resize_mfma_plan_test.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout
