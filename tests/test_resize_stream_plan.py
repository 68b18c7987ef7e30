"""The one-launch vector-pipe resize (imagemagick_amd/csrc/resize_stream.hip) walks tables built on
the host (resize_stream_plan.hpp).  tests/cpu/resize_stream_plan_test.cpp emulates that walk — a lane
per source column, the window rows with dense scalar weights, the wave's row of the intermediate,
the listed edge columns, strips and chunks — on the CPU and compares it with the plain two-pass
evaluation of the same contribution lists; it also pins what the plan declines.  No GPU."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stream_plan_walk_matches_two_passes():
    exe = os.path.join(tempfile.mkdtemp(prefix="mh_plan_"), "stream_plan_test")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "imagemagick_amd", "csrc"),
                    os.path.join(ROOT, "tests", "cpu", "resize_stream_plan_test.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    assert out.returncode == 0 and "ALL OK" in out.stdout, out.stdout
