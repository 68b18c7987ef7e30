"""The NumPy model of the exact-integer matrix-core blur pass (tools/model_exact_i8.py: byte planes x
balanced tap digits, dropped weight classes, the ambiguity window of convolve_fused_exact.hip, against
the reference's fp64 loop restated operation by operation) as a test: no result outside the window may
round to another level than the reference's, on random, opaque, tiny-alpha, sparse-alpha,
half-transparent and exact-tie rows."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sigma", ["10", "2", "5.3"])
def test_no_level_differs_outside_the_ambiguity_window(sigma):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "model_exact_i8.py"), sigma, "12"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if "wrong outside the window" in l]
    assert len(lines) >= 6, out.stdout
    for line in lines:
        wrong = int(re.search(r"wrong outside the window\s+(\d+)", line).group(1))
        assert wrong == 0, line
    assert "eligible True" in out.stdout, out.stdout
