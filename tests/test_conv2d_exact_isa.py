"""convolve2d_exact.hip reads its matrix operands through asm and waits by count; that is only
sound while the compiler leaves the product loops free of copies and spills of the registers those
reads fill and of scalar loads (tools/check_conv2d_exact_isa.py has the reasoning).  This test builds
the ISA with the library's flags and runs that check on every instantiation — hipcc cross-compiles
without a GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_loops_hold_no_copies_spills_or_scalar_loads():
    if shutil.which("make") is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc / make here")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_conv2d_exact_isa.py")],
                         capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-1000:]
    lines = [l for l in out.stdout.splitlines() if "products in the loop" in l]
    assert len(lines) >= 16, out.stdout
    assert all(l.rstrip().endswith(" 0 suspicious") for l in lines), out.stdout
