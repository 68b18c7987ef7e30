import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def im():
    """The product binding; on a GPU box the native library MUST load."""
    import imagemagick_amd
    imagemagick_amd.load()
    # The library's default is FAST (what an unchanged caller gets); the parity tests are written
    # against the bit-identical mode and switch to FAST where they test it.
    imagemagick_amd.set_precision(imagemagick_amd.PRECISION_EXACT)
    # FAST ResizeImage keeps the two passes on frames too small to fill the chip with its one-launch walks
    # (resize.hip, launch_resize_fused); the suites' frames are small on purpose and must keep exercising them
    imagemagick_amd.set_option("MAGICKHIP_RESIZE_ONE_LAUNCH_MIN_PIXELS", "0")
    return imagemagick_amd


class _Options:
    """Library switches for one test (MhSetOption): the library reads MAGICKHIP_* from the
    environment once, at start-up, so tests flip its switches through the API; undone afterwards."""

    def __init__(self, im):
        self.im, self.saved = im, {}

    def set(self, name, value="1"):
        self.saved.setdefault(name, self.im.get_option(name))
        self.im.set_option(name, value)

    def setenv(self, name, value):          # the spelling the tests used with monkeypatch
        self.set(name, value)

    def restore(self):
        for name, value in self.saved.items():
            self.im.set_option(name, value)


@pytest.fixture
def options(im):
    o = _Options(im)
    yield o
    o.restore()


@pytest.fixture(scope="session")
def vectors():
    """Committed outputs of the reference's own CPU code (tests/golden/make_golden.py)."""
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.npz"))


@pytest.fixture(scope="session")
def refmod():
    """The compiled reference oracle (oracle/_ref); skip when it was not built."""
    from oracle import ref
    if not ref.available(False) or not ref.available(True):
        pytest.skip("compiled reference oracle (oracle/_ref) is not built")
    return ref


def make_pixels(rows, cols, channels, dtype, seed=42, kind="random"):
    """Deterministic synthetic Quantum buffers."""
    rng = np.random.default_rng(seed)
    if kind == "random":
        a = rng.integers(0, 65536, (rows, cols, channels), dtype=np.uint16)
    elif kind == "smooth":
        y, x = np.mgrid[0:rows, 0:cols]
        base = (x * 40000.0 / max(cols - 1, 1) + y * 20000.0 / max(rows - 1, 1))
        a = np.empty((rows, cols, channels), dtype=np.float64)
        for c in range(channels):
            a[:, :, c] = base * (0.6 + 0.1 * c) + rng.integers(0, 400, (rows, cols))
        a = np.clip(a, 0, 65535).astype(np.uint16)
    elif kind == "opaque":
        a = rng.integers(0, 65536, (rows, cols, channels), dtype=np.uint16)
        a[:, :, channels - 1] = 65535
    elif kind == "binary":
        a = (rng.random((rows, cols, channels)) > 0.6).astype(np.uint16) * 65535
    else:
        raise ValueError(kind)
    if dtype == np.float32:
        f = a.astype(np.float32)
        if kind == "random":
            f += rng.random((rows, cols, channels), dtype=np.float32)  # non-integral HDRI values
            f = np.minimum(f, np.float32(65535.0))
        return np.ascontiguousarray(f)
    return np.ascontiguousarray(a)


def to_device(array):
    import torch
    if array.dtype == np.uint16:
        return torch.from_numpy(array.view(np.int16)).cuda().view(torch.uint16)
    return torch.from_numpy(array).cuda()


def ulp_diff_f32(a, b):
    """Distance in float32 ULPs (monotone integer mapping of the bit patterns)."""
    ai = a.astype(np.float32).view(np.int32).astype(np.int64)
    bi = b.astype(np.float32).view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def assert_parity(got, want, exact=True, what="", max_ulp=0, residue=0.0):
    """EXACT mode: Q16 results are bit-identical; float Quantum results are
    bit-identical too (max_ulp=0) except where the caller allows 1 float ULP
    because a libm function (pow in the Lab transform) is evaluated by a
    different library on the device.  FAST mode (Q16 only): +-1 Quantum level."""
    assert got.shape == want.shape, "%s shape %s != %s" % (what, got.shape, want.shape)
    if want.dtype == np.uint16:
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        limit = 0 if exact else 1
        assert d.max() <= limit, "%s: max |diff| = %d (limit %d), %d of %d differ" % (
            what, d.max(), limit, int((d > limit).sum()), d.size)
        return float((d == 0).mean())
    u = ulp_diff_f32(got, want)
    if residue > 0.0:
        # FAST on float Quantum: a result that is the residue of a cancellation (1e-20 out of terms of
        # 1e-8) has no meaningful last place; such values agree to `residue`, an absolute bound far
        # below a float ULP at the scale of the samples
        with np.errstate(invalid="ignore"):
            u = np.where(np.abs(got.astype(np.float64) - want.astype(np.float64)) <= residue, 0, u)
    assert u.max() <= max_ulp, "%s: max ULP diff = %d (limit %d), %d of %d over" % (
        what, u.max(), max_ulp, int((u > max_ulp).sum()), u.size)
    return float((u == 0).mean())
