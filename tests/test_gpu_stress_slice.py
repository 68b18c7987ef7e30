"""A fixed-seed slice of the randomised differential run (tests/stress_parity.py) inside the suite the driver runs:
every operator family of the stress driver, a few seconds each, against the compiled reference through the C ABI.
The reference pins this path with tolerance goldens only (PerlMagick/t/filter.t:39-201); the randomised run is what
found FAST ResizeImage thousands of levels off in round 5 — it must not live outside the suite."""
import time

import pytest

pytestmark = pytest.mark.gpu

# seconds per family: the whole module stays under a minute and a half on the GPU box (the reference's CPU time dominates)
BUDGET = {4: 7.0, 11: 7.0, 14: 10.0}
DEFAULT_BUDGET = 4.0


@pytest.fixture(scope="module")
def stress(im, refmod):
    import stress_parity
    stress_parity.setup(1)
    yield stress_parity
    im.set_precision(im.PRECISION_EXACT)


@pytest.mark.parametrize("op", range(16))
def test_stress_family(stress, im, op):
    assert op < stress.NUMBER_OF_OPS
    stress.setup(6000 + op)
    budget = BUDGET.get(op, DEFAULT_BUDGET)
    t0 = time.time()
    cases = failures = 0
    while time.time() - t0 < budget or cases < 3:
        failures += stress.run_case(op)
        cases += 1
    assert im.get_precision() == im.PRECISION_EXACT        # every case restores the mode it found
    assert failures == 0, "%d of %d random cases of family %d differ from the reference (see the MISMATCH lines)" % (
        failures, cases, op)


def test_stress_families_are_all_covered(stress):
    """The parametrisation above names every family the driver knows."""
    assert stress.NUMBER_OF_OPS == 16
