import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "envelope_fallback_ok: the test expects the library's f16x3 envelope guard to trip")


@pytest.fixture(autouse=True)
def _no_silent_envelope_fallback(request, capfd):
    """A precision="f16x3" model whose checkpoint the library judges outside the validated envelope is silently served by the
    exact-fp32 kernels (one line on stderr).  In a parity test that would make every f16x3 assertion vacuous -- round 5 met exactly
    that: a head-parallel kernel returning NaN, caught by the guard's probe, all comparisons green on fp32 -- so a test during which
    the guard trips FAILS unless it is marked envelope_fallback_ok."""
    yield
    if request.node.get_closest_marker("envelope_fallback_ok") is not None:
        return
    err = capfd.readouterr().err
    assert "outside the validated f16x3 envelope" not in err, "the f16x3 envelope guard tripped during this test:\n" + err[-600:]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
