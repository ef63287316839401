import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import positionbaseddynamics_amd as pbd
        return pbd.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def have_gpu():
    return _has_gpu()


@pytest.fixture(autouse=True)
def _bounds_record_stays_empty(request):
    """When the library under test is the sanitizer-grade debug build (PBDX_LIB=.../libpbdx_bounds.so, csrc/pbdx_bounds.h) every GPU test also
    requires that none of its kernels addressed anything out of range."""
    yield
    if request.node.get_closest_marker("gpu") is None or "bounds" not in os.path.basename(os.environ.get("PBDX_LIB", "")):
        return
    import positionbaseddynamics_amd as pbd
    rep = pbd.bounds_report(0, reset=True)
    assert rep["checked"], "PBDX_LIB names a bounds build but the loaded library checks nothing"
    assert rep["violations"] == 0, "out-of-range access in %s: %r" % (request.node.name, rep)
