import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def emu_backend():
    """Install the torch emulation of libsvdx so host-side orchestration can run on CPU (tests only)."""
    from svd_xtend_amd import kernels
    import emul
    prev = kernels._backend
    kernels._set_backend_for_tests(emul.EmuBackend())
    yield
    kernels._set_backend_for_tests(prev)
