import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is memory-bound torch fp32 work: on the GPU box's 2 x 64-core host one step of the 64x40-level block takes 30 s on 32
    # threads, 43 s on 64, 81 s on PyTorch's default of 128 and 419 s on 256 (profiles/r6b_oracle_threads.txt) -- pin it.
    import torch
    n = int(os.environ.get("SVDX_ORACLE_THREADS", "32"))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a box without an MI355X (or without the built library) skips the `gpu` tests instead of erroring."""
    import torch
    lib = os.path.join(ROOT, "svd_xtend_amd", "csrc", "libsvdx.so")
    why = None
    if not torch.cuda.is_available():
        why = "no HIP device visible"
    elif not os.path.exists(lib):
        why = f"{lib} not built"
    if why is None:
        return
    skip = pytest.mark.skip(reason=why)
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def emu_backend():
    """Install the torch emulation of libsvdx so host-side orchestration can run on CPU (tests only)."""
    from svd_xtend_amd import kernels
    import emul
    prev = kernels._backend
    kernels._set_backend_for_tests(emul.EmuBackend())
    yield
    kernels._set_backend_for_tests(prev)
