"""SimBackend: the ctypes binding of svd_xtend_amd.kernels pointed at tests/sim/_build/libsvdx_sim.so -- the SAME C-ABI entry points and
the same kernel sources, executed lane by lane on the host (tests/sim/sim_rt.h).  TEST INFRASTRUCTURE: constructed only by tests."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from svd_xtend_amd import kernels as K  # noqa: E402


class SimBackend(K.HipBackend):
    def __init__(self, force_build: bool = False):
        import build_sim
        self.lib = K.load_library(build_sim.build(force=force_build))
        self._zero_page = torch.zeros(1024, dtype=torch.uint8)
        self._log_extra = None
        self.n_calls = 0
        self.launch_log = None

    @staticmethod
    def _stream():
        return None
