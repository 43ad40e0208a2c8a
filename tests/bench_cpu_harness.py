"""bench.py's host logic on CPU ranks (test infrastructure): the multi-rank paths of the driver's `bench.py --gpus N` -- schedule probe,
direct-exchange check, the timed loop, the real loop, the exchange timed alone, the JSON line -- executed under torch.distributed.run with
gloo, the kernel emulation of tests/emul.py as the backend and the handful of torch.cuda calls of the bench stubbed.  What it cannot show is
anything about RCCL, streams or peer mapping; what it does show is that no rank takes a different branch, hangs in a collective or trips
over a name.  Launched by tests/test_bench_multirank_cpu.py."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
os.environ.update(SVDX_BENCH_DEVICE="cpu", SVDX_DIST_BACKEND="gloo", SVDX_NO_CLOCK_SAMPLER="1")


class _Event:
    def __init__(self, enable_timing=False):
        self.t = None

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


if __name__ == "__main__":
    from emul import EmuBackend
    from svd_xtend_amd import kernels as K
    K._set_backend_for_tests(EmuBackend())
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda *a, **k: None
    torch.cuda.Event = _Event
    torch.cuda._sleep = lambda cycles: None
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // max(1, int(os.environ.get("WORLD_SIZE", "1")))))
    import bench
    bench.main()
