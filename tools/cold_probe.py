"""Where does a short linear's in-step penalty come from?  Inside the step the K <= 1280 linears take 3-12 us longer than back to back in
isolation (profiles/r6j_step_categories.txt against r6m_ring_time_tile36.txt).  This probe times one launch of each shape
    warm      back to back (operands in the caches),
    cold      after a 1 GB write has flushed L2 and the Infinity Cache,
    cold+W    cold, then the weights alone touched again (a read pass over B),
    cold+A    cold, then the activations alone touched again,
with the tile / split the cost model picks: which operand's first touch is the penalty, and what a prefetch of it could win.

    python tools/cold_probe.py
"""
import os
import sys
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from svd_xtend_amd import kernels as K  # noqa: E402
from svd_xtend_amd import ops  # noqa: E402

dev = torch.device("cuda")
be = K.backend()
SHAPES = [(35840, 320, 320), (35840, 320, 1280), (35840, 320, 2560), (35840, 960, 320), (8960, 640, 640), (8960, 640, 2560), (8960, 1920, 640),
          (2240, 1280, 1280), (2240, 1280, 5120), (2240, 3840, 1280), (560, 1280, 1280)]


def main():
    dt = torch.float16
    rt = SimpleNamespace(gemm_variant=4, split_k=True)
    thrash = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    print(f"{'M x N x K':>22s}  cfg        warm    cold  cold+W  cold+A   (us; median of 9)")
    for (M, N, Kd) in SHAPES:
        g = torch.Generator(device="cpu").manual_seed(M + N + Kd)
        A = torch.randn(M, Kd, generator=g).to(dt).to(dev)
        B = (torch.randn(N, Kd, generator=g) * Kd ** -0.5).to(dt).to(dev)
        out = torch.empty(M, N, dtype=dt, device=dev)
        split, v = ops.choose_cfg(rt, M, N, Kd, N, 0)
        slabs = torch.empty(split, M, N, device=dev) if split > 1 else None

        def run():
            if split == 1:
                be.gemm(A, B, out, M, N, Kd, Kd, Kd, N, variant=v)
            else:
                be.gemm(A, B, slabs, M, N, Kd, Kd, Kd, N, out_mode=K.OUT_F32_SLAB, split_k=split, variant=v)
                be.gemm_finalize(slabs, split, M * N, out, M, N, N)

        def once(prep):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(9):
                prep()
                e0.record()
                run()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            return sorted(ts)[len(ts) // 2]

        for _ in range(3):
            run()
        flush = lambda: thrash.add_(1)                                                   # noqa: E731
        res = [once(lambda: None), once(flush), once(lambda: (flush(), B.view(torch.int16).sum())), once(lambda: (flush(), A.view(torch.int16).sum()))]
        print(f"{M:7d} x{N:5d} x{Kd:6d}  ({split},{v:2d})  " + "  ".join(f"{t:6.1f}" for t in res), flush=True)


if __name__ == "__main__":
    main()
