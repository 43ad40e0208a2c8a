"""LayerNorm + GEGLU projection as one launch (svdx_ln_geglu_fwd) against the two launches it replaces, at the benched level shapes.
    python tools/ffn_bench.py [--dtype fp16] [--iters 30]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svd_xtend_amd import kernels as K  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda")
    k = K.backend()
    for (M, C) in [(35840, 320), (71680, 320)]:
        F = 4 * C
        g = torch.Generator(device="cpu").manual_seed(0)
        x = torch.randn(M, C, generator=g).to(dt).to(dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        w1 = (torch.randn(2 * F, C, generator=g) * C ** -0.5).to(dt).to(dev)
        b1 = torch.zeros(2 * F, device=dev)
        n, st = torch.empty(M, C, dtype=dt, device=dev), torch.empty(M, 2, device=dev)
        pre, hh = torch.empty(M, 2 * F, dtype=dt, device=dev), torch.empty(M, F, dtype=dt, device=dev)

        def fused():
            k.ln_geglu_fwd(x, gamma, beta, 1e-5, w1, b1, n, st, pre, hh, M, C, F)

        def unfused():
            k.ln_fwd(x, gamma, beta, n, st, M, C, 1e-5)
            k.gemm(n, w1, pre, M, 2 * F, C, C, C, 2 * F, bias=b1, variant=4, epilogue=K.EPI_GEGLU_FWD, aux_out=hh, aux_dim=F)

        out = {"shape": dict(M=M, C=C, F=F), "gflop": 2.0 * M * 2 * F * C / 1e9, "bytes_MB": (M * C * 2 * 2 + M * 3 * F * 2) / 1e6}
        for name, fn in (("unfused", unfused), ("fused", fused)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / args.iters * 1e3
            out[name] = dict(us=us, tflops=out["gflop"] / us / 1e3, tb_per_s=out["bytes_MB"] / us)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
