"""Fused temporal self-attention op (svdx_tsa_fwd) against the four launches it replaces, at the benched level shapes.
FLOPs of the op: 8 M C^2 (q/k/v + out projections) + 4 M T C (attention core).  Prints one JSON line per shape:
    python tools/tsa_bench.py [--dtype fp16] [--iters 30]
Run it under rocprofv3 --kernel-trace --stats for kernel-side durations / --pmc SQ_VALU_MFMA_BUSY_CYCLES for MFMA busy."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svd_xtend_amd import kernels as K  # noqa: E402
from svd_xtend_amd.ops import Runtime, gemm_act  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda")
    k = K.backend()
    rt = Runtime(dt, dev)
    for (B, T, HW, heads) in [(1, 14, 2560, 5), (1, 14, 640, 5), (2, 14, 2560, 5)]:
        C = heads * 64
        M = B * T * HW
        g = torch.Generator(device="cpu").manual_seed(0)
        x = torch.randn(M, C, generator=g).to(dt).to(dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        wqkv = (torch.randn(3 * C, C, generator=g) * C ** -0.5).to(dt).to(dev)
        wo = (torch.randn(C, C, generator=g) * C ** -0.5).to(dt).to(dev)
        bo, cvec = torch.zeros(C, device=dev), torch.randn(B, C, generator=g).to(dev)
        n1, st = torch.empty(M, C, dtype=dt, device=dev), torch.empty(M, 2, device=dev)
        qkv, o, h1 = torch.empty(M, 3 * C, dtype=dt, device=dev), torch.empty(M, C, dtype=dt, device=dev), torch.empty(M, C, dtype=dt, device=dev)

        def fused():
            k.tsa_fwd(x, gamma, beta, 1e-5, wqkv, wo, bo, cvec, C, T * HW, 0, n1, st, qkv, o, h1, B, T, HW, C, heads, 0.125)

        def unfused():
            k.ln_fwd(x, gamma, beta, n1, st, M, C, 1e-5)
            gemm_act(rt, n1, wqkv, qkv, M, 3 * C, C, C, C, 3 * C)
            k.tattn_fwd(qkv, qkv[:, C:], qkv[:, 2 * C:], o, B, T, HW, heads, 3 * C, C, 0.125)
            gemm_act(rt, o, wo, h1, M, C, C, C, C, C, bias=bo, rowvec=cvec, rv_ld=C, rv_rpg=T * HW, res=x, ldres=C)

        flops = 8.0 * M * C * C + 4.0 * M * T * C
        out = {"shape": dict(B=B, T=T, HW=HW, C=C, M=M), "gflop": flops / 1e9, "bands": B * HW // K.tsa_pixels_per_band(T, HW),
               "rows_per_band": K.tsa_pixels_per_band(T, HW) * T}
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
            out[name] = {"us": us, "tflops": flops / us / 1e6, "frac_mfma_peak": flops / us / 1e6 / 2500.0}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
