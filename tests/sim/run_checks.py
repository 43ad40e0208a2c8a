"""Developer entry: run groups of tests/kernel_checks.py against the simulator build.  python tests/sim/run_checks.py [group ...] [--bf16]"""
import math
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE), HERE]
import kernel_checks as kc  # noqa: E402
from backend import SimBackend  # noqa: E402


def groups(P, dt):
    fns = {"gemm_tn": lambda: kc.check_gemm_tn(P, dt), "gemm_tn_s3": lambda: kc.check_gemm_tn(P, dt, 3), "gemm_tn_s4": lambda: kc.check_gemm_tn(P, dt, 4),
           "gemm_tn_v18": lambda: kc.check_gemm_tn(P, dt, 18),
           "gemm_tn_flat": lambda: kc.check_gemm_tn(P, dt, 64), "gemm_tn_v18_flat": lambda: kc.check_gemm_tn(P, dt, 18 | 64), "gemm_geglu": lambda: kc.check_gemm_geglu(P, dt),
           "small": lambda: kc.check_small(P, dt), "groupnorm": lambda: kc.check_groupnorm(P, dt), "layernorm": lambda: kc.check_layernorm(P, dt),
           "attention": lambda: kc.check_attention(P, dt), "temporal_attention": lambda: kc.check_temporal_attention(P, dt),
           "tsa": lambda: kc.check_tsa(P, dt), "encoders": lambda: kc.check_encoders(P, dt), "elementwise": lambda: kc.check_elementwise(P, dt),
           "optim": lambda: kc.check_optim(P, dt)}
    for v in (4, 6, 8, 16, 18, 22, 23, 24, 26, 32, 34, 36):
        fns[f"gemm_gn_v{v}"] = lambda v=v: kc.check_gemm_gn(P, dt, v)
    for v in (17, 18, 21, 26, 27, 32, 34):
        fns[f"gemm_geglu_v{v}"] = lambda v=v: kc.check_gemm_geglu(P, dt, v)
    for v in (1, 4, 6, 16, 18, 20, 23, 25, 27, 28, 32, 34, 36):
        fns[f"gemm_plain_v{v}"] = lambda v=v: kc.check_gemm_plain(P, dt, v)
        fns[f"gemm_gather_v{v}"] = lambda v=v: kc.check_gemm_gather(P, dt, v)
    return fns


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    dts = [torch.bfloat16] if "--bf16" in sys.argv else [torch.float16] if "--f16" in sys.argv else [torch.float16, torch.bfloat16]
    P = kc.Pair(SimBackend(), torch.device("cpu"))
    nbad = 0
    for dt in dts:
        fns = groups(P, dt)
        for name in (args or list(fns)):
            t = time.time()
            res = fns[name]()
            bad = [(l, e, tl) for l, e, tl in res if not (e <= tl and math.isfinite(e))]
            nbad += len(bad)
            print(f"{name:24s} {str(dt):16s} {len(res):4d} checks {len(bad):3d} bad  {time.time() - t:7.1f}s", flush=True)
            for b in bad[:6]:
                print("    ", b)
    sys.exit(1 if nbad else 0)
