"""Isolated timing of the spatial attention kernels (csrc/attention.hip) at the shapes the c2 / c4 steps run, random data.

    python tools/attn_bench.py                      # current build
    SVDX_LIB=<other libsvdx.so> python tools/attn_bench.py   # a second build, for A/B on one box

Prints one line per shape: forward, dQ kernel, dK/dV kernel in us and TFLOP/s (4*S*S*64 flop per head for the forward,
3 resp. 4 matmuls of 2*S*S*64 in the two backward kernels)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from svd_xtend_amd import kernels as K  # noqa: E402

dev = torch.device("cuda")
be = K.backend()
SHAPES = [("c2 L0", 14, 5, 2560), ("c2 L1", 14, 10, 640), ("c2 L2", 14, 20, 160), ("c2 L3", 14, 20, 40), ("c4 L0", 25, 5, 9216),
          ("c4 L1", 25, 10, 2304)]


def timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dt = torch.float16
    out = []
    for name, nb, heads, S in SHAPES:
        C = heads * 64
        g = torch.Generator(device="cpu").manual_seed(S)
        qkv = torch.randn(nb * S, 3 * C, generator=g).to(dt).to(dev)
        d_o = torch.randn(nb * S, C, generator=g).to(dt).to(dev)
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        o = torch.empty(nb * S, C, dtype=dt, device=dev)
        lse = torch.empty(nb * heads * S, dtype=torch.float32, device=dev)
        D = torch.empty_like(lse)
        dqkv = torch.empty_like(qkv)
        sc = 0.125
        it = 5 if S > 4000 else 20
        be.attn_fwd(q, k, v, o, lse, nb, heads, S, 3 * C, C, sc)
        be.attn_bwd_prep(o, d_o, D, nb, heads, S, C)
        t_f = timeit(lambda: be.attn_fwd(q, k, v, o, lse, nb, heads, S, 3 * C, C, sc), it)
        t_q = timeit(lambda: be.attn_bwd_dq(q, k, v, d_o, lse, D, dqkv[:, :C], nb, heads, S, 3 * C, C, 3 * C, sc), it)
        t_kv = timeit(lambda: be.attn_bwd_dkv(q, k, v, d_o, lse, D, dqkv[:, C:2 * C], dqkv[:, 2 * C:], nb, heads, S, 3 * C, C, 3 * C, sc), it)
        mm = 2.0 * S * S * 64 * nb * heads
        out.append(f"{name} S={S}: fwd {t_f:.1f}us {2 * mm / t_f / 1e6:.0f}TF | dq {t_q:.1f}us {3 * mm / t_q / 1e6:.0f}TF | "
                   f"dkv {t_kv:.1f}us {4 * mm / t_kv / 1e6:.0f}TF")
    print(os.environ.get("SVDX_LIB", "default"), flush=True)
    print("\n".join(out), flush=True)


if __name__ == "__main__":
    main()
