"""Micro-benchmark of svdx_gemm on the shapes of the SVD UNet step (run on the GPU box)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from svd_xtend_amd import kernels as K  # noqa: E402


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    be = K.backend()
    dev = torch.device("cuda")
    dt = torch.float16
    variants = [int(v) for v in os.environ.get("VARIANTS", "1,4").split(",")]
    shapes = [("sq4096", 4096, 4096, 4096, None), ("sq8192", 8192, 8192, 8192, None),
              ("L0 ff1 35840x2560x320", 35840, 2560, 320, None), ("L0 ff2 35840x320x1280", 35840, 320, 1280, None),
              ("L0 qkv 35840x960x320", 35840, 960, 320, None), ("L0 proj 35840x320x320", 35840, 320, 320, None),
              ("L1 ff1 8960x5120x640", 8960, 5120, 640, None), ("L1 ff2 8960x640x2560", 8960, 640, 2560, None),
              ("L2 ff1 2240x10240x1280", 2240, 10240, 1280, None), ("L2 ff2 2240x1280x5120", 2240, 1280, 5120, None),
              ("L3 ff1 560x10240x1280", 560, 10240, 1280, None),
              ("dW L0 ff1 2560x320x35840", 2560, 320, 35840, "dw"), ("dW L2 ff1 10240x1280x2240", 10240, 1280, 2240, "dw"),
              ("conv L0 320->320", 35840, 320, 2880, ("conv", 14, 40, 64, 320)),
              ("conv L0 960->320", 35840, 320, 8640, ("conv", 14, 40, 64, 960)),
              ("conv L1 640->640", 8960, 640, 5760, ("conv", 14, 20, 32, 640)),
              ("conv L2 1280->1280", 2240, 1280, 11520, ("conv", 14, 10, 16, 1280)),
              ("conv L3 1280->1280", 560, 1280, 11520, ("conv", 14, 5, 8, 1280)),
              ("tconv L0 320", 35840, 320, 960, ("t3", 1, 14, 2560, 320)),
              ("tconv L2 1280", 2240, 1280, 3840, ("t3", 1, 14, 160, 1280))]
    out = []
    for name, M, N, Kd, kind in shapes:
        B = (torch.randn(N, Kd, device=dev) * Kd ** -0.5).to(dt)
        gather = None
        kw = {}
        if kind is None or kind == "dw":
            A = torch.randn(M, Kd, device=dev).to(dt)
            lda = Kd
            if kind == "dw":
                C = torch.zeros(M, N, device=dev)
                tiles = ((M + 127) // 128) * ((N + 127) // 128)
                sk = max(1, min(512 // tiles, (Kd // 64) // 4, 64))
                kw = dict(out_mode=K.OUT_F32_ATOMIC, split_k=sk)
            else:
                C = torch.zeros(M, N, device=dev, dtype=dt)
        elif kind[0] == "conv":
            _, n, h, w, ci = kind
            A = torch.randn(n * h * w, ci, device=dev).to(dt)
            lda = ci
            gather = K.Gather(K.GATHER_CONV3X3, n_img=n, hi=h, wi=w, ho=h, wo=w, cin=ci, stride=1, lda=ci)
            C = torch.zeros(M, N, device=dev, dtype=dt)
        else:
            _, b, t, hw, ci = kind
            A = torch.randn(b * t * hw, ci, device=dev).to(dt)
            lda = ci
            gather = K.Gather(K.GATHER_TEMPORAL3, n_img=b, cin=ci, t=t, hw=hw, lda=ci)
            C = torch.zeros(M, N, device=dev, dtype=dt)
        row = dict(name=name, M=M, N=N, K=Kd)
        for v in variants:
            ms = bench(lambda: be.gemm(A, B, C, M, N, Kd, lda, Kd, N, gather=gather, variant=v, **kw))
            row[f"v{v}_ms"] = ms
            row[f"v{v}_tflops"] = 2.0 * M * N * Kd / ms / 1e9
        if kind is None:       # yardstick only: the vendor library on the same shape (never a product path)
            Bt = B.t()
            ms = bench(lambda: torch.matmul(A, Bt))
            row["hipblaslt_tflops"] = 2.0 * M * N * Kd / ms / 1e9
        print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
        out.append(row)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "gemm_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
