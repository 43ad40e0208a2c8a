"""Which clock / power does the chip hold under each kind of load of the step?  Loops ONE kernel for ~1.5 s at a time while bench.ClockSampler
reads the device's sysfs nodes every 50 ms: the two-stage 160 x 160 tile and the two-role 160 x 320 tile on the 64x40-level convolution, a
long-K linear, AdamW (pure HBM streaming), and the whole captured step -- with the rate each loop reaches.  If the matrix-pipe loops sit at a
lower clock than the step's average, their in-step ties are DVFS: a kernel that finishes its MFMAs sooner only raises the power it is capped at.

    python tools/clock_probe.py
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from svd_xtend_amd import kernels as K  # noqa: E402

dev = torch.device("cuda", 0)


def sampled(fn, seconds=1.5, label=""):
    import threading
    s = bench.ClockSampler(0)                      # only for its discovery of THIS device's sysfs nodes; sampled here every 50 ms
    stop = threading.Event()

    def run():
        while not stop.wait(0.05):
            s._read_sysfs()

    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=run, daemon=True)
    if s.node:
        th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    stop.set()
    if s.node:
        th.join(timeout=5)
    clk = sorted(s.samples)
    pw = sorted(s.power)
    med = lambda v: v[len(v) // 2] if v else float("nan")                                 # noqa: E731
    return el / n, med(clk), (clk[0] if clk else float("nan")), (clk[-1] if clk else float("nan")), med(pw), len(clk)


def main():
    be = K.backend()
    dt = torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    rows = []
    # 64x40-level 3x3 convolution, 320 -> 320 channels
    M, N, ci = 35840, 320, 320
    A = torch.randn(M, ci, generator=g).to(dt).to(dev)
    B = (torch.randn(N, 9 * ci, generator=g) * (9 * ci) ** -0.5).to(dt).to(dev)
    out = torch.empty(M, N, dtype=dt, device=dev)
    gather = K.Gather(K.GATHER_CONV3X3, n_img=14, hi=40, wi=64, ho=40, wo=64, cin=ci, stride=1, lda=ci)
    fl = 2.0 * M * N * 9 * ci
    for v in (6, 34, 36):
        t, c, c0, c1, p, n = sampled(lambda: be.gemm(A, B, out, M, N, 9 * ci, ci, 9 * ci, N, gather=gather, variant=v))
        rows.append((f"conv 35840x320x2880 tile {v}", t * 1e6, fl / t / 1e12, c, c0, c1, p, n))
    # square GEMM (the isolated figure of DESIGN 3.1)
    n8 = 8192
    A8 = (torch.rand(n8, n8, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    B8 = (torch.rand(n8, n8, generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    o8 = torch.empty(n8, n8, dtype=torch.bfloat16, device=dev)
    for v in (18, 32):
        t, c, c0, c1, p, n = sampled(lambda: be.gemm(A8, B8, o8, n8, n8, n8, n8, n8, n8, variant=v))
        rows.append((f"8192^3 bf16 tile {v}", t * 1e6, 2.0 * n8 ** 3 / t / 1e12, c, c0, c1, p, n))
    del A8, B8, o8
    # pure streaming
    x = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    y = torch.empty_like(x)
    t, c, c0, c1, p, n = sampled(lambda: y.copy_(x))
    rows.append(("copy 1 GiB -> 1 GiB (TB/s in the rate column)", t * 1e6, 2.0 * x.numel() * 4 / t / 1e12, c, c0, c1, p, n))
    del x, y
    # the whole step
    from svd_xtend_amd.train import GraphedStep, Trainer
    from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
    with torch.device(dev):
        model = UNetSpatioTemporalConditionModel()
    bench.init_weights_(model, seed=1234)
    tr = Trainer(model, dtype=dt, lr=1e-5)
    batch = bench.make_batch(1, 14, 40, 64, model.config.cross_attention_dim, seed=123, dev=dev)
    for _ in range(2):
        tr.step(batch)
    gs = GraphedStep(tr, batch)
    t, c, c0, c1, p, n = sampled(gs, seconds=3.0)
    rows.append(("captured step (24.80 TFLOP)", t * 1e6, 24.80e12 / t / 1e12, c, c0, c1, p, n))
    print(f"{'load':48s} {'us / launch':>12s} {'TF/s':>8s} {'sclk med':>9s} {'min':>6s} {'max':>6s} {'W med':>7s} samples")
    for r in rows:
        print(f"{r[0]:48s} {r[1]:12.1f} {r[2]:8.1f} {r[3]:9.0f} {r[4]:6.0f} {r[5]:6.0f} {r[6]:7.0f} {r[7]:4d}")


if __name__ == "__main__":
    main()
