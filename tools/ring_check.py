"""Development check of the ring-staged GEMM tiles (run on the GPU box):

  1. every kernel-check group of the new variants against the fp32 emulation (tests/kernel_checks.py), both dtypes;
  2. a race screen: on step-sized problems every tile variant must reproduce the two-stage 128x160 kernel BIT FOR BIT (the K order per
     output element is the same in all of them), several launches each, with a cache-thrashing kernel in between;
  3. an isolated timing table of the step's heaviest problems over (variant, split-K).

    python tools/ring_check.py [check] [race] [time]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from svd_xtend_amd import kernels as K  # noqa: E402
from svd_xtend_amd.ops import STAGED_TILES  # noqa: E402
from svd_xtend_amd.ops import TILE_OF_VARIANT as _COST_MODEL_TILES  # noqa: E402

TILE_OF_VARIANT = {**_COST_MODEL_TILES, **STAGED_TILES}      # the staged tuner candidates are screened and timed like the rest

dev = torch.device("cuda")
be = K.backend()


def problem(kind, dt, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    name, M, N, Kd, spec = kind
    B = (torch.randn(N, Kd, generator=g) * Kd ** -0.5).to(dt).to(dev)
    gather = None
    if spec is None:
        A = torch.randn(M, Kd, generator=g).to(dt).to(dev)
        lda = Kd
    elif spec[0] == "conv":
        _, n, h, w, ci = spec
        A = torch.randn(n * h * w, ci, generator=g).to(dt).to(dev)
        lda = ci
        gather = K.Gather(K.GATHER_CONV3X3, n_img=n, hi=h, wi=w, ho=h, wo=w, cin=ci, stride=1, lda=ci)
    else:
        _, b, t, hw, ci = spec
        A = torch.randn(b * t * hw, ci, generator=g).to(dt).to(dev)
        lda = ci
        gather = K.Gather(K.GATHER_TEMPORAL3, n_img=b, cin=ci, t=t, hw=hw, lda=ci)
    return A, B, lda, gather


SHAPES = [("L0 conv 320", 35840, 320, 2880, ("conv", 14, 40, 64, 320)),
          ("L0 conv 960->320", 35840, 320, 8640, ("conv", 14, 40, 64, 960)),
          ("L0 ff2 K1280", 35840, 320, 1280, None), ("L0 dx K2560", 35840, 320, 2560, None), ("L0 proj K320", 35840, 320, 320, None),
          ("L0 qkv", 35840, 960, 320, None),
          ("L1 conv 640", 8960, 640, 5760, ("conv", 14, 20, 32, 640)), ("L1 dx K5120", 8960, 640, 5120, None),
          ("L1 ff2 K2560", 8960, 640, 2560, None), ("L1 proj K640", 8960, 640, 640, None),
          ("L2 conv 1280", 2240, 1280, 11520, ("conv", 14, 10, 16, 1280)), ("L2 dx K10240", 2240, 1280, 10240, None),
          ("L2 ff2 K5120", 2240, 1280, 5120, None), ("L2 proj K1280", 2240, 1280, 1280, None),
          ("L2 tconv", 2240, 1280, 3840, ("t3", 1, 14, 160, 1280)),
          ("L3 conv 1280", 560, 1280, 11520, ("conv", 14, 5, 8, 1280)), ("L3 proj K1280", 560, 1280, 1280, None),
          ("L3 tconv", 560, 1280, 3840, ("t3", 1, 14, 40, 1280))]


def run_gemm(A, B, lda, gather, M, N, Kd, variant, split, out, slabs):
    if split == 1:
        be.gemm(A, B, out, M, N, Kd, lda, Kd, N, gather=gather, variant=variant)
    else:
        be.gemm(A, B, slabs, M, N, Kd, lda, Kd, N, gather=gather, out_mode=K.OUT_F32_SLAB, split_k=split, variant=variant)
        be.gemm_finalize(slabs, split, M * N, out, M, N, N)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3      # us


def do_check():
    import kernel_checks as kc
    P = kc.Pair(be, dev)
    bad = 0
    n = 0
    for dt in (torch.float16, torch.bfloat16):
        groups = []
        for v in kc.RING_VARIANTS:
            groups += [(f"plain v{v}", lambda v=v: kc.check_gemm_plain(P, dt, v)), (f"gather v{v}", lambda v=v: kc.check_gemm_gather(P, dt, v))]
        for v in (32, 34):          # the two-role tiles of round 6
            groups += [(f"plain v{v}", lambda v=v: kc.check_gemm_plain(P, dt, v)), (f"gather v{v}", lambda v=v: kc.check_gemm_gather(P, dt, v)),
                       (f"gn v{v}", lambda v=v: kc.check_gemm_gn(P, dt, v))]
        groups += [(f"geglu v{v}", lambda v=v: kc.check_gemm_geglu(P, dt, v)) for v in (17, 18, 21, 32, 34)]
        groups += [(f"tn s{s}", lambda s=s: kc.check_gemm_tn(P, dt, s)) for s in (0, 3, 4, 18)]
        for name, fn in groups:
            try:
                rows = fn()
            except Exception as e:  # noqa: BLE001
                print(f"EXC [{dt}] {name}: {e!r}", flush=True)
                bad += 1
                continue
            fails = [(l, e, t) for l, e, t in rows if not (e <= t)]
            n += len(rows)
            bad += len(fails)
            print(f"[{dt}] {name}: {len(rows)} checks, {len(fails)} failed" + (f"  first: {fails[:3]}" if fails else ""), flush=True)
    print(f"CHECK SUMMARY: {n} checks, {bad} failed", flush=True)


def do_race():
    dt = torch.float16
    thrash = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    total_bad = 0
    for kind in [SHAPES[0], SHAPES[6], SHAPES[10], SHAPES[15], SHAPES[3], SHAPES[13]]:
        name, M, N, Kd, _ = kind
        A, B, lda, gather = problem(kind, dt)
        ref = torch.empty(M, N, dtype=dt, device=dev)
        run_gemm(A, B, lda, gather, M, N, Kd, 7, 1, ref, None)
        torch.cuda.synchronize()
        for v in TILE_OF_VARIANT:
            bm, bn = TILE_OF_VARIANT[v][:2]
            if (bn == 160 and N % 160) or (bn == 256 and N % 256) or (bn == 320 and N % 320):
                continue
            nbad = 0
            for rep in range(4):
                out = torch.full((M, N), float("nan"), dtype=dt, device=dev)
                thrash.add_(1)
                run_gemm(A, B, lda, gather, M, N, Kd, v, 1, out, None)
                torch.cuda.synchronize()
                if not torch.equal(out, ref):
                    nbad += 1
                    d = (out.float() - ref.float()).abs()
                    print(f"  RACE? {name} v{v} rep {rep}: {int((out != ref).sum())} elements differ, max {float(d.nan_to_num(1e9).max()):.3e}", flush=True)
            total_bad += nbad
            print(f"race {name} v{v}: {'ok' if nbad == 0 else 'MISMATCH x' + str(nbad)}", flush=True)
        # split-K through the slabs: same bits for every tile shape at equal split (the per-slice K ranges are the same)
        for sp in (2, 4):
            refs = None
            for v in (7, 16, 20, 21):
                if TILE_OF_VARIANT[v][1] == 160 and N % 160:
                    continue
                slabs = torch.empty(sp, M, N, device=dev)
                out = torch.empty(M, N, dtype=dt, device=dev)
                run_gemm(A, B, lda, gather, M, N, Kd, v, sp, out, slabs)
                torch.cuda.synchronize()
                if refs is None:
                    refs = out.clone()
                elif not torch.equal(out, refs):
                    total_bad += 1
                    print(f"  RACE? {name} v{v} split {sp}: differs from v7", flush=True)
    print(f"RACE SUMMARY: {total_bad} mismatching runs", flush=True)


def do_time():
    dt = torch.float16
    rows = []
    for kind in SHAPES:
        name, M, N, Kd, _ = kind
        A, B, lda, gather = problem(kind, dt)
        out = torch.empty(M, N, dtype=dt, device=dev)
        kt = Kd // 64
        res = {}
        for v, (bm, bn, st, waves) in TILE_OF_VARIANT.items():
            if (bn == 160 and N % 160) or (bn == 256 and N % 256) or (bn == 320 and N % 320) or (bn == 128 and N % 128) or (waves == 8 and M < 2 * bm):
                continue
            tiles = -(-M // bm) * -(-N // bn)
            for sp in (1, 2, 3, 4, 6, 8, 12):
                if sp > 1 and (tiles * sp > 768 or kt // sp < 8):
                    continue
                slabs = torch.empty(sp, M, N, device=dev) if sp > 1 else None
                us = timeit(lambda: run_gemm(A, B, lda, gather, M, N, Kd, v, sp, out, slabs))
                res[f"v{v}s{sp}"] = us
        best = sorted(res.items(), key=lambda kv: kv[1])[:6]
        old = {k: v for k, v in res.items() if k.startswith(("v6s", "v7s", "v8s"))}
        bo = min(old.items(), key=lambda kv: kv[1]) if old else ("-", float("nan"))
        fl = 2.0 * M * N * Kd
        print(f"{name:18s} {M:6d}x{N:5d}x{Kd:6d}  old best {bo[0]:7s} {bo[1]:7.1f} us {fl / bo[1] / 1e6:6.0f} TF | " +
              "  ".join(f"{k} {u:.1f} ({fl / u / 1e6:.0f})" for k, u in best), flush=True)
        rows.append(dict(name=name, M=M, N=N, K=Kd, us=res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "ring_time.json"), "w"))


def do_big():
    """square problems on uniform random [-1, 1) operands (the data fill cdna_hip_programming.md quotes its 8-phase template on)"""
    dt = torch.bfloat16
    for n in (4096, 8192):
        g = torch.Generator(device="cpu").manual_seed(n)
        A = (torch.rand(n, n, generator=g) * 2 - 1).to(dt).to(dev)
        B = (torch.rand(n, n, generator=g) * 2 - 1).to(dt).to(dev)
        out = torch.empty(n, n, dtype=dt, device=dev)
        res = {}
        for v in (6, 8, 16, 17, 18, 32, 34):
            if TILE_OF_VARIANT[v][1] in (160, 320) and n % TILE_OF_VARIANT[v][1]:
                continue
            res[v] = timeit(lambda: be.gemm(A, B, out, n, n, n, n, n, n, variant=v), iters=10)
        fl = 2.0 * n ** 3
        print(f"square {n}: " + "  ".join(f"v{v} {u:.0f} us ({fl / u / 1e6:.0f} TF)" for v, u in sorted(res.items(), key=lambda kv: kv[1])), flush=True)


TN_SHAPES = [(35840, 2560, 320), (35840, 320, 1280), (35840, 960, 320), (35840, 320, 320), (8960, 5120, 640), (8960, 640, 2560), (8960, 1920, 640),
             (8960, 640, 640), (2240, 10240, 1280), (2240, 1280, 5120), (2240, 3840, 1280), (2240, 1280, 1280), (560, 10240, 1280), (560, 1280, 5120)]


def do_tn_time():
    """weight-gradient shapes of the step (R rows, output N x K) over (kernel variant, row slices)"""
    dt = torch.float16
    tiles_of = {2: (128, 128), 18: (256, 256), 12: (128, 256), 13: (128, 384), 21: (256, 128)}
    for (R, N, Kd) in TN_SHAPES:
        A = torch.randn(R, N, device=dev).to(dt)
        B = torch.randn(R, Kd, device=dev).to(dt)
        dst = torch.zeros(N, Kd, device=dev)
        cs_out = torch.zeros(N, device=dev)
        rt = (R + 63) // 64
        res = {}
        for v, (tm, tk) in tiles_of.items():
            tiles = -(-N // tm) * -(-Kd // tk)
            for sk in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32):
                if sk > 1 and (tiles * sk > (1200 if v == 2 else 600) or rt // sk < 4):
                    continue
                if sk == 1:
                    fn = lambda: be.gemm_tn(A, B, dst, R, N, Kd, N, Kd, Kd, out_mode=K.OUT_F32, a_colsum=cs_out, stages=v)
                else:
                    slabs = torch.empty(sk, N, Kd, device=dev)
                    cs = torch.empty(sk, N, device=dev)

                    def fn(slabs=slabs, cs=cs, sk=sk):
                        be.gemm_tn(A, B, slabs, R, N, Kd, N, Kd, Kd, out_mode=K.OUT_F32_SLAB, split_k=sk, a_colsum=cs, stages=v)
                        be.gemm_finalize(slabs, sk, N * Kd, dst, N, Kd, Kd, accumulate_f32=2, dtype=dt, colsum_slabs=cs, colsum_out=cs_out)
                res[f"v{v}s{sk}"] = timeit(fn)
        fl = 2.0 * R * N * Kd
        old = min(((k, u) for k, u in res.items() if k.startswith("v2s")), key=lambda kv: kv[1])
        best = sorted(res.items(), key=lambda kv: kv[1])[:5]
        print(f"tn R={R:6d} {N:5d}x{Kd:5d}  old best {old[0]:7s} {old[1]:7.1f} us {fl / old[1] / 1e6:5.0f} TF | " +
              "  ".join(f"{k} {u:.1f} ({fl / u / 1e6:.0f})" for k, u in best), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "race", "time"]
    if "check" in what:
        do_check()
    if "race" in what:
        do_race()
    if "time" in what:
        do_time()
    if "tn" in what:
        do_tn_time()
    if "big" in what:
        do_big()
