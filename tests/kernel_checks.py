"""Per-kernel parity checks: libsvdx (HIP) vs the fp32 torch emulation (tests/emul.py) on identical seeded inputs.

Each check returns a list of (label, err, tol) triples; `run_all` executes everything without stopping so one GPU
run reports every kernel.  Used by tests/test_kernels_gpu.py (pytest -m gpu) and tools/gpu_report.py.
"""
import math

import torch

import emul
from svd_xtend_amd import kernels as K

DTYPES = (torch.float16, torch.bfloat16)


def tol_for(dt, scale=1.0):
    return (2e-3 if dt == torch.float16 else 1.6e-2) * scale


def relerr(a, b):
    """max |a - b| / max |b|, and -- for matrices -- the same ratio per ROW with the row's own magnitude in the denominator
    (floored at 5 % of the tensor's), weighted 1/4: an error confined to rows that are far smaller than the tensor's maximum (a
    mis-masked padded frame, a wrong tile in a quiet region) cannot hide behind the global maximum."""
    a, b = a.float(), b.float()
    if not torch.isfinite(a).all():
        return float("inf")
    gmax = b.abs().max() + 1e-6
    err = float((a - b).abs().max() / gmax)
    if a.ndim == 2 and a.shape[0] > 1:
        rows = (a - b).abs().amax(1) / (b.abs().amax(1) + 0.05 * gmax)
        err = max(err, 0.25 * float(rows.max()))
    return err


def cos_rows_min(a, b):
    """smallest per-row cosine between two matrices (rows with a negligible reference norm are skipped)."""
    a, b = a.double(), b.double()
    nb = b.norm(dim=1)
    keep = nb > 1e-3 * nb.max()
    c = (a * b).sum(1)[keep] / (a.norm(dim=1)[keep] * nb[keep] + 1e-30)
    return float(c.min())


def rnd(shape, dt, dev, g, scale=1.0):
    return (torch.randn(shape, generator=g, device="cpu") * scale).to(dt).to(dev)


def rndf(shape, dev, g, scale=1.0):
    return (torch.randn(shape, generator=g, device="cpu") * scale).to(dev)


class Pair:
    """Runs the same call on the implementation under test and on the emulation, on cloned outputs."""

    def __init__(self, impl, dev):
        self.impl, self.ref, self.dev = impl, emul.EmuBackend(), dev

    def run(self, name, args_fn, outs):
        """args_fn(outs_dict) -> (args, kwargs).  `outs` maps name -> initial tensor.  Returns (impl_outs, ref_outs)."""
        o1 = {k: v.clone() for k, v in outs.items()}
        o2 = {k: v.clone() for k, v in outs.items()}
        a, kw = args_fn(o1)
        getattr(self.impl, name)(*a, **kw)
        a, kw = args_fn(o2)
        getattr(self.ref, name)(*a, **kw)
        if self.dev.type == "cuda":
            torch.cuda.synchronize()
        return o1, o2


def check_gemm_plain(P, dt, variant):
    g = torch.Generator().manual_seed(1)
    res = []
    shapes = [(128, 128, 64), (200, 320, 320), (1000, 4, 576), (130, 2560, 128), (64, 640, 1280), (3, 320, 64), (300, 960, 192)]
    for (M, N, Kd) in shapes:
        A, B = rnd((M, Kd), dt, P.dev, g), rnd((N, Kd), dt, P.dev, g, Kd ** -0.5)
        bias, R = rndf((N,), P.dev, g), rnd((M, N), dt, P.dev, g)
        rv = rndf((4, N), P.dev, g)
        rpg = (M + 3) // 4
        for mode in ("plain", "bias_res", "rowvec", "rowvec_mod", "f32", "atomic_split"):
            kw = dict(variant=variant)
            out = torch.zeros(M, N, dtype=dt, device=P.dev)
            if mode == "bias_res":
                kw.update(bias=bias, res=R, ldres=N)
            elif mode == "rowvec":
                kw.update(bias=bias, rowvec=rv, rv_ld=N, rv_rpg=rpg)
            elif mode == "rowvec_mod":
                kw.update(rowvec=rv, rv_ld=N, rv_mod=4)
            elif mode == "f32":
                kw.update(out_mode=K.OUT_F32, alpha=0.5)
                out = torch.zeros(M, N, dtype=torch.float32, device=P.dev)
            elif mode == "atomic_split":
                sk = 2 if Kd >= 128 else 1
                kw.update(out_mode=K.OUT_F32_ATOMIC, split_k=sk, bias=bias)
                out = torch.ones(M, N, dtype=torch.float32, device=P.dev)
            o1, o2 = P.run("gemm", lambda o: ((A, B, o["C"], M, N, Kd, Kd, Kd, N), kw), dict(C=out))
            res.append((f"gemm v{variant} {M}x{N}x{Kd} {mode}", relerr(o1["C"], o2["C"]), tol_for(dt)))
    # second operand pair (svdx_gemm_dual, the LoRA term): strided A2 / B2 inside wider buffers, K2 = 64 / 192, K = 64 (one main tile)
    if 2 <= variant < 16:          # the second-operand loop exists for the two-stage four-wave tiles only
        for (M, N, Kd, K2, pa, pb) in [(200, 320, 320, 64, 64, 64), (300, 960, 320, 192, 192, 192), (130, 640, 64, 64, 192, 256),
                                       (1000, 1280, 1280, 192, 192, 192), (70, 128, 128, 128, 128, 128)]:
            A, B = rnd((M, Kd), dt, P.dev, g), rnd((N, Kd), dt, P.dev, g, Kd ** -0.5)
            A2w, B2w = rnd((M, pa), dt, P.dev, g), rnd((N, pb), dt, P.dev, g, K2 ** -0.5)
            A2, B2 = A2w[:, pa - K2:], B2w[:, pb - K2:]
            bias, R, rv = rndf((N,), P.dev, g), rnd((M, N), dt, P.dev, g), rndf((4, N), P.dev, g)
            for mode in ("plain", "bias_res_rowvec", "f32"):
                kw = dict(variant=variant, dual=(A2, B2, K2, pa, pb))
                out = torch.zeros(M, N, dtype=dt, device=P.dev)
                if mode == "bias_res_rowvec":
                    kw.update(bias=bias, res=R, ldres=N, rowvec=rv, rv_ld=N, rv_mod=4)
                elif mode == "f32":
                    kw.update(out_mode=K.OUT_F32, alpha=0.5)
                    out = torch.zeros(M, N, dtype=torch.float32, device=P.dev)
                o1, o2 = P.run("gemm", lambda o: ((A, B, o["C"], M, N, Kd, Kd, Kd, N), kw), dict(C=out))
                res.append((f"gemm v{variant} dual {M}x{N}x{Kd}+{K2} {mode}", relerr(o1["C"], o2["C"]), tol_for(dt)))
        # segmented A2 (fused q/k/v adapters): 3 segments of 320 (160-wide tiles) / 128 and 256 (128-wide tiles, variants 4 and 8)
        for (M, N, Kd, seg, var) in [(300, 960, 320, 320, variant), (200, 384, 128, 128, variant), (150, 768, 256, 256, 8)]:
            nseg = N // seg
            A, B = rnd((M, Kd), dt, P.dev, g), rnd((N, Kd), dt, P.dev, g, Kd ** -0.5)
            A2, B2 = rnd((M, nseg * 64), dt, P.dev, g), rnd((N, 64), dt, P.dev, g, 0.125)
            R = rnd((M, N), dt, P.dev, g)
            kw = dict(variant=var, dual=(A2, B2, 64, nseg * 64, 64, seg), res=R, ldres=N)
            o1, o2 = P.run("gemm", lambda o: ((A, B, o["C"], M, N, Kd, Kd, Kd, N), kw), dict(C=torch.zeros(M, N, dtype=dt, device=P.dev)))
            res.append((f"gemm v{var} dual segmented {M}x{N}x{Kd} seg={seg}", relerr(o1["C"], o2["C"]), tol_for(dt)))
    # split-K into a float scratch + finalize epilogue
    M, N, Kd = 200, 320, 1280
    A, B = rnd((M, Kd), dt, P.dev, g), rnd((N, Kd), dt, P.dev, g, Kd ** -0.5)
    bias, R, rv = rndf((N,), P.dev, g), rnd((M, N), dt, P.dev, g), rndf((4, N), P.dev, g)

    def splitk(be, o):
        acc = torch.zeros(3, M, N, device=P.dev)
        be.gemm(A, B, acc, M, N, Kd, Kd, Kd, N, out_mode=K.OUT_F32_SLAB, split_k=3, variant=variant)
        if o.dtype != torch.float32:
            be.gemm_finalize(acc, 3, M * N, o, M, N, N, bias=bias, rowvec=rv, rv_ld=N, rv_rpg=50, res=R, ldres=N)
        else:
            o.fill_(1.0)
            be.gemm_finalize(acc, 3, M * N, o, M, N, N, accumulate_f32=True, dtype=dt)
    c1, c2 = torch.zeros(M, N, dtype=dt, device=P.dev), torch.zeros(M, N, dtype=dt, device=P.dev)
    splitk(P.impl, c1)
    splitk(P.ref, c2)
    res.append((f"gemm v{variant} split-k + finalize", relerr(c1, c2), tol_for(dt)))
    f1, f2 = torch.zeros(M, N, device=P.dev), torch.zeros(M, N, device=P.dev)
    splitk(P.impl, f1)
    splitk(P.ref, f2)
    res.append((f"gemm v{variant} split-k + f32 accumulate finalize", relerr(f1, f2), tol_for(dt)))
    # strided operands / output views (fused qkv buffers)
    M, N, Kd = 256, 128, 128
    big = rnd((M, 3 * Kd), dt, P.dev, g)
    B = rnd((N, Kd), dt, P.dev, g, Kd ** -0.5)
    outb = torch.zeros(M, 3 * N, dtype=dt, device=P.dev)
    o1, o2 = P.run("gemm", lambda o: ((big[:, Kd:], B, o["C"][:, N:], M, N, Kd, 3 * Kd, Kd, 3 * N), dict(variant=variant)),
                   dict(C=outb))
    res.append((f"gemm v{variant} strided views", relerr(o1["C"], o2["C"]), tol_for(dt)))
    # 2 / 4 / 8 K slices: every slice on XCDs of its own (launch_gemm_v4 `z_xcd`) -- ragged tile grids, every slab element written once
    if variant >= 4:
        for (M2, N2, K2, sk) in [(700, 640, 512, 2), (700, 640, 512, 4), (330, 1280, 512, 8), (900, 1280, 256, 2), (90, 320, 512, 8)]:
            A2, B2 = rnd((M2, K2), dt, P.dev, g), rnd((N2, K2), dt, P.dev, g, K2 ** -0.5)
            o1, o2 = P.run("gemm", lambda o: ((A2, B2, o["C"], M2, N2, K2, K2, K2, N2), dict(variant=variant, out_mode=K.OUT_F32_SLAB, split_k=sk)),
                           dict(C=torch.full((sk, M2, N2), float("nan"), device=P.dev)))
            res.append((f"gemm v{variant} {M2}x{N2}x{K2} slabs x{sk} (slices on their own XCDs)", relerr(o1["C"], o2["C"]), tol_for(dt)))
    return res


def check_gemm_tn(P, dt, stages=0):
    g = torch.Generator().manual_seed(11)
    res = []
    for (R, N, Kd) in [(64, 128, 128), (200, 320, 64), (1000, 2560, 320), (77, 8, 1280), (560, 640, 640), (130, 256, 128)]:
        A, B = rnd((R, N), dt, P.dev, g), rnd((R, Kd), dt, P.dev, g, R ** -0.5)
        for mode, sk in ((K.OUT_F32, 1), (K.OUT_F32_ADD, 1), (K.OUT_F32_SLAB, 3)):
            if mode == K.OUT_F32_SLAB:
                rt_ = (R + 63) // 64
                if -(-rt_ // 3) * 2 >= rt_:          # a slice without rows: refused by the library (its slab would stay unwritten)
                    continue
                outs = dict(C=torch.zeros(3, N, Kd, device=P.dev))
            else:
                outs = dict(C=torch.ones(N, Kd, device=P.dev))
            # column sums of A: a running total when the reduction is not split, one stored row per slice when it is
            outs["cs"] = torch.ones(N, device=P.dev) if mode != K.OUT_F32_SLAB else torch.full((3, N), 7.0, device=P.dev)
            o1, o2 = P.run("gemm_tn", lambda o: ((A, B, o["C"], R, N, Kd, N, Kd, Kd),
                                                 dict(out_mode=mode, split_k=sk, a_colsum=o["cs"], stages=stages)), outs)
            res.append((f"gemm_tn s{stages} {R}x{N}x{Kd} mode={mode}", relerr(o1["C"], o2["C"]), tol_for(dt)))
            res.append((f"gemm_tn {R}x{N}x{Kd} mode={mode} colsum(A)", relerr(o1["cs"], o2["cs"]), 2e-3))
            if mode == K.OUT_F32_SLAB:
                slabs, cs = o2["C"], o2["cs"]
                o1, o2 = P.run("gemm_finalize", lambda o: ((slabs, 3, N * Kd, o["W"], N, Kd, Kd),
                                                           dict(accumulate_f32=2, dtype=dt, colsum_slabs=cs, colsum_out=o["b"])),
                               dict(W=torch.ones(N, Kd, device=P.dev), b=torch.ones(N, device=P.dev)))
                res.append((f"gemm_finalize {N}x{Kd} of 3 slabs + colsum rows", max(relerr(o1["W"], o2["W"]), relerr(o1["b"], o2["b"])), 1e-5))
    # the row-slice counts the host asks for (ops._tn_slices): 2 / 4 (a slice owns several XCDs, arranged over its tile grid) and
    # multiples of 8 (an XCD owns whole slices); rectangular outputs so that both arrangements of the XCDs occur
    for (R, N, Kd, sk) in [(2100, 320, 320, 2), (2100, 640, 128, 4), (2100, 128, 640, 4), (2038, 320, 320, 8), (4066, 320, 256, 16), (3052, 256, 320, 24)]:
        A, B = rnd((R, N), dt, P.dev, g), rnd((R, Kd), dt, P.dev, g, R ** -0.5)
        outs = dict(C=torch.zeros(sk, N, Kd, device=P.dev), cs=torch.full((sk, N), 7.0, device=P.dev))
        o1, o2 = P.run("gemm_tn", lambda o: ((A, B, o["C"], R, N, Kd, N, Kd, Kd), dict(out_mode=K.OUT_F32_SLAB, split_k=sk, a_colsum=o["cs"], stages=stages)), outs)
        res.append((f"gemm_tn s{stages} {R}x{N}x{Kd} {sk} slices", relerr(o1["C"].view(sk * N, Kd), o2["C"].view(sk * N, Kd)), tol_for(dt)))
        res.append((f"gemm_tn s{stages} {R}x{N}x{Kd} {sk} slices colsum(A)", relerr(o1["cs"], o2["cs"]), 2e-3))
    big = rnd((300, 3 * 128), dt, P.dev, g)
    X = rnd((300, 64), dt, P.dev, g)
    o1, o2 = P.run("gemm_tn", lambda o: ((big[:, 128:], X, o["C"], 300, 128, 64, 384, 64, 64), dict(out_mode=K.OUT_F32, stages=stages)),
                   dict(C=torch.zeros(128, 64, device=P.dev)))
    res.append(("gemm_tn strided A", relerr(o1["C"], o2["C"]), tol_for(dt)))
    # found_inf (GradScaler's inf check where the gradient is written): raised iff a value left in C is not finite, in the store and the
    # += form, by an inf that sits in ONE element of one operand row; left alone by finite operands
    for (R, N, Kd) in [(200, 320, 64), (130, 256, 384)]:
        A, B = rnd((R, N), dt, P.dev, g), rnd((R, Kd), dt, P.dev, g, R ** -0.5)
        Abad = A.clone()
        Abad[R // 2, N // 3] = float("inf")
        for mode in (K.OUT_F32, K.OUT_F32_ADD):
            for bad, Aop in ((0.0, A), (1.0, Abad)):
                flag = torch.zeros(4, device=P.dev)
                P.impl.gemm_tn(Aop, B, torch.ones(N, Kd, device=P.dev), R, N, Kd, N, Kd, Kd, out_mode=mode, stages=stages, found_inf=flag[3:4])
                if P.dev.type == "cuda":
                    torch.cuda.synchronize()
                res.append((f"gemm_tn s{stages} {R}x{N}x{Kd} mode={mode} found_inf with {'an inf' if bad else 'finite'} operand",
                            abs(float(flag[3]) - bad) + float(flag[:3].abs().sum()), 0.0))
    if stages == 0:
        # the same flag from the table-driven reduction, and svdx_check_finite_spans over a span table
        slabs = rndf((3, 64, 128), P.dev, g)
        slabs_bad = slabs.clone()
        slabs_bad[1, 5, 77] = float("nan")
        for bad, sl in ((0.0, slabs), (1.0, slabs_bad)):
            for store in (True, False):
                flag = torch.zeros(4, device=P.dev)
                P.impl.grad_finalize_batch([(sl, 3, 64 * 128, torch.ones(64, 128, device=P.dev), 64 * 128, None, None, store, flag[3:4])])
                if P.dev.type == "cuda":
                    torch.cuda.synchronize()
                res.append((f"grad_finalize_batch found_inf store={store} bad={bad}", abs(float(flag[3]) - bad) + float(flag[:3].abs().sum()), 0.0))
        buf = rndf((9000,), P.dev, g)
        spans = torch.tensor([[0, 64], [128, 4], [1000, 2048], [8996, 4]], dtype=torch.int32, device=P.dev)
        for idx, want in ((70, 0.0), (130, 1.0), (3047, 1.0), (3048, 0.0), (8999, 1.0), (8990, 0.0)):
            bb = buf.clone()
            bb[idx] = float("-inf")
            for be, tag in ((P.impl, "impl"), (P.ref, "emul")):
                st = torch.zeros(16, device=P.dev)
                be.check_finite_spans(bb, spans, 4, st)
                if P.dev.type == "cuda":
                    torch.cuda.synchronize()
                res.append((f"check_finite_spans [{tag}] inf at {idx}", abs(float(st[3]) - want) + float(st[:3].abs().sum()), 0.0))
        # table-driven float finalize (svdx_grad_finalize_batch): store and accumulate forms, with and without bias-gradient slabs,
        # bit-compared with one svdx_gemm_finalize launch each on the implementation under test
        g2 = torch.Generator().manual_seed(77)
        jobs, singles = [], []
        for (n, kd, sk, store, with_cs) in [(320, 320, 5, True, True), (960, 320, 3, False, False), (64, 1280, 2, True, True), (1280, 64, 7, False, True)]:
            slabs = rndf((sk, n, kd), P.dev, g2)
            cs = rndf((sk, n), P.dev, g2) if with_cs else None
            d0, c0 = rndf((n, kd), P.dev, g2), (rndf((n,), P.dev, g2) if with_cs else None)
            jobs.append((slabs, sk, n * kd, d0.clone(), n * kd, cs, None if c0 is None else c0.clone(), store))
            singles.append((slabs, sk, n, kd, d0.clone(), cs, None if c0 is None else c0.clone(), store))
        for be, tag in ((P.impl, "impl"), (P.ref, "emul")):
            jb = [(a, ns, st, d.clone(), cnt, cs, None if co is None else co.clone(), sto) for a, ns, st, d, cnt, cs, co, sto in jobs]
            be.grad_finalize_batch(jb)
            for (a, ns, st, d, cnt, cs, co, sto), (slabs, sk, n, kd, d1, cs1, c1, store) in zip(jb, singles):
                d1, c1 = d1.clone(), (None if c1 is None else c1.clone())
                be.gemm_finalize(slabs, sk, n * kd, d1, n, kd, kd, accumulate_f32=2 if store else 1, dtype=dt, colsum_slabs=cs1, colsum_out=c1)
                res.append((f"grad_finalize_batch [{tag}] {n}x{kd} split {sk} store={store}", float((d - d1).abs().max()), 0.0))
                if c1 is not None:
                    res.append((f"grad_finalize_batch [{tag}] {n}x{kd} colsum", float((co - c1).abs().max()), 0.0))
    return res


def check_gemm_geglu(P, dt, variant=4):
    g = torch.Generator().manual_seed(12)
    res = []
    for (M, C, F) in [(200, 64, 128), (300, 320, 1280), (130, 128, 640), (700, 192, 256)]:
        x, W1, b1 = rnd((M, C), dt, P.dev, g), rnd((2 * F, C), dt, P.dev, g, C ** -0.5), rndf((2 * F,), P.dev, g)
        o1, o2 = P.run("gemm", lambda o: ((x, W1, o["pre"], M, 2 * F, C, C, C, 2 * F),
                                          dict(bias=b1, variant=variant, epilogue=K.EPI_GEGLU_FWD, aux_out=o["h"], aux_dim=F)),
                       dict(pre=torch.zeros(M, 2 * F, dtype=dt, device=P.dev), h=torch.zeros(M, F, dtype=dt, device=P.dev)))
        res.append((f"gemm v{variant} geglu-fwd {M}x{C}x{F} pre", relerr(o1["pre"], o2["pre"]), tol_for(dt)))
        res.append((f"gemm v{variant} geglu-fwd {M}x{C}x{F} h", relerr(o1["h"], o2["h"]), tol_for(dt, 2)))
        pre = o2["pre"]
        dy, W2t = rnd((M, C), dt, P.dev, g), rnd((F, C), dt, P.dev, g, C ** -0.5)
        o1, o2 = P.run("gemm", lambda o: ((dy, W2t, o["dpre"], M, F, C, C, C, 2 * F),
                                          dict(variant=variant, epilogue=K.EPI_GEGLU_BWD, aux_in=pre, aux_dim=F)),
                       dict(dpre=torch.zeros(M, 2 * F, dtype=dt, device=P.dev)))
        res.append((f"gemm v{variant} geglu-bwd {M}x{C}x{F}", relerr(o1["dpre"], o2["dpre"]), tol_for(dt, 2)))
    return res


def check_gemm_gather(P, dt, variant):
    g = torch.Generator().manual_seed(2)
    res = []
    cases = []
    # (label, Gather, M, cin, cout, nsrc_rows)
    n, h, w, ci, co = 3, 10, 12, 64, 96
    cases.append(("conv3x3 s1", K.Gather(K.GATHER_CONV3X3, n_img=n, hi=h, wi=w, ho=h, wo=w, cin=ci, stride=1, lda=ci),
                  n * h * w, ci, co, n * h * w, 9))
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    cases.append(("conv3x3 s2", K.Gather(K.GATHER_CONV3X3, n_img=n, hi=h, wi=w, ho=ho, wo=wo, cin=ci, stride=2, lda=ci),
                  n * ho * wo, ci, co, n * h * w, 9))
    cases.append(("conv3x3 ups", K.Gather(K.GATHER_CONV3X3, n_img=n, hi=2 * h, wi=2 * w, ho=2 * h, wo=2 * w, cin=ci, stride=1,
                                          ups=1, lda=ci), n * 4 * h * w, ci, co, n * h * w, 9))
    cases.append(("conv3x3 dgrad2", K.Gather(K.GATHER_CONV3X3_DGRAD2, n_img=n, hi=ho, wi=wo, ho=h, wo=w, cin=ci, lda=ci),
                  n * h * w, ci, co, n * ho * wo, 9))
    Bc, T, hw = 2, 5, 37
    cases.append(("temporal3", K.Gather(K.GATHER_TEMPORAL3, n_img=Bc, cin=ci, t=T, hw=hw, lda=ci), Bc * T * hw, ci, co,
                  Bc * T * hw, 3))
    cases.append(("temporal3 T=1", K.Gather(K.GATHER_TEMPORAL3, n_img=3, cin=ci, t=1, hw=hw, lda=ci), 3 * hw, ci, co, 3 * hw, 3))
    cases.append(("conv3x3 s1 cout=320", K.Gather(K.GATHER_CONV3X3, n_img=n, hi=h, wi=w, ho=h, wo=w, cin=ci, stride=1, lda=ci),
                  n * h * w, ci, 320, n * h * w, 9))
    cases.append(("temporal3 cout=160", K.Gather(K.GATHER_TEMPORAL3, n_img=Bc, cin=ci, t=T, hw=hw, lda=ci), Bc * T * hw, ci, 160,
                  Bc * T * hw, 3))
    for label, ga, M, ci_, co_, nsrc, taps in cases:
        A = rnd((nsrc, ci_), dt, P.dev, g)
        B = rnd((co_, taps * ci_), dt, P.dev, g, (taps * ci_) ** -0.5)
        bias = rndf((co_,), P.dev, g)
        out = torch.zeros(M, co_, dtype=dt, device=P.dev)
        o1, o2 = P.run("gemm", lambda o: ((A, B, o["C"], M, co_, taps * ci_, ci_, taps * ci_, co_),
                                          dict(bias=bias, gather=ga, variant=variant)), dict(C=out))
        res.append((f"gemm v{variant} {label}", relerr(o1["C"], o2["C"]), tol_for(dt)))
        if 2 <= variant < 16 and label in ("conv3x3 s1", "temporal3"):       # taps, then the plain second operand pair
            A2, B2 = rnd((M, 64), dt, P.dev, g), rnd((co_, 64), dt, P.dev, g, 0.125)
            o1, o2 = P.run("gemm", lambda o: ((A, B, o["C"], M, co_, taps * ci_, ci_, taps * ci_, co_),
                                              dict(bias=bias, gather=ga, variant=variant, dual=(A2, B2, 64, 64, 64))), dict(C=out))
            res.append((f"gemm v{variant} {label} + dual", relerr(o1["C"], o2["C"]), tol_for(dt)))
    # stride-1 3x3 convolutions at the image widths of every UNet level: several channel slices, tiles that straddle image / batch
    # borders, residual epilogue, and split-K into float slabs
    for (n_, h_, w_, ci_, co_) in [(2, 20, 64, 192, 320), (3, 9, 16, 128, 160), (1, 5, 8, 64, 320), (2, 7, 32, 320, 160)]:
        ga = K.Gather(K.GATHER_CONV3X3, n_img=n_, hi=h_, wi=w_, ho=h_, wo=w_, cin=ci_, stride=1, lda=ci_)
        M = n_ * h_ * w_
        A = rnd((M, ci_), dt, P.dev, g)
        B = rnd((co_, 9 * ci_), dt, P.dev, g, (9 * ci_) ** -0.5)
        bias, R = rndf((co_,), P.dev, g), rnd((M, co_), dt, P.dev, g)
        o1, o2 = P.run("gemm", lambda o: ((A, B, o["C"], M, co_, 9 * ci_, ci_, 9 * ci_, co_),
                                          dict(bias=bias, res=R, ldres=co_, gather=ga, variant=variant)),
                       dict(C=torch.zeros(M, co_, dtype=dt, device=P.dev)))
        res.append((f"gemm v{variant} conv3x3 {n_}x{h_}x{w_} {ci_}->{co_} bias+res", relerr(o1["C"], o2["C"]), tol_for(dt)))
        sk = ci_ // 64
        if sk > 1:
            o1, o2 = P.run("gemm", lambda o: ((A, B, o["C"], M, co_, 9 * ci_, ci_, 9 * ci_, co_),
                                              dict(gather=ga, variant=variant, out_mode=K.OUT_F32_SLAB, split_k=sk)),
                           dict(C=torch.zeros(sk, M, co_, device=P.dev)))
            res.append((f"gemm v{variant} conv3x3 {n_}x{h_}x{w_} {ci_}->{co_} slabs x{sk}",
                        relerr(o1["C"].sum(0), o2["C"].sum(0)), 1e-3))
    return res


def same_or_err(xs, ys):
    """0.0 when the two lists of tensors are bit-identical, else their largest relative difference"""
    if all(torch.equal(a, b) for a, b in zip(xs, ys)):
        return 0.0
    return max(max(relerr(a.float(), b.float()) for a, b in zip(xs, ys)), 1e-30)


def EXACT_TOL(P):
    """A table-driven launch against its single-job launches: bit equality on the simulator (same device function, no contraction);
    on hardware the two kernels are separate compilations of that function and are held to rounding level."""
    return 0.0 if P.dev.type == "cpu" else 2e-6


def check_small(P, dt):
    g = torch.Generator().manual_seed(3)
    res = []
    for (M, N, Kd) in [(1, 1280, 320), (2, 96, 64), (14, 320, 1280), (25, 640, 96)]:
        X, W, b = rndf((M, Kd), P.dev, g), rnd((N, Kd), dt, P.dev, g, Kd ** -0.5), rndf((N,), P.dev, g)
        for silu_in, acc in ((0, 0), (1, 1)):
            o1, o2 = P.run("small_linear", lambda o: ((X, W, b, o["Y"], M, N, Kd, Kd), dict(trans=0, silu_in=silu_in, accumulate=acc)),
                           dict(Y=torch.ones(M, N, device=P.dev)))
            res.append((f"small_linear nt {M}x{N}x{Kd} silu={silu_in} acc={acc}", relerr(o1["Y"], o2["Y"]), 1e-4))
        dY = rndf((M, N), P.dev, g)
        o1, o2 = P.run("small_linear", lambda o: ((dY, W, None, o["Y"], M, N, Kd, Kd), dict(trans=1)),
                       dict(Y=torch.zeros(M, Kd, device=P.dev)))
        res.append((f"small_linear nn {M}x{N}x{Kd}", relerr(o1["Y"], o2["Y"]), 1e-4))
        o1, o2 = P.run("outer_acc", lambda o: ((dY, X, o["W"], M, N, Kd), dict(scale=0.5)), dict(W=torch.ones(N, Kd, device=P.dev)))
        res.append((f"outer_acc {M}x{N}x{Kd}", relerr(o1["W"], o2["W"]), 1e-4))
    # table-driven launches (svdx_small_linear_batch / svdx_outer_acc_batch): every job must equal its single-job launch BIT FOR BIT
    # (the deferred cross-attention chain of the step relies on it) and the emulation to tolerance; 50 jobs also cross the 48-job pack
    for M, shapes in ((1, [(320, 1024), (640, 1024), (1280, 1024), (1280, 320), (96, 64), (320, 320)]), (3, [(640, 1024), (64, 64), (1280, 1280)]),
                      (6, [(96, 64)] * 50)):
        Xs = [rndf((M, Kd), P.dev, g) for N, Kd in shapes]
        Ws = [rnd((N, Kd), dt, P.dev, g, Kd ** -0.5) for N, Kd in shapes]
        bs = [rndf((N,), P.dev, g) if i % 2 == 0 else None for i, (N, Kd) in enumerate(shapes)]
        dYs = [rndf((M, N), P.dev, g) for N, Kd in shapes]
        fl = [(i % 3 == 1, i % 2 == 1) for i in range(len(shapes))]           # (silu_in, accumulate)
        for trans in (0, 1):
            single = [torch.ones(M, Kd if trans else N, device=P.dev) for N, Kd in shapes]
            batch = [t.clone() for t in single]
            refo = [t.clone() for t in single]
            for i, (N, Kd) in enumerate(shapes):
                if trans == 0:
                    P.impl.small_linear(Xs[i], Ws[i], bs[i], single[i], M, N, Kd, Kd, 0, int(fl[i][0]), int(fl[i][1]))
                else:
                    P.impl.small_linear(dYs[i], Ws[i], None, single[i], M, N, Kd, Kd, 1, 0, int(fl[i][1]))
            jobs = lambda outs: [((Xs[i], Ws[i], bs[i], outs[i], N, Kd, Kd, fl[i][0], fl[i][1]) if trans == 0 else
                                  (dYs[i], Ws[i], None, outs[i], N, Kd, Kd, False, fl[i][1])) for i, (N, Kd) in enumerate(shapes)]
            P.impl.small_linear_batch(jobs(batch), M, trans)
            P.ref.small_linear_batch(jobs(refo), M, trans)
            if P.dev.type == "cuda":
                torch.cuda.synchronize()
            res.append((f"small_linear_batch trans={trans} M={M} x{len(shapes)} == single launches", same_or_err(single, batch), EXACT_TOL(P)))
            res.append((f"small_linear_batch trans={trans} M={M} x{len(shapes)}", max(relerr(a, b) for a, b in zip(batch, refo)), 1e-4))
        single = [torch.ones(N, Kd if i % 3 else 1, device=P.dev) for i, (N, Kd) in enumerate(shapes)]
        batch = [t.clone() for t in single]
        refo = [t.clone() for t in single]
        ones = torch.ones(M, 1, device=P.dev)
        for i, (N, Kd) in enumerate(shapes):
            if i % 3:
                P.impl.outer_acc(dYs[i], Xs[i], single[i], M, N, Kd, 0.5)
            else:
                P.impl.outer_acc(dYs[i], ones, single[i], M, N, 1, 0.5)
        jobs = lambda outs: [(dYs[i], Xs[i] if i % 3 else None, outs[i], N, Kd if i % 3 else 1, 0.5) for i, (N, Kd) in enumerate(shapes)]
        P.impl.outer_acc_batch(jobs(batch), M)
        P.ref.outer_acc_batch(jobs(refo), M)
        if P.dev.type == "cuda":
            torch.cuda.synchronize()
        res.append((f"outer_acc_batch M={M} x{len(shapes)} == single launches", same_or_err(single, batch), EXACT_TOL(P)))
        res.append((f"outer_acc_batch M={M} x{len(shapes)}", max(relerr(a, b) for a, b in zip(batch, refo)), 1e-4))
    t = torch.tensor([0.0, 0.31, -1.7, 127.0, 7.0, 24.0], device=P.dev)
    for dim in (320, 256, 64):
        o1, o2 = P.run("timestep_embed", lambda o: ((t, o["E"], 6, dim), {}), dict(E=torch.zeros(6, dim, device=P.dev)))
        res.append((f"timestep_embed dim={dim}", float((o1["E"] - o2["E"]).abs().max()), 2e-4))
    return res


def check_gemm_gn(P, dt, variant):
    """svdx_gemm_gn: the GEMM result against the emulation, and the GroupNorm statistics it leaves against a statistics pass (emulated)
    over the tensor THE LAUNCH ITSELF wrote -- same rounded values, so only the fp32 summation order differs."""
    from svd_xtend_amd.ops import TILE_OF_VARIANT, STAGED_TILES, _tile_launched, gn_tile_ok
    g = torch.Generator().manual_seed(11)
    res = []
    # (samples, rows per sample, N, channels per group, K, gather?)   -- 2-D norms over frames shorter / longer than a tile, a clip-wide norm
    cases = [(6, 40, 320, 10, 128), (2, 400, 640, 20, 192), (1, 700, 1280, 40, 64), (14, 160, 1280, 40, 128), (3, 96, 640, 20, 64), (2, 2560, 320, 10, 64)]
    for (n_s, rows, N, cg, Kd) in cases:
        M, G = n_s * rows, N // cg
        tile = _tile_launched(variant, M, N)
        if not gn_tile_ok(tile, N, rows, cg):
            continue
        A, B = rnd((M, Kd), dt, P.dev, g), rnd((N, Kd), dt, P.dev, g, Kd ** -0.5)
        bias, R = rndf((N,), P.dev, g), rnd((M, N), dt, P.dev, g)
        rv = rndf((n_s, N), P.dev, g)
        for mode in ("plain", "bias_res", "rowvec"):
            kw = dict(variant=variant)
            if mode == "bias_res":
                kw.update(bias=bias, res=R, ldres=N)
            elif mode == "rowvec":
                kw.update(bias=bias, rowvec=rv, rv_ld=N, rv_rpg=rows)
            outs = dict(C=torch.zeros(M, N, dtype=dt, device=P.dev), st=torch.zeros(K.GN_REPLICAS, n_s, G, K.GN_STAT_FLOATS, device=P.dev))
            o1, o2 = P.run("gemm", lambda o: ((A, B, o["C"], M, N, Kd, Kd, Kd, N), dict(kw, gn=(o["st"], rows, cg))), outs)
            res.append((f"gemm_gn v{variant} {n_s}x{rows}x{N} cg={cg} K={Kd} {mode} C", relerr(o1["C"], o2["C"]), tol_for(dt)))
            own = torch.zeros_like(outs["st"])
            P.ref.gn_stats(o1["C"], own, n_s, rows, N, G, prezeroed=1)
            res.append((f"gemm_gn v{variant} {n_s}x{rows}x{N} cg={cg} K={Kd} {mode} stats",
                        relerr(emul.gn_decode(o1["st"], n_s, G, rows * cg, 0).view(-1, 2), emul.gn_decode(own, n_s, G, rows * cg, 0).view(-1, 2)), 1e-4))
    # split-K form: the statistics come from the reducing launch
    for (n_s, rows, N, cg, Kd, sk) in [(14, 40, 1280, 40, 384, 3), (3, 160, 640, 20, 128, 2), (1, 300, 320, 10, 128, 2), (5, 8, 1280, 40, 128, 2)]:
        M, G = n_s * rows, N // cg
        A, B = rnd((M, Kd), dt, P.dev, g), rnd((N, Kd), dt, P.dev, g, Kd ** -0.5)
        bias, R = rndf((N,), P.dev, g), rnd((M, N), dt, P.dev, g)

        def splitk(be, o, st):
            acc = torch.zeros(sk, M, N, device=P.dev)
            be.gemm(A, B, acc, M, N, Kd, Kd, Kd, N, out_mode=K.OUT_F32_SLAB, split_k=sk, variant=variant)
            be.gemm_finalize(acc, sk, M * N, o, M, N, N, bias=bias, res=R, ldres=N, gn=(st, rows, cg))
        c1, c2 = torch.zeros(M, N, dtype=dt, device=P.dev), torch.zeros(M, N, dtype=dt, device=P.dev)
        s1, s2 = (torch.zeros(K.GN_REPLICAS, n_s, G, K.GN_STAT_FLOATS, device=P.dev) for _ in range(2))
        splitk(P.impl, c1, s1)
        splitk(P.ref, c2, s2)
        res.append((f"gemm_finalize_gn v{variant} {n_s}x{rows}x{N} split {sk} C", relerr(c1, c2), tol_for(dt)))
        own = torch.zeros_like(s1)
        P.ref.gn_stats(c1, own, n_s, rows, N, G, prezeroed=1)
        res.append((f"gemm_finalize_gn v{variant} {n_s}x{rows}x{N} split {sk} stats",
                    relerr(emul.gn_decode(s1, n_s, G, rows * cg, 0).view(-1, 2), emul.gn_decode(own, n_s, G, rows * cg, 0).view(-1, 2)), 1e-4))
    return res


def check_groupnorm(P, dt):
    g = torch.Generator().manual_seed(4)
    res = []
    # last case: a clip-wide norm over 2^18 rows of small activations (|x| ~ 1e-2): the fixed-point sum of squares must keep them
    for (n_s, rows, C, mag) in [(3, 70, 64, 1.5), (2, 200, 320, 1.5), (1, 333, 960, 1.5), (2, 64, 2560, 1.5), (5, 16, 192, 1.5), (1, 4000, 320, 1.5),
                                (1, 262144, 64, 0.01)]:
        x = (rnd((n_s * rows, C), dt, P.dev, g) * mag + 0.2 * mag).to(dt)
        dy = rnd((n_s * rows, C), dt, P.dev, g)
        add = rnd((n_s * rows, C), dt, P.dev, g)
        gamma, beta = 1 + 0.1 * rndf((C,), P.dev, g), 0.1 * rndf((C,), P.dev, g)
        cnt = rows * (C // 32)
        st = torch.zeros(K.GN_REPLICAS, n_s, 32, K.GN_STAT_FLOATS, device=P.dev)
        o1, o2 = P.run("gn_stats", lambda o: ((x, o["st"], n_s, rows, C, 32), {}), dict(st=st))
        res.append((f"gn_stats {n_s}x{rows}x{C}", relerr(emul.gn_decode(o1["st"], n_s, 32, cnt, 0).view(-1, 2),
                                                          emul.gn_decode(o2["st"], n_s, 32, cnt, 0).view(-1, 2)), 1e-4))
        stats = o2["st"]
        for silu in (0, 1):
            o1, o2 = P.run("gn_apply", lambda o: ((x, stats, gamma, beta, o["y"], n_s, rows, C, 32, 1e-5, silu), {}),
                           dict(y=torch.zeros_like(x)))
            res.append((f"gn_apply {n_s}x{rows}x{C} silu={silu}", relerr(o1["y"], o2["y"]), tol_for(dt)))
            o1, o2 = P.run("gn_bwd_stats", lambda o: ((dy, x, stats, gamma, beta, o["b"], n_s, rows, C, 32, 1e-5, silu), {}),
                           dict(b=torch.zeros(K.GN_REPLICAS, n_s, 32, K.GN_STAT_FLOATS, device=P.dev)))
            res.append((f"gn_bwd_stats {n_s}x{rows}x{C} silu={silu}", relerr(emul.gn_decode(o1["b"], n_s, 32, cnt, 1).view(-1, 2),
                                                                            emul.gn_decode(o2["b"], n_s, 32, cnt, 1).view(-1, 2)), 2e-3))
            bst = o2["b"]
            for ad in (None, add):
                o1, o2 = P.run("gn_bwd_apply", lambda o: ((dy, x, stats, bst, gamma, beta, ad, o["dx"], n_s, rows, C, 32, 1e-5, silu), {}),
                               dict(dx=torch.zeros_like(x)))
                res.append((f"gn_bwd_apply {n_s}x{rows}x{C} silu={silu} add={ad is not None}", relerr(o1["dx"], o2["dx"]), tol_for(dt)))
    return res


def check_layernorm(P, dt):
    g = torch.Generator().manual_seed(5)
    res = []
    for (rows, C) in [(100, 64), (777, 320), (130, 640), (50, 1280), (9, 128), (9001, 320), (3000, 640)]:
        x = (rnd((rows, C), dt, P.dev, g) * 2 + 0.5).to(dt)
        dy, add = rnd((rows, C), dt, P.dev, g), rnd((rows, C), dt, P.dev, g)
        gamma, beta = 1 + 0.1 * rndf((C,), P.dev, g), 0.1 * rndf((C,), P.dev, g)
        o1, o2 = P.run("ln_fwd", lambda o: ((x, gamma, beta, o["y"], o["st"], rows, C, 1e-5), {}),
                       dict(y=torch.zeros_like(x), st=torch.zeros(rows, 2, device=P.dev)))
        res.append((f"ln_fwd {rows}x{C} y", relerr(o1["y"], o2["y"]), tol_for(dt)))
        res.append((f"ln_fwd {rows}x{C} stats", relerr(o1["st"], o2["st"]), 1e-4))
        st = o2["st"]
        for affine in (False, True, "scratch"):
            outs = dict(dx=torch.zeros_like(x), dg=torch.ones(C, device=P.dev), db=torch.ones(C, device=P.dev))
            scr = torch.full((K.LN_PARTIAL_ROWS * 2 * C,), float("nan"), device=P.dev) if affine == "scratch" else None
            o1, o2 = P.run("ln_bwd", lambda o: ((dy, x, st, gamma, add if affine else None, o["dx"],
                                                 o["dg"] if affine else None, o["db"] if affine else None, rows, C),
                                                dict(scratch=scr, add2=dy if affine == "scratch" else None, add2_scale=0.37)), outs)
            res.append((f"ln_bwd {rows}x{C} affine={affine} dx", relerr(o1["dx"], o2["dx"]), tol_for(dt)))
            if affine:
                res.append((f"ln_bwd {rows}x{C} affine={affine} dgamma", relerr(o1["dg"], o2["dg"]), 2e-3))
                res.append((f"ln_bwd {rows}x{C} affine={affine} dbeta", relerr(o1["db"], o2["db"]), 2e-3))
    # deferred affine-gradient reduction: svdx_ln_bwd(defer_reduce) x 5 + ONE svdx_ln_param_reduce_batch == the five immediate forms, bit for bit
    cases = [(100, 64), (777, 320), (3000, 640), (50, 1280), (9001, 320)]
    imm, dfr, jobs, keep = [], [], [], []
    for (rows, C) in cases:
        x = (rnd((rows, C), dt, P.dev, g) * 2 + 0.5).to(dt)
        dy = rnd((rows, C), dt, P.dev, g)
        gamma, beta = 1 + 0.1 * rndf((C,), P.dev, g), 0.1 * rndf((C,), P.dev, g)
        y, st = torch.zeros_like(x), torch.zeros(rows, 2, device=P.dev)
        P.impl.ln_fwd(x, gamma, beta, y, st, rows, C, 1e-5)
        a = dict(dx=torch.zeros_like(x), dg=torch.ones(C, device=P.dev), db=torch.ones(C, device=P.dev))
        b = {k_: v.clone() for k_, v in a.items()}
        P.impl.ln_bwd(dy, x, st, gamma, None, a["dx"], a["dg"], a["db"], rows, C,
                      scratch=torch.full((K.LN_PARTIAL_ROWS * 2 * C,), float("nan"), device=P.dev))
        nblk = K.ln_bwd_blocks(rows, C)
        scr = torch.full((nblk * 2 * C,), float("nan"), device=P.dev)
        P.impl.ln_bwd(dy, x, st, gamma, None, b["dx"], b["dg"], b["db"], rows, C, scratch=scr, defer_reduce=True)
        jobs.append((scr, b["dg"], b["db"], nblk, C))
        imm.append(a)
        dfr.append(b)
        keep.append((x, dy, st))
    P.impl.ln_param_reduce_batch(jobs)
    if P.dev.type == "cuda":
        torch.cuda.synchronize()
    res.append(("ln_bwd defer_reduce + ln_param_reduce_batch == immediate reduction",
                same_or_err([a[k_] for a in imm for k_ in ("dx", "dg", "db")], [b[k_] for b in dfr for k_ in ("dx", "dg", "db")]), EXACT_TOL(P)))
    if hasattr(P.impl, "lib"):
        ok = all(P.impl.lib.svdx_ln_bwd_blocks(r, c) == K.ln_bwd_blocks(r, c) for r in (1, 9, 100, 777, 9001, 35840, 10 ** 6) for c in (64, 320, 640, 768, 1280))
        res.append(("svdx_ln_bwd_blocks == kernels.ln_bwd_blocks", 0.0 if ok else 1.0, 0.0))
    return res


def check_attention(P, dt):
    g = torch.Generator().manual_seed(6)
    res = []
    # 2560 = the spatial sequence of the benched c2 shape (latent 40 x 64), 9216 = config 4's (72 x 128)
    # (3, 5, 136): 15 (head, sample) pairs -- more than the 8 XCDs the pairs are dealt to, not a multiple of 8, two row tiles with a partial one
    for (nb, heads, S) in [(2, 2, 40), (1, 5, 160), (3, 1, 200), (1, 2, 640), (1, 1, 16), (3, 5, 136), (2, 2, 2560), (1, 1, 9216)]:
        C = heads * 64
        qkv = rnd((nb * S, 3 * C), dt, P.dev, g, 1.0)
        if S >= 160:
            # force the online-softmax rescale branch: keys grow along the sequence, so the running max jumps by far more than the
            # deferred-rescale threshold at later KV tiles (cdna_hip_programming.md rule 26)
            ramp = torch.linspace(0.3, 4.0, S, device=P.dev).repeat(nb)[:, None]
            qkv[:, C:2 * C] = (qkv[:, C:2 * C].float() * ramp).to(dt)
        d_o = rnd((nb * S, C), dt, P.dev, g)
        q, k, v = qkv, qkv[:, C:], qkv[:, 2 * C:]
        scale = 0.125
        o1, o2 = P.run("attn_fwd", lambda o: ((q, k, v, o["o"], o["lse"], nb, heads, S, 3 * C, C, scale), {}),
                       dict(o=torch.zeros(nb * S, C, dtype=dt, device=P.dev), lse=torch.zeros(nb * heads * S, device=P.dev)))
        res.append((f"attn_fwd nb={nb} h={heads} S={S} o", relerr(o1["o"], o2["o"]), tol_for(dt, 2)))
        res.append((f"attn_fwd nb={nb} h={heads} S={S} o 1-cos(rows)", 1.0 - cos_rows_min(o1["o"], o2["o"]), 1e-4 if dt == torch.float16 else 2e-3))
        res.append((f"attn_fwd nb={nb} h={heads} S={S} lse", float((o1["lse"] - o2["lse"]).abs().max()), 2e-2))
        o_ref, lse = o2["o"], o2["lse"]
        o1, o2 = P.run("attn_bwd_prep", lambda o: ((o_ref, d_o, o["D"], nb, heads, S, C), {}), dict(D=torch.zeros(nb * heads * S, device=P.dev)))
        res.append((f"attn_bwd_prep nb={nb} h={heads} S={S}", relerr(o1["D"], o2["D"]), 1e-3))
        D = o2["D"]
        dqkv = torch.zeros(nb * S, 3 * C, dtype=dt, device=P.dev)
        o1, o2 = P.run("attn_bwd_dkv", lambda o: ((q, k, v, d_o, lse, D, o["d"][:, C:], o["d"][:, 2 * C:],
                                                   nb, heads, S, 3 * C, C, 3 * C, scale), {}), dict(d=dqkv))
        res.append((f"attn_bwd_dkv nb={nb} h={heads} S={S} dk", relerr(o1["d"][:, C:2 * C], o2["d"][:, C:2 * C]), tol_for(dt, 4)))
        res.append((f"attn_bwd_dkv nb={nb} h={heads} S={S} dv", relerr(o1["d"][:, 2 * C:], o2["d"][:, 2 * C:]), tol_for(dt, 4)))
        o1, o2 = P.run("attn_bwd_dq", lambda o: ((q, k, v, d_o, lse, D, o["d"], nb, heads, S, 3 * C, C, 3 * C, scale), {}),
                       dict(d=dqkv))
        res.append((f"attn_bwd_dq nb={nb} h={heads} S={S}", relerr(o1["d"][:, :C], o2["d"][:, :C]), tol_for(dt, 4)))
        res.append((f"attn_bwd_dq nb={nb} h={heads} S={S} 1-cos(rows)", 1.0 - cos_rows_min(o1["d"][:, :C], o2["d"][:, :C]),
                    1e-3 if dt == torch.float16 else 1e-2))
    return res


def check_temporal_attention(P, dt):
    g = torch.Generator().manual_seed(7)
    res = []
    for (B, T, HW, heads) in [(1, 14, 9, 2), (2, 25, 5, 1), (1, 3, 16, 5), (1, 16, 4, 1), (1, 1, 4, 1)]:
        C = heads * 64
        M = B * T * HW
        qkv = rnd((M, 3 * C), dt, P.dev, g)
        d_o = rnd((M, C), dt, P.dev, g)
        q, k, v = qkv, qkv[:, C:], qkv[:, 2 * C:]
        o1, o2 = P.run("tattn_fwd", lambda o: ((q, k, v, o["o"], B, T, HW, heads, 3 * C, C, 0.125), {}),
                       dict(o=torch.zeros(M, C, dtype=dt, device=P.dev)))
        res.append((f"tattn_fwd B={B} T={T} HW={HW} h={heads}", relerr(o1["o"], o2["o"]), tol_for(dt)))
        o1, o2 = P.run("tattn_bwd", lambda o: ((q, k, v, d_o, o["d"], o["d"][:, C:], o["d"][:, 2 * C:], B, T, HW, heads, 3 * C, C,
                                                3 * C, 0.125), {}), dict(d=torch.zeros(M, 3 * C, dtype=dt, device=P.dev)))
        for i, nm in enumerate(("dq", "dk", "dv")):
            res.append((f"tattn_bwd B={B} T={T} HW={HW} h={heads} {nm}",
                        relerr(o1["d"][:, i * C:(i + 1) * C], o2["d"][:, i * C:(i + 1) * C]), tol_for(dt, 2)))
    return res


def check_tsa(P, dt):
    """svdx_tsa_fwd (LayerNorm -> q/k/v -> attention over frames -> out-projection + bias + row vector + residual in one launch)
    against the emulation of the four launches it replaces.  Shapes: the benched 64x40 level (T = 14, HW = 2560, C = 320: 10 pixels
    per band, 256 bands), partial bands, T = 16 / 3 / 1, B = 2 with the rv_mod grouping, every C the kernel admits."""
    g = torch.Generator().manual_seed(17)
    res = []
    for (B, T, HW, heads, mod) in [(1, 14, 2560, 5, 0), (2, 14, 36, 5, 2), (1, 16, 9, 1, 0), (1, 3, 48, 2, 0), (2, 5, 7, 4, 0), (1, 1, 16, 3, 0)]:
        C = heads * 64
        M = B * T * HW
        x = rnd((M, C), dt, P.dev, g)
        x[:, :8] += 3.0                                     # a mean the LayerNorm has to remove
        gamma, beta = 1.0 + 0.1 * rndf((C,), P.dev, g), 0.1 * rndf((C,), P.dev, g)
        wqkv, wo = rnd((3 * C, C), dt, P.dev, g, C ** -0.5), rnd((C, C), dt, P.dev, g, C ** -0.5)
        wqkv[:C] *= 2.0                                     # peaked softmax rows
        bo, cvec = 0.1 * rndf((C,), P.dev, g), rndf((B, C), P.dev, g)
        rpg = 0 if mod else T * HW
        outs = dict(n1=torch.zeros(M, C, dtype=dt, device=P.dev), st=torch.zeros(M, 2, device=P.dev),
                    qkv=torch.zeros(M, 3 * C, dtype=dt, device=P.dev), o=torch.zeros(M, C, dtype=dt, device=P.dev),
                    h1=torch.zeros(M, C, dtype=dt, device=P.dev))
        o1, o2 = P.run("tsa_fwd", lambda o: ((x, gamma, beta, 1e-5, wqkv, wo, bo, cvec, C, rpg, mod, o["n1"], o["st"], o["qkv"], o["o"],
                                              o["h1"], B, T, HW, C, heads, 0.125), {}), outs)
        tag = f"tsa_fwd B={B} T={T} HW={HW} C={C}"
        res.append((f"{tag} n1", relerr(o1["n1"], o2["n1"]), tol_for(dt)))
        res.append((f"{tag} stats", relerr(o1["st"], o2["st"]), 1e-4))
        res.append((f"{tag} qkv", relerr(o1["qkv"], o2["qkv"]), tol_for(dt, 2)))
        res.append((f"{tag} o", relerr(o1["o"], o2["o"]), tol_for(dt, 2)))
        res.append((f"{tag} h1", relerr(o1["h1"], o2["h1"]), tol_for(dt, 2)))
        res.append((f"{tag} h1 1-cos(rows)", 1.0 - cos_rows_min(o1["h1"], o2["h1"]), 1e-4 if dt == torch.float16 else 2e-3))
    return res


def check_encoders(P, dt):
    """svdx_patch_rows / svdx_softmax_rows / svdx_act_rows / svdx_attn_small_fwd (csrc/encoders.hip) and the pad-0 stride-2 gather of the VAE downsample."""
    g = torch.Generator().manual_seed(18)
    res = []
    for (n, C, H, W, kh, st, pad, ldk) in [(2, 3, 16, 24, 3, 1, 1, 64), (1, 3, 28, 42, 14, 14, 0, 640), (3, 3, 9, 7, 3, 1, 1, 32)]:
        ho, wo = (H + 2 * pad - kh) // st + 1, (W + 2 * pad - kh) // st + 1
        img = rndf((n, C, H, W), P.dev, g)
        o1, o2 = P.run("patch_rows", lambda o: ((img, o["y"], n, C, H, W, kh, kh, st, pad, ho, wo, ldk, 0.5), {}),
                       dict(y=torch.ones(n * ho * wo, ldk, dtype=dt, device=P.dev)))
        res.append((f"patch_rows {n}x{C}x{H}x{W} k{kh} s{st}", relerr(o1["y"], o2["y"]), tol_for(dt, 0.5)))
    for (rows, cols, cols_out) in [(5, 2560, 2560), (7, 257, 320), (3, 15, 64), (2, 8, 8)]:
        x = rnd((rows, cols_out), dt, P.dev, g, 3.0)
        o1, o2 = P.run("softmax_rows", lambda o: ((x, o["y"], rows, cols, cols_out, cols_out, cols_out, 0.7), {}),
                       dict(y=torch.ones(rows, cols_out, dtype=dt, device=P.dev)))
        res.append((f"softmax_rows {rows}x{cols}->{cols_out}", relerr(o1["y"], o2["y"]), tol_for(dt)))
    x = rnd((8 * 333,), dt, P.dev, g, 2.0)
    for act in (0, 1):
        o1, o2 = P.run("act_rows", lambda o: ((x, o["y"], x.numel(), act), {}), dict(y=torch.zeros_like(x)))
        res.append((f"act_rows act={act}", relerr(o1["y"], o2["y"]), tol_for(dt)))
    # the CLIP tower's attention (257 tokens, heads of 80 channels) and ragged shapes either side of its 16-query blocks / 4-key P.V groups
    for (n, S, heads, d) in [(1, 257, 2, 80), (2, 37, 3, 64), (1, 5, 1, 128), (1, 66, 2, 16)]:
        qkv = rnd((n * S, 3 * heads * d), dt, P.dev, g, 1.5)
        o1, o2 = P.run("attn_small_fwd", lambda o: ((qkv, o["y"], n, S, heads, d, d, 3 * heads * d, heads * d, d ** -0.5), {}),
                       dict(y=torch.ones(n * S, heads * d, dtype=dt, device=P.dev)))
        res.append((f"attn_small_fwd n={n} S={S} heads={heads} d={d}", relerr(o1["y"], o2["y"]), tol_for(dt)))
    # Downsample2D(padding=0): stride 2 over F.pad(x, (0, 1, 0, 1))
    for (n, h, w, cin, cout) in [(2, 8, 12, 64, 64), (1, 6, 6, 128, 192)]:
        ho, wo = h // 2, w // 2
        x = rnd((n * h * w, cin), dt, P.dev, g)
        wt = rnd((cout, 9 * cin), dt, P.dev, g, (9 * cin) ** -0.5)
        ga = K.Gather(K.GATHER_CONV3X3_PAD0, n_img=n, hi=h, wi=w, ho=ho, wo=wo, cin=cin, stride=2, lda=cin)
        M = n * ho * wo
        o1, o2 = P.run("gemm", lambda o: ((x, wt, o["C"], M, cout, 9 * cin, cin, 9 * cin, cout), dict(gather=ga, variant=4)),
                       dict(C=torch.zeros(M, cout, dtype=dt, device=P.dev)))
        res.append((f"gemm pad0-gather {n}x{h}x{w} {cin}->{cout}", relerr(o1["C"], o2["C"]), tol_for(dt)))
        # and the emulated gather against torch's own padded convolution
        import torch.nn.functional as F
        xi = x.float().view(n, h, w, cin).permute(0, 3, 1, 2)
        w4 = wt.float().view(cout, 9, cin).permute(0, 2, 1).reshape(cout, cin, 3, 3)
        want = F.conv2d(F.pad(xi, (0, 1, 0, 1)), w4, stride=2).permute(0, 2, 3, 1).reshape(M, cout)
        res.append((f"pad0-gather emulation vs F.conv2d {cin}->{cout}", relerr(o2["C"].float(), want), tol_for(dt)))
    return res


def check_large_offsets(P, dt):
    """Tensors beyond 2^31 bytes (the temporal VAE decoder at the reference's 1024 x 576 validation size puts 2.4 GB in front of its last
    up block; vae.decode no longer refuses such chunks): the non-GEMM kernels on that path against the emulation with VALUES compared --
    a 32-bit offset that wraps would leave finite but wrong numbers in the tail rows.  GPU only (2.4 GB operands), fp16."""
    res = []
    if P.dev.type != "cuda":
        return res
    n_s, rows, C = 8, 576 * 1024, 256                   # 8 frames of the 1024 x 576 decode at 256 channels: 2.42 GB per tensor
    g = torch.Generator(device=P.dev).manual_seed(3)
    x = (torch.randn(n_s * rows, C, generator=g, device=P.dev, dtype=torch.float32) * 1.5 + 0.2).to(dt)
    gamma, beta = torch.ones(C, device=P.dev), torch.zeros(C, device=P.dev)
    st = torch.zeros(K.GN_REPLICAS, n_s, 32, K.GN_STAT_FLOATS, device=P.dev)
    P.impl.gn_stats(x, st, n_s, rows, C, 32, prezeroed=1)
    ref = torch.zeros_like(st)
    for i in range(n_s):                                # the emulation sample by sample (float copies of 2.4 GB at once are not needed)
        r1 = torch.zeros(K.GN_REPLICAS, 1, 32, K.GN_STAT_FLOATS, device=P.dev)
        P.ref.gn_stats(x[i * rows:(i + 1) * rows], r1, 1, rows, C, 32, prezeroed=1)
        emul.gn_view(ref, n_s, 32)[:, i] = emul.gn_view(r1, 1, 32)[:, 0]
    cnt = rows * (C // 32)
    res.append((f"gn_stats {n_s}x{rows}x{C} (2.4 GB)", relerr(emul.gn_decode(st, n_s, 32, cnt, 0).view(-1, 2), emul.gn_decode(ref, n_s, 32, cnt, 0).view(-1, 2)), 1e-4))
    y = torch.empty_like(x)
    P.impl.gn_apply(x, ref, gamma, beta, y, n_s, rows, C, 32, 1e-5, 1)
    worst = 0.0
    for i in (0, n_s - 1):                              # first and last sample: the last one lies wholly beyond 2^31 bytes
        yr = torch.empty(rows, C, dtype=dt, device=P.dev)
        r1 = torch.zeros(K.GN_REPLICAS, 1, 32, K.GN_STAT_FLOATS, device=P.dev)
        emul.gn_view(r1, 1, 32)[:, 0] = emul.gn_view(ref, n_s, 32)[:, i]
        P.ref.gn_apply(x[i * rows:(i + 1) * rows], r1, gamma, beta, yr, 1, rows, C, 32, 1e-5, 1)
        worst = max(worst, relerr(y[i * rows:(i + 1) * rows], yr))
    res.append((f"gn_apply {n_s}x{rows}x{C} (2.4 GB), first and last sample", worst, tol_for(dt)))
    n = n_s * rows * C
    b = x.flip(0)
    out = torch.empty_like(x)
    P.impl.add(x, b, out, n)
    tail = slice((n_s - 1) * rows, n_s * rows)
    res.append(("add over 2.4 GB operands, last sample", relerr(out[tail], (x[tail].float() + b[tail].float()).to(dt)), tol_for(dt)))
    return res


def check_elementwise(P, dt):
    g = torch.Generator().manual_seed(8)
    res = []
    M, F = 77, 256
    pre, dout = rnd((M, 2 * F), dt, P.dev, g, 1.5), rnd((M, F), dt, P.dev, g)
    o1, o2 = P.run("geglu_fwd", lambda o: ((pre, o["y"], M, F), {}), dict(y=torch.zeros(M, F, dtype=dt, device=P.dev)))
    res.append(("geglu_fwd", relerr(o1["y"], o2["y"]), tol_for(dt)))
    o1, o2 = P.run("geglu_bwd", lambda o: ((dout, pre, o["d"], M, F), {}), dict(d=torch.zeros(M, 2 * F, dtype=dt, device=P.dev)))
    res.append(("geglu_bwd", relerr(o1["d"], o2["d"]), tol_for(dt)))
    n = 8 * 1237
    a, b = rnd((n,), dt, P.dev, g), rnd((n,), dt, P.dev, g)
    mix = torch.tensor([0.37], device=P.dev)
    o1, o2 = P.run("add", lambda o: ((a, b, o["y"], n), {}), dict(y=torch.zeros(n, dtype=dt, device=P.dev)))
    res.append(("add", relerr(o1["y"], o2["y"]), tol_for(dt)))
    o1, o2 = P.run("blend", lambda o: ((a, b, mix, o["y"], n), {}), dict(y=torch.zeros(n, dtype=dt, device=P.dev)))
    res.append(("blend", relerr(o1["y"], o2["y"]), tol_for(dt)))
    o1, o2 = P.run("blend_bwd", lambda o: ((a, mix, o["da"], o["db"], n), {}),
                   dict(da=torch.zeros(n, dtype=dt, device=P.dev), db=torch.zeros(n, dtype=dt, device=P.dev)))
    res.append(("blend_bwd da", relerr(o1["da"], o2["da"]), tol_for(dt)))
    res.append(("blend_bwd db", relerr(o1["db"], o2["db"]), tol_for(dt)))
    rows, C = 150, 192
    x = rnd((rows, C), dt, P.dev, g)
    vec = rndf((5, 2 * C), P.dev, g)
    for rpg, mod in ((30, 0), (0, 5)):
        o1, o2 = P.run("add_rowvec", lambda o: ((x, vec[:, C:], o["y"], rows, C, 2 * C, rpg, mod), {}), dict(y=torch.zeros_like(x)))
        res.append((f"add_rowvec rpg={rpg} mod={mod}", relerr(o1["y"], o2["y"]), tol_for(dt)))
        for acc in (0, 1):
            o1, o2 = P.run("colsum", lambda o: ((x, o["s"], rows, C, C, 5, rpg, mod), dict(accumulate=acc)),
                           dict(s=torch.ones(5, C, device=P.dev)))
            res.append((f"colsum rpg={rpg} mod={mod} acc={acc}", relerr(o1["s"], o2["s"]), 1e-3))
            scr = torch.full((K.colsum_slabs(rows, rpg, mod) * 5 * C,), float("nan"), device=P.dev)      # deterministic (slab) form
            o1, o2 = P.run("colsum", lambda o: ((x, o["s"], rows, C, C, 5, rpg, mod), dict(accumulate=acc, scratch=scr)),
                           dict(s=torch.ones(5, C, device=P.dev)))
            res.append((f"colsum rpg={rpg} mod={mod} acc={acc} slabs", relerr(o1["s"], o2["s"]), 1e-3))
    tall = rnd((3000, 320), dt, P.dev, g)                   # several row slabs per group, the last one ragged, one group shorter
    for rpg, mod, ng in ((1400, 0, 3), (0, 2, 2)):
        scr = torch.full((K.colsum_slabs(3000, rpg, mod) * ng * 320,), float("nan"), device=P.dev)
        o1, o2 = P.run("colsum", lambda o: ((tall, o["s"], 3000, 320, 320, ng, rpg, mod), dict(scratch=scr)), dict(s=torch.ones(ng, 320, device=P.dev)))
        res.append((f"colsum 3000 rows rpg={rpg} mod={mod} slabs", relerr(o1["s"], o2["s"]), 1e-3))
    big = rnd((700, 3 * 320), dt, P.dev, g)
    o1, o2 = P.run("colsum", lambda o: ((big[:, 320:], o["s"], 700, 320, 960, 1, 700, 0), {}), dict(s=torch.zeros(1, 320, device=P.dev)))
    res.append(("colsum strided 1 group", relerr(o1["s"], o2["s"]), 1e-3))
    for (r, c) in [(100, 64), (333, 200), (64, 1000)]:
        xx = rnd((r, c + 8), dt, P.dev, g)
        ldo = (r + 63) // 64 * 64
        o1, o2 = P.run("transpose", lambda o: ((xx, c + 8, o["t"], ldo, r, c), {}), dict(t=torch.full((c, ldo), 3.0, dtype=dt, device=P.dev)))
        res.append((f"transpose {r}x{c}", relerr(o1["t"], o2["t"]), 0.0))
        wf = rndf((r, c), P.dev, g)
        o1, o2 = P.run("cast_transpose_from_f32", lambda o: ((wf, o["t"], r, c), {}), dict(t=torch.zeros(c, r, dtype=dt, device=P.dev)))
        res.append((f"cast_transpose {r}x{c}", relerr(o1["t"], o2["t"]), 0.0))
    a2, b2 = rnd((90, 64), dt, P.dev, g), rnd((90, 128), dt, P.dev, g)
    o1, o2 = P.run("concat2", lambda o: ((a2, 64, b2, 128, o["c"], 90), {}), dict(c=torch.zeros(90, 192, dtype=dt, device=P.dev)))
    res.append(("concat2", relerr(o1["c"], o2["c"]), 0.0))
    cat = o2["c"]
    o1, o2 = P.run("split2", lambda o: ((cat, o["a"], 64, o["b"], 128, 90), {}),
                   dict(a=torch.zeros(90, 64, dtype=dt, device=P.dev), b=torch.zeros(90, 128, dtype=dt, device=P.dev)))
    res.append(("split2", max(relerr(o1["a"], o2["a"]), relerr(o1["b"], o2["b"])), 0.0))
    xin = rnd((2 * 6 * 10, 64), dt, P.dev, g)
    o1, o2 = P.run("sum2x2", lambda o: ((xin, o["y"], 2, 3, 5, 64), {}), dict(y=torch.zeros(2 * 3 * 5, 64, dtype=dt, device=P.dev)))
    res.append(("sum2x2", relerr(o1["y"], o2["y"]), tol_for(dt)))
    wf = rndf((1000 * 8 + 3,), P.dev, g)
    o1, o2 = P.run("cast_from_f32", lambda o: ((wf, o["y"], wf.numel()), {}), dict(y=torch.zeros(wf.numel(), dtype=dt, device=P.dev)))
    res.append(("cast_from_f32", relerr(o1["y"], o2["y"]), 0.0))
    img = rndf((3, 8, 6, 10), P.dev, g)
    o1, o2 = P.run("nchw_to_rows", lambda o: ((img, o["y"], 3, 8, 6, 10, 64), dict(mul=2.0)), dict(y=torch.ones(180, 64, dtype=dt, device=P.dev)))
    res.append(("nchw_to_rows", relerr(o1["y"], o2["y"]), tol_for(dt)))
    rows_t = rnd((180, 4), dt, P.dev, g)
    o1, o2 = P.run("rows_to_nchw", lambda o: ((rows_t, o["y"], 3, 4, 6, 10, 4), {}), dict(y=torch.zeros(3, 4, 6, 10, device=P.dev)))
    res.append(("rows_to_nchw", relerr(o1["y"], o2["y"]), 0.0))
    return res


def check_optim(P, dt):
    g = torch.Generator().manual_seed(9)
    res = []
    B, T, C, HW = 2, 3, 4, 48
    pred = rnd((B * T * HW, C), dt, P.dev, g)
    noisy, target = rndf((B, T, C, HW), P.dev, g), rndf((B, T, C, HW), P.dev, g)
    sigma = torch.tensor([0.7, 3.1], device=P.dev)
    st = torch.tensor([0, 1024.0, 0, 0, 1, 1, 1, 0] + [1.0] + [0.0] * 7, dtype=torch.float32, device=P.dev)
    o1, o2 = P.run("edm_loss", lambda o: ((pred, C, noisy, target, sigma, o["loss"], o["d"], B, T, C, HW, st), {}),
                   dict(loss=torch.zeros(1, device=P.dev), d=torch.zeros(B * T * HW, 64, dtype=dt, device=P.dev)))
    res.append(("edm_loss loss", relerr(o1["loss"], o2["loss"]), 1e-4))
    res.append(("edm_loss dpred", relerr(o1["d"], o2["d"]), tol_for(dt)))
    n = 4 * 5000
    p, gr = rndf((n,), P.dev, g), rndf((n,), P.dev, g, 100.0)
    m, v = rndf((n,), P.dev, g, 0.1), rndf((n,), P.dev, g).abs()
    # (found_inf, schedule slots 9..15: kind, warmup, total, cycles, power, lr_end ratio, scheduler steps per step)
    cases = [(False, [0, 0, 0, 0, 0, 0, 0]), (True, [0, 0, 0, 0, 0, 0, 0]), (False, [1, 10, 0, 0, 0, 0, 1]), (False, [2, 2, 40, 0, 0, 0, 2]),
             (False, [3, 1, 20, 0.5, 0, 0, 1]), (False, [4, 1, 20, 3, 0, 0, 2]), (False, [5, 2, 30, 0, 2.0, 1e-2, 1]),
             (False, [5, 1, 2, 0, 1.0, 1e-2, 1]),
             (False, [6, 2, 0, 0, 0, 0, 1]), (False, [6, 2, 0, 0, 0, 0, 4])]     # piecewise_constant: 2 rules behind the state (boundaries 2 and 11)
    for found, sched in cases:
        gg = gr.clone()
        if found:
            gg[1234] = float("inf")
        st0 = torch.tensor([3, 1024.0, 5, 0, 1, 1, 1, 0, 1.0] + sched + [2.0, 0.7, 11.0, 0.3, 0.05] + [0.0] * 19, dtype=torch.float32, device=P.dev)
        outs = dict(st=st0, p=p.clone(), m=m.clone(), v=v.clone(), pa=torch.zeros(n, dtype=dt, device=P.dev))

        def seq(be, o):
            be.check_finite(gg, n, o["st"])
            be.optim_prep(o["st"], 0.9, 0.999, 2.0, 0.5, 7, 1)
            be.adamw(o["p"], gg, o["m"], o["v"], n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 0.5, o["st"], o["pa"])
        o1 = {k_: t.clone() for k_, t in outs.items()}
        o2 = {k_: t.clone() for k_, t in outs.items()}
        seq(P.impl, o1)
        seq(P.ref, o2)
        if P.dev.type == "cuda":
            torch.cuda.synchronize()
        res.append((f"optim found_inf={found} sched={sched[0]} state", relerr(o1["st"], o2["st"]), 1e-5))
        res.append((f"optim sched={sched[0]} lr multiplier", abs(float(o1["st"][8]) - float(o2["st"][8])), 5e-6))
        for nm in ("p", "m", "v"):
            res.append((f"adamw found_inf={found} {nm}", relerr(o1[nm], o2[nm]), 1e-5))
        res.append((f"adamw found_inf={found} p_act", relerr(o1["pa"], o2["pa"]), tol_for(dt)))
    # param_mode 1 (SVDX_PARAMS_BF16_REFERENCE): torch.optim.AdamW's op sequence on bf16 tensors, three steps in a row on bf16-valued state.
    # Every stored value is a bf16 number; kernel and emulation may differ by one bf16 step on the rare element where a float scalar of the
    # device (bias corrections by powf) and torch's double differ in the last place -- count them
    nb = 4 * 5000
    pb = rndf((nb,), P.dev, g).to(torch.bfloat16).float()
    for tiled in (False, True):
        o = [dict(p=pb.clone(), m=torch.zeros(nb, device=P.dev), v=torch.zeros(nb, device=P.dev), pa=torch.zeros(nb, dtype=torch.bfloat16, device=P.dev),
                  pt=torch.zeros(nb, dtype=torch.bfloat16, device=P.dev), st=torch.tensor([0, 1.0, 0, 0, 1, 1, 1, 0, 1.0] + [0.0] * 27, device=P.dev)) for _ in range(2)]
        tiles_b = torch.tensor([[i * 2000, 52, 36, 48, -1, 0] for i in range(10)], dtype=torch.int32, device=P.dev)      # 10 tiles of 36 x 48 (pitch 52)
        for step in range(3):
            gb = rndf((nb,), P.dev, g) * 0.1
            for be, oo in ((P.impl, o[0]), (P.ref, o[1])):
                be.optim_prep(oo["st"], 0.9, 0.999, 2.0, 0.5, 2000, 0)
                if tiled:
                    be.adamw_tiled(oo["p"], gb, oo["m"], oo["v"], tiles_b, 10, 1e-2, 0.9, 0.999, 1e-8, 1e-2, 1.0, oo["st"], oo["pa"], oo["pt"], param_mode=1)
                else:
                    be.adamw(oo["p"], gb, oo["m"], oo["v"], nb, 1e-2, 0.9, 0.999, 1e-8, 1e-2, 1.0, oo["st"], oo["pa"], param_mode=1)
        if P.dev.type == "cuda":
            torch.cuda.synchronize()
        tag = "adamw_tiled" if tiled else "adamw"
        for nm in ("p", "m", "v"):
            a, b = o[0][nm], o[1][nm]
            res.append((f"{tag} bf16-reference {nm} stays bf16-valued", float((a - a.to(torch.bfloat16).float()).abs().max()), 0.0))
            res.append((f"{tag} bf16-reference {nm} elements off the emulation", float((a != b).float().mean()), 2e-3))
            res.append((f"{tag} bf16-reference {nm}", relerr(a, b), 1e-2))      # (a differing element is off by ONE bf16 step, 0.4-0.8 %)
    for n_e in (4 * 3000, 4 * 3000 + 3):
        sh, w = rndf((n_e,), P.dev, g), rndf((n_e,), P.dev, g)
        o1, o2 = P.run("ema_lerp", lambda o: ((o["s"], w, n_e, 0.013), {}), dict(s=sh))
        res.append((f"ema_lerp n={n_e}", relerr(o1["s"], o2["s"]), 1e-6))
    # svdx_allreduce_grads: the ranks of a node as separate buffers of ONE device (the peer mapping is the caller's business): after
    # reduce-scatter + all-gather every buffer holds the sum, added in rank order -- bit-equal to that sum and across ranks
    for world, n_e in ((2, 4 * 1000), (3, 4 * 1001), (4, 4 * 6), (8, 4 * 5003), (8, 8), (5, 4 * 777)):
        bufs = [rndf((n_e,), P.dev, g) for _ in range(world)]
        want = bufs[0].clone()
        for q in range(1, world):
            want = want + bufs[q]
        is_lib = hasattr(P.impl, "lib")
        for be, mine in ((P.impl, [b.clone() for b in bufs]), (P.ref, [b.clone() for b in bufs])):
            handles = [b.data_ptr() for b in mine] if (be is P.impl and is_lib) else mine
            for phase in (-1, 0, 1):
                for r in range(world):
                    be.allreduce_grads(handles, r, n_e, phase)
            if P.dev.type == "cuda":
                torch.cuda.synchronize()
            tag = "impl" if be is P.impl else "emul"
            res.append((f"allreduce_grads {tag} world={world} n={n_e} equals the rank-ordered sum", float(max((b - want).abs().max() for b in mine)), 0.0))
    # span zeroing and the float-store finalize (write-once weight gradients)
    buf = rndf((5000,), P.dev, g)
    spans = torch.tensor([[0, 64], [128, 4], [1000, 2048], [4996, 4]], dtype=torch.int32, device=P.dev)
    o1, o2 = P.run("zero_spans", lambda o: ((o["b"], spans, 4), {}), dict(b=buf))
    res.append(("zero_spans", relerr(o1["b"], o2["b"]), 0.0))
    slabs = rndf((3, 40, 64), P.dev, g)
    for mode in (1, 2):
        o1, o2 = P.run("gemm_finalize", lambda o: ((slabs, 3, 40 * 64, o["c"], 40, 64, 64), dict(accumulate_f32=mode, dtype=dt)),
                       dict(c=torch.full((40, 64), 3.0, device=P.dev)))
        res.append((f"gemm_finalize float mode {mode}", relerr(o1["c"], o2["c"]), 1e-6))
    # tiled AdamW with transposed twins: two matrices fused into one [K, N1+N2] twin, a plain matrix, a bias
    from svd_xtend_amd.train import build_adam_tiles
    shapes = [(128, 192), (68, 192), (100, 64), (320,)]
    ps = [torch.nn.Parameter(rndf(sh, P.dev, g)) for sh in shapes]
    offs, off = [], 0
    for q in ps:
        offs.append(off)
        off = (off + q.numel() + 63) // 64 * 64
    n = off
    wt_map = {id(ps[0]): (64, 196), id(ps[1]): (64 + 128, 196)}       # twin [192, 196] at offset 64
    tiles = build_adam_tiles(ps, offs, wt_map, P.dev)
    p0 = torch.zeros(n, device=P.dev)
    for q, o in zip(ps, offs):
        p0[o:o + q.numel()] = q.data.reshape(-1)
    gr, m, v = rndf((n,), P.dev, g, 10.0), rndf((n,), P.dev, g, 0.1), rndf((n,), P.dev, g).abs()
    st = torch.tensor([3, 64.0, 5, 0, 1 / 64.0, 0.271, 0.003, 0, 0.37] + [0.0] * 7, dtype=torch.float32, device=P.dev)
    outs = dict(p=p0, m=m, v=v, pa=torch.zeros(n, dtype=dt, device=P.dev), pt=torch.zeros(64 + 192 * 196, dtype=dt, device=P.dev))
    o1, o2 = P.run("adamw_tiled", lambda o: ((o["p"], gr, o["m"], o["v"], tiles, tiles.shape[0], 1e-3, 0.9, 0.999, 1e-8, 1e-2, 0.5, st,
                                              o["pa"], o["pt"]), {}), outs)
    for nm in ("p", "m", "v"):
        res.append((f"adamw_tiled {nm}", relerr(o1[nm], o2[nm]), 1e-5))
    res.append(("adamw_tiled p_act", relerr(o1["pa"], o2["pa"]), tol_for(dt)))
    res.append(("adamw_tiled transposed twin", relerr(o1["pt"], o2["pt"]), tol_for(dt)))
    wt = o1["pt"][64:].view(192, 196)
    res.append(("adamw_tiled twin == W^T", relerr(wt[:, :128].float(), o1["pa"][offs[0]:offs[0] + 128 * 192].view(128, 192).t().float()), 0.0))
    return res


RING_VARIANTS = (16, 18, 20, 23, 25) # between them every ring-staged instantiation of gemm_v4_kernel (N % 160 picks the 160- or 128-wide one)


def run_all(impl, dev, dtypes=DTYPES, verbose=True):
    P = Pair(impl, dev)
    out = []
    for dt in dtypes:
        checks = [("gemm_plain_v1", lambda: check_gemm_plain(P, dt, 1))]
        for v in (4, 6) + RING_VARIANTS:
            checks += [(f"gemm_plain_v{v}", lambda v=v: check_gemm_plain(P, dt, v)), (f"gemm_gather_v{v}", lambda v=v: check_gemm_gather(P, dt, v))]
        checks += [("gemm_tn", lambda: check_gemm_tn(P, dt)), ("gemm_tn_s3", lambda: check_gemm_tn(P, dt, 3)), ("gemm_tn_s4", lambda: check_gemm_tn(P, dt, 4)),
                   ("gemm_tn_v18", lambda: check_gemm_tn(P, dt, 18)),
                   ("gemm_tn_flat", lambda: check_gemm_tn(P, dt, K.TN_FLAT)), ("gemm_tn_v18_flat", lambda: check_gemm_tn(P, dt, 18 | K.TN_FLAT)),
                   ("gemm_gn_v4", lambda: check_gemm_gn(P, dt, 4)), ("gemm_gn_v6", lambda: check_gemm_gn(P, dt, 6)), ("gemm_gn_v23", lambda: check_gemm_gn(P, dt, 23)),
                   ("gemm_gn_v18", lambda: check_gemm_gn(P, dt, 18)), ("gemm_gn_v24", lambda: check_gemm_gn(P, dt, 24)), ("gemm_gn_v26", lambda: check_gemm_gn(P, dt, 26)),
                   ("gemm_geglu", lambda: check_gemm_geglu(P, dt))]
        checks += [(f"gemm_geglu_v{v}", lambda v=v: check_gemm_geglu(P, dt, v)) for v in (17, 18, 21, 26)]
        checks += [
                  ("small", lambda: check_small(P, dt)), ("groupnorm", lambda: check_groupnorm(P, dt)),
                  ("layernorm", lambda: check_layernorm(P, dt)), ("attention", lambda: check_attention(P, dt)),
                  ("temporal_attention", lambda: check_temporal_attention(P, dt)), ("tsa", lambda: check_tsa(P, dt)),
                  ("encoders", lambda: check_encoders(P, dt)), ("elementwise", lambda: check_elementwise(P, dt)),
                  ("optim", lambda: check_optim(P, dt))]
        for name, fn in checks:
            try:
                for label, err, tol in fn():
                    ok = err <= tol and math.isfinite(err)
                    out.append(dict(group=name, dtype=str(dt), label=label, err=err, tol=tol, ok=ok))
                    if verbose and not ok:
                        print(f"FAIL [{dt}] {label}: err={err:.3e} tol={tol:.1e}", flush=True)
            except Exception as e:  # noqa: BLE001 - report and continue with the other kernels
                out.append(dict(group=name, dtype=str(dt), label=f"{name} raised", err=float("inf"), tol=0.0, ok=False,
                                exc=repr(e)[:500]))
                if verbose:
                    print(f"EXC  [{dt}] {name}: {e!r}", flush=True)
    return out
