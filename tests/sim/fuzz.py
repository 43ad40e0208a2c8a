"""Randomised shapes through the simulated kernels against the fp32 emulation (tests/emul.py): the fixed shapes of tests/kernel_checks.py
are the UNet's; a drop-in library has to be right on every shape its argument checks admit.  TEST INFRASTRUCTURE.

    python tests/sim/fuzz.py [family ...] [--n 60] [--seed 0] [--bf16]

Families: gemm, gemm_gn, gradfin, gather, tn, geglu, norm, lnbwd, attn, tattn, tsa, small, batch, rows, optim.  Prints every failing case with the arguments that reproduce it."""
import math
import os
import random
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(os.path.dirname(HERE)), os.path.dirname(HERE), HERE]
import emul  # noqa: E402
import kernel_checks as kc  # noqa: E402
from svd_xtend_amd import kernels as K  # noqa: E402

NT_VARIANTS = (1, 4, 6, 7, 8, 16, 17, 18, 20, 21, 22, 23, 24, 25, 26, 27, 28, 36)
HEAD = 64


def pick_dim(rng, lo, hi, mult=1):
    """log-uniform size in [lo, hi], rounded to a multiple of `mult`, with a taste for tile boundaries +- 1."""
    v = int(math.exp(rng.uniform(math.log(lo), math.log(hi))))
    if rng.random() < 0.3:
        v = rng.choice((16, 32, 64, 96, 128, 160, 192, 256, 320)) * rng.randint(1, 4) + rng.choice((-1, 0, 0, 1))
    return max(mult, (max(lo, min(hi, v)) + mult - 1) // mult * mult)


def fuzz_gemm(P, dt, rng, g):
    v = rng.choice(NT_VARIANTS)
    M, N, Kd = pick_dim(rng, 1, 700), pick_dim(rng, 1, 700), 64 * rng.randint(1, 12)
    if rng.random() < 0.5:
        N = (N + 7) // 8 * 8
    pad_a, pad_b, pad_c = (8 * rng.randint(0, 3) for _ in range(3))
    lda, ldb = Kd + pad_a, Kd + pad_b
    mode = rng.choice(("plain", "bias_res", "rowvec", "rowvec_mod", "f32", "slab", "add"))
    ldc = N + (pad_c if N % 8 == 0 and mode != "slab" else 0)
    Aw, Bw = kc.rnd((M, lda), dt, P.dev, g), kc.rnd((N, ldb), dt, P.dev, g, Kd ** -0.5)
    A, B = Aw[:, :Kd], Bw[:, :Kd]
    bias, R, rv = kc.rndf((N,), P.dev, g), kc.rnd((M, ldc), dt, P.dev, g), kc.rndf((5, N), P.dev, g)
    kw = dict(variant=v)
    out = torch.zeros(M, ldc, dtype=dt, device=P.dev)
    sk = 1
    if mode == "bias_res":
        kw.update(bias=bias, res=R, ldres=ldc)
    elif mode == "rowvec":
        kw.update(bias=bias, rowvec=rv, rv_ld=N, rv_rpg=(M + 4) // 5)
    elif mode == "rowvec_mod":
        kw.update(rowvec=rv, rv_ld=N, rv_mod=5)
    elif mode == "f32":
        kw.update(out_mode=K.OUT_F32, alpha=0.5)
        out = torch.zeros(M, ldc, dtype=torch.float32, device=P.dev)
    elif mode == "add":
        kw.update(out_mode=K.OUT_F32_ADD)
        out = torch.ones(M, ldc, dtype=torch.float32, device=P.dev)
    elif mode == "slab":
        sk = rng.choice([x for x in (1, 2, 3, 4, 5, 8) if x <= Kd // 64])
        kw.update(out_mode=K.OUT_F32_SLAB, split_k=sk)
        out = torch.full((sk, M, ldc), float("nan"), dtype=torch.float32, device=P.dev)        # every slab element has exactly one writer
    desc = f"gemm v{v} M={M} N={N} K={Kd} lda={lda} ldb={ldb} ldc={ldc} {mode} sk={sk}"
    o1, o2 = P.run("gemm", lambda o: ((A, B, o["C"], M, N, Kd, lda, ldb, ldc), kw), dict(C=out))
    a, b = o1["C"], o2["C"]          # slab mode: slice by slice (the emulation cuts K like the kernel; launch_gemm_v4 may give every slice XCDs of its own)
    return desc, kc.relerr(a[..., :N], b[..., :N]), kc.tol_for(dt)


def fuzz_gather(P, dt, rng, g):
    v = rng.choice(tuple(x for x in NT_VARIANTS if x != 18))
    ci, co = 64 * rng.randint(1, 4), rng.choice((32, 64, 96, 128, 160, 200, 256, 320, 8, 4))
    kind = rng.choice(("s1", "s2", "ups", "dgrad2", "t3", "pad0"))
    n = rng.randint(1, 3)
    h, w = rng.randint(1, 14), rng.randint(1, 18)
    taps = 9
    if kind == "s1":
        ga, M, nsrc = K.Gather(K.GATHER_CONV3X3, n_img=n, hi=h, wi=w, ho=h, wo=w, cin=ci, stride=1, lda=ci), n * h * w, n * h * w
    elif kind == "s2":
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        ga, M, nsrc = K.Gather(K.GATHER_CONV3X3, n_img=n, hi=h, wi=w, ho=ho, wo=wo, cin=ci, stride=2, lda=ci), n * ho * wo, n * h * w
    elif kind == "pad0":
        h, w = 2 * rng.randint(1, 7), 2 * rng.randint(1, 9)
        ga, M, nsrc = K.Gather(K.GATHER_CONV3X3_PAD0, n_img=n, hi=h, wi=w, ho=h // 2, wo=w // 2, cin=ci, stride=2, lda=ci), n * (h // 2) * (w // 2), n * h * w
    elif kind == "ups":
        ga, M, nsrc = K.Gather(K.GATHER_CONV3X3, n_img=n, hi=2 * h, wi=2 * w, ho=2 * h, wo=2 * w, cin=ci, stride=1, ups=1, lda=ci), n * 4 * h * w, n * h * w
    elif kind == "dgrad2":
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        ga, M, nsrc = K.Gather(K.GATHER_CONV3X3_DGRAD2, n_img=n, hi=ho, wi=wo, ho=h, wo=w, cin=ci, lda=ci), n * h * w, n * ho * wo
    else:
        T, hw, taps = rng.randint(1, 9), rng.randint(1, 60), 3
        ga, M, nsrc = K.Gather(K.GATHER_TEMPORAL3, n_img=n, cin=ci, t=T, hw=hw, lda=ci), n * T * hw, n * T * hw
    A = kc.rnd((nsrc, ci), dt, P.dev, g)
    B = kc.rnd((co, taps * ci), dt, P.dev, g, (taps * ci) ** -0.5)
    bias = kc.rndf((co,), P.dev, g)
    kw = dict(bias=bias, gather=ga, variant=v)
    sk = rng.choice((1, 1, 2, 3))
    sk = min(sk, taps * ci // 64)
    if sk > 1:
        kw = dict(gather=ga, variant=v, out_mode=K.OUT_F32_SLAB, split_k=sk)
        out = torch.zeros(sk, M, co, device=P.dev)
    else:
        out = torch.zeros(M, co, dtype=dt, device=P.dev)
    desc = f"gather v{v} {kind} n={n} h={h} w={w} ci={ci} co={co} M={M} sk={sk} {ga}"
    o1, o2 = P.run("gemm", lambda o: ((A, B, o["C"], M, co, taps * ci, ci, taps * ci, co), kw), dict(C=out))
    a, b = (o1["C"].sum(0), o2["C"].sum(0)) if sk > 1 else (o1["C"], o2["C"])
    return desc, kc.relerr(a, b), kc.tol_for(dt)


def fuzz_tn(P, dt, rng, g):
    stages = rng.choice((0, 3, 4, 18, 0 | K.TN_FLAT, 18 | K.TN_FLAT))     # + the L2-prefetch forms and rounds 1-4's flat staging
    R, N, Kd = pick_dim(rng, 1, 2400), pick_dim(rng, 8, 600, 8), pick_dim(rng, 8, 600, 8)
    A, B = kc.rnd((R, N), dt, P.dev, g), kc.rnd((R, Kd), dt, P.dev, g, R ** -0.5)
    mode = rng.choice((K.OUT_F32, K.OUT_F32_ADD, K.OUT_F32_SLAB))
    sk = 1
    if mode == K.OUT_F32_SLAB:
        kt = (R + 63) // 64
        sk = rng.choice([x for x in (1, 2, 3, 4, 8, 16, 24) if x <= kt] or [1])       # incl. the slice counts ops._tn_slices asks for (XCD maps of csrc/gemm.hip tn_who)
        while sk > 1 and (kt + sk - 1) // sk * (sk - 1) >= kt:
            sk -= 1
        outs = dict(C=torch.full((sk, N, Kd), float("nan"), device=P.dev), cs=torch.full((sk, N), 7.0, device=P.dev))      # every slab element has exactly one writer
    else:
        outs = dict(C=torch.ones(N, Kd, device=P.dev), cs=torch.ones(N, device=P.dev))
    with_cs = rng.random() < 0.6
    desc = f"tn stages={stages} R={R} N={N} K={Kd} mode={mode} sk={sk} colsum={with_cs}"
    o1, o2 = P.run("gemm_tn", lambda o: ((A, B, o["C"], R, N, Kd, N, Kd, Kd), dict(out_mode=mode, split_k=sk, a_colsum=o["cs"] if with_cs else None, stages=stages)), outs)
    e = kc.relerr(o1["C"], o2["C"])                  # slabs slice by slice
    if with_cs:
        e = max(e, kc.relerr(o1["cs"], o2["cs"]))
    return desc, e, kc.tol_for(dt)


def fuzz_geglu(P, dt, rng, g):
    v = rng.choice((4, 6, 8, 16, 17, 18, 20, 21, 22, 24, 26, 27))
    M, C, F = pick_dim(rng, 1, 600), 64 * rng.randint(1, 6), rng.choice((64 * rng.randint(1, 10), 160 * rng.randint(1, 4) * 2))
    x, W1, b1 = kc.rnd((M, C), dt, P.dev, g), kc.rnd((2 * F, C), dt, P.dev, g, C ** -0.5), kc.rndf((2 * F,), P.dev, g)
    desc = f"geglu v{v} M={M} C={C} F={F}"
    o1, o2 = P.run("gemm", lambda o: ((x, W1, o["pre"], M, 2 * F, C, C, C, 2 * F), dict(bias=b1, variant=v, epilogue=K.EPI_GEGLU_FWD, aux_out=o["h"], aux_dim=F)),
                   dict(pre=torch.zeros(M, 2 * F, dtype=dt, device=P.dev), h=torch.zeros(M, F, dtype=dt, device=P.dev)))
    e = max(kc.relerr(o1["pre"], o2["pre"]), 0.5 * kc.relerr(o1["h"], o2["h"]))
    if F % 128 == 0 or F % 160 == 0:
        pre = o2["pre"]
        dy, W2t = kc.rnd((M, C), dt, P.dev, g), kc.rnd((F, C), dt, P.dev, g, C ** -0.5)
        o1, o2 = P.run("gemm", lambda o: ((dy, W2t, o["dpre"], M, F, C, C, C, 2 * F), dict(variant=v, epilogue=K.EPI_GEGLU_BWD, aux_in=pre, aux_dim=F)),
                       dict(dpre=torch.zeros(M, 2 * F, dtype=dt, device=P.dev)))
        e = max(e, 0.5 * kc.relerr(o1["dpre"], o2["dpre"]))
    return desc, e, kc.tol_for(dt)


def fuzz_norm(P, dt, rng, g):
    if rng.random() < 0.5:
        rows, C = pick_dim(rng, 1, 3000), 64 * rng.randint(1, 20)
        x = (kc.rnd((rows, C), dt, P.dev, g) * 2 + 0.5).to(dt)
        dy, add = kc.rnd((rows, C), dt, P.dev, g), kc.rnd((rows, C), dt, P.dev, g)
        gamma, beta = 1 + 0.1 * kc.rndf((C,), P.dev, g), 0.1 * kc.rndf((C,), P.dev, g)
        desc = f"layernorm rows={rows} C={C}"
        o1, o2 = P.run("ln_fwd", lambda o: ((x, gamma, beta, o["y"], o["st"], rows, C, 1e-5), {}), dict(y=torch.zeros_like(x), st=torch.zeros(rows, 2, device=P.dev)))
        e = max(kc.relerr(o1["y"], o2["y"]), kc.relerr(o1["st"], o2["st"]) * 20)
        return desc, e, kc.tol_for(dt)
    n_s, rows, C = rng.randint(1, 4), pick_dim(rng, 1, 500), 32 * rng.choice((2, 4, 6, 10, 20, 30, 40, 60, 80))
    x = (kc.rnd((n_s * rows, C), dt, P.dev, g) * 1.5 + 0.3).to(dt)
    dy = kc.rnd((n_s * rows, C), dt, P.dev, g)
    gamma, beta = 1 + 0.1 * kc.rndf((C,), P.dev, g), 0.1 * kc.rndf((C,), P.dev, g)
    cnt = rows * (C // 32)
    silu = rng.randint(0, 1)
    desc = f"groupnorm n_s={n_s} rows={rows} C={C} silu={silu}"
    st = torch.zeros(K.GN_REPLICAS, n_s, 32, K.GN_STAT_FLOATS, device=P.dev)
    o1, o2 = P.run("gn_stats", lambda o: ((x, o["st"], n_s, rows, C, 32), {}), dict(st=st))
    e = 20 * kc.relerr(emul.gn_decode(o1["st"], n_s, 32, cnt, 0).view(-1, 2), emul.gn_decode(o2["st"], n_s, 32, cnt, 0).view(-1, 2))
    stats = o2["st"]
    o1, o2 = P.run("gn_apply", lambda o: ((x, stats, gamma, beta, o["y"], n_s, rows, C, 32, 1e-5, silu), {}), dict(y=torch.zeros_like(x)))
    e = max(e, kc.relerr(o1["y"], o2["y"]))
    o1, o2 = P.run("gn_bwd_stats", lambda o: ((dy, x, stats, gamma, beta, o["b"], n_s, rows, C, 32, 1e-5, silu), {}),
                   dict(b=torch.zeros(K.GN_REPLICAS, n_s, 32, K.GN_STAT_FLOATS, device=P.dev)))
    e = max(e, kc.relerr(emul.gn_decode(o1["b"], n_s, 32, cnt, 1).view(-1, 2), emul.gn_decode(o2["b"], n_s, 32, cnt, 1).view(-1, 2)))
    bst = o2["b"]
    o1, o2 = P.run("gn_bwd_apply", lambda o: ((dy, x, stats, bst, gamma, beta, None, o["dx"], n_s, rows, C, 32, 1e-5, silu), {}), dict(dx=torch.zeros_like(x)))
    e = max(e, kc.relerr(o1["dx"], o2["dx"]))
    return desc, e, kc.tol_for(dt)


def fuzz_attn(P, dt, rng, g):
    nb, heads, S = rng.randint(1, 3), rng.randint(1, 3), pick_dim(rng, 1, 400)
    C = heads * 64
    scale = rng.choice((0.125, 0.1))
    qkv = kc.rnd((nb * S, 3 * C), dt, P.dev, g, 1.0)
    if rng.random() < 0.5 and S > 1:
        ramp = torch.linspace(0.3, 4.0, S).repeat(nb)[:, None]
        qkv[:, C:2 * C] = (qkv[:, C:2 * C].float() * ramp).to(dt)
    d_o = kc.rnd((nb * S, C), dt, P.dev, g)
    q, k, v = qkv, qkv[:, C:], qkv[:, 2 * C:]
    desc = f"attn nb={nb} heads={heads} S={S} scale={scale}"
    o1, o2 = P.run("attn_fwd", lambda o: ((q, k, v, o["o"], o["lse"], nb, heads, S, 3 * C, C, scale), {}),
                   dict(o=torch.zeros(nb * S, C, dtype=dt, device=P.dev), lse=torch.zeros(nb * heads * S, device=P.dev)))
    e = max(0.5 * kc.relerr(o1["o"], o2["o"]), float((o1["lse"] - o2["lse"]).abs().max()) * 0.1 * kc.tol_for(dt) / 2e-3)
    o_ref, lse = o2["o"], o2["lse"]
    o1, o2 = P.run("attn_bwd_prep", lambda o: ((o_ref, d_o, o["D"], nb, heads, S, C), {}), dict(D=torch.zeros(nb * heads * S, device=P.dev)))
    e = max(e, kc.relerr(o1["D"], o2["D"]))
    D = o2["D"]
    dqkv = torch.zeros(nb * S, 3 * C, dtype=dt, device=P.dev)
    o1, o2 = P.run("attn_bwd_dkv", lambda o: ((q, k, v, d_o, lse, D, o["d"][:, C:], o["d"][:, 2 * C:], nb, heads, S, 3 * C, C, 3 * C, scale), {}), dict(d=dqkv))
    e = max(e, 0.25 * kc.relerr(o1["d"][:, 2 * C:], o2["d"][:, 2 * C:]))
    if S == 1:                                   # dk and dq are zero up to rounding noise on both sides: no relative error to speak of
        return desc, max(e, float(o1["d"][:, C:2 * C].float().abs().max())), kc.tol_for(dt)
    e = max(e, 0.25 * kc.relerr(o1["d"][:, C:2 * C], o2["d"][:, C:2 * C]))
    o1, o2 = P.run("attn_bwd_dq", lambda o: ((q, k, v, d_o, lse, D, o["d"], nb, heads, S, 3 * C, C, 3 * C, scale), {}), dict(d=dqkv))
    e = max(e, 0.25 * kc.relerr(o1["d"][:, :C], o2["d"][:, :C]))
    return desc, e, kc.tol_for(dt)


def fuzz_tattn(P, dt, rng, g):
    B, T, HW, heads = rng.randint(1, 3), rng.randint(1, 32), rng.randint(1, 40), rng.randint(1, 5)
    C, M = heads * 64, B * T * HW
    qkv, d_o = kc.rnd((M, 3 * C), dt, P.dev, g), kc.rnd((M, C), dt, P.dev, g)
    q, k, v = qkv, qkv[:, C:], qkv[:, 2 * C:]
    desc = f"tattn B={B} T={T} HW={HW} heads={heads}"
    o1, o2 = P.run("tattn_fwd", lambda o: ((q, k, v, o["o"], B, T, HW, heads, 3 * C, C, 0.125), {}), dict(o=torch.zeros(M, C, dtype=dt, device=P.dev)))
    e = kc.relerr(o1["o"], o2["o"])
    o1, o2 = P.run("tattn_bwd", lambda o: ((q, k, v, d_o, o["d"], o["d"][:, C:], o["d"][:, 2 * C:], B, T, HW, heads, 3 * C, C, 3 * C, 0.125), {}),
                   dict(d=torch.zeros(M, 3 * C, dtype=dt, device=P.dev)))
    for i in range(3):
        e = max(e, 0.5 * kc.relerr(o1["d"][:, i * C:(i + 1) * C], o2["d"][:, i * C:(i + 1) * C]))
    return desc, e, kc.tol_for(dt)


def fuzz_tsa(P, dt, rng, g):
    B, T, heads = rng.randint(1, 2), rng.randint(1, 16), rng.randint(1, 5)
    HW = rng.choice((1, 2, 3, 5, 7, 9, 10, 12, 16, 20, 27, 36, 48, 60, 64, 77, 100))
    mod = B if (B > 1 and rng.random() < 0.5) else 0
    C, M = heads * 64, B * T * HW
    x = kc.rnd((M, C), dt, P.dev, g)
    x[:, :8] += 3.0
    gamma, beta = 1.0 + 0.1 * kc.rndf((C,), P.dev, g), 0.1 * kc.rndf((C,), P.dev, g)
    wqkv, wo = kc.rnd((3 * C, C), dt, P.dev, g, C ** -0.5), kc.rnd((C, C), dt, P.dev, g, C ** -0.5)
    bo, cvec = 0.1 * kc.rndf((C,), P.dev, g), kc.rndf((B, C), P.dev, g)
    rpg = 0 if mod else T * HW
    outs = dict(n1=torch.zeros(M, C, dtype=dt, device=P.dev), st=torch.zeros(M, 2, device=P.dev), qkv=torch.zeros(M, 3 * C, dtype=dt, device=P.dev),
                o=torch.zeros(M, C, dtype=dt, device=P.dev), h1=torch.zeros(M, C, dtype=dt, device=P.dev))
    desc = f"tsa B={B} T={T} HW={HW} heads={heads} mod={mod}"
    o1, o2 = P.run("tsa_fwd", lambda o: ((x, gamma, beta, 1e-5, wqkv, wo, bo, cvec, C, rpg, mod, o["n1"], o["st"], o["qkv"], o["o"], o["h1"], B, T, HW, C, heads, 0.125), {}), outs)
    e = max(kc.relerr(o1["n1"], o2["n1"]), 0.5 * kc.relerr(o1["qkv"], o2["qkv"]), 0.5 * kc.relerr(o1["o"], o2["o"]), 0.5 * kc.relerr(o1["h1"], o2["h1"]))
    return desc, e, kc.tol_for(dt)


def fuzz_lnbwd(P, dt, rng, g):
    rows, C = pick_dim(rng, 1, 5000), 64 * rng.randint(1, 20)
    x = (kc.rnd((rows, C), dt, P.dev, g) * 2 + 0.5).to(dt)
    dy, add = kc.rnd((rows, C), dt, P.dev, g), kc.rnd((rows, C), dt, P.dev, g)
    gamma, beta = 1 + 0.1 * kc.rndf((C,), P.dev, g), 0.1 * kc.rndf((C,), P.dev, g)
    o1, o2 = P.run("ln_fwd", lambda o: ((x, gamma, beta, o["y"], o["st"], rows, C, 1e-5), {}), dict(y=torch.zeros_like(x), st=torch.zeros(rows, 2, device=P.dev)))
    st = o2["st"]
    affine = rng.choice((False, True, "scratch"))
    outs = dict(dx=torch.zeros_like(x), dg=torch.ones(C, device=P.dev), db=torch.ones(C, device=P.dev))
    scr = torch.full((K.LN_PARTIAL_ROWS * 2 * C,), float("nan"), device=P.dev) if affine == "scratch" else None
    desc = f"ln_bwd rows={rows} C={C} affine={affine}"
    o1, o2 = P.run("ln_bwd", lambda o: ((dy, x, st, gamma, add if affine else None, o["dx"], o["dg"] if affine else None, o["db"] if affine else None, rows, C),
                                        dict(scratch=scr, add2=dy if affine == "scratch" else None, add2_scale=0.37)), outs)
    e = kc.relerr(o1["dx"], o2["dx"])
    if affine:
        e = max(e, kc.relerr(o1["dg"], o2["dg"]), kc.relerr(o1["db"], o2["db"]))
    return desc, e, kc.tol_for(dt)


def fuzz_small(P, dt, rng, g):
    M, N, Kd = rng.randint(1, 30), pick_dim(rng, 1, 1500), 8 * rng.randint(1, 200)
    X, W, b = kc.rndf((M, Kd), P.dev, g), kc.rnd((N, Kd), dt, P.dev, g, Kd ** -0.5), kc.rndf((N,), P.dev, g)
    silu_in, acc = rng.randint(0, 1), rng.randint(0, 1)
    desc = f"small_linear M={M} N={N} K={Kd} silu={silu_in} acc={acc}"
    o1, o2 = P.run("small_linear", lambda o: ((X, W, b, o["Y"], M, N, Kd, Kd), dict(trans=0, silu_in=silu_in, accumulate=acc)), dict(Y=torch.ones(M, N, device=P.dev)))
    e = kc.relerr(o1["Y"], o2["Y"])
    dY = kc.rndf((M, N), P.dev, g)
    o1, o2 = P.run("small_linear", lambda o: ((dY, W, None, o["Y"], M, N, Kd, Kd), dict(trans=1)), dict(Y=torch.zeros(M, Kd, device=P.dev)))
    e = max(e, kc.relerr(o1["Y"], o2["Y"]))
    o1, o2 = P.run("outer_acc", lambda o: ((dY, X, o["W"], M, N, Kd), dict(scale=0.5)), dict(W=torch.ones(N, Kd, device=P.dev)))
    e = max(e, kc.relerr(o1["W"], o2["W"]))
    return desc, e * 20, kc.tol_for(dt)          # float kernels: held to 1e-4 (fp16 bar / 20)


def fuzz_batch(P, dt, rng, g):
    """table-driven skinny launches: random job tables (1..60 jobs: the 48-job pack boundary included) -- each job must equal its
    single-job launch bit for bit; and svdx_ln_bwd(defer_reduce) + svdx_ln_param_reduce_batch against the immediate reduction"""
    M, nj = rng.randint(1, 12), rng.choice((1, 2, 3, 5, 17, 47, 48, 49, 60))
    shapes = [(pick_dim(rng, 1, 400), 8 * rng.randint(1, 40)) for _ in range(nj)]
    kind = rng.choice(("nt", "nn", "outer", "lnred"))
    desc = f"batch {kind} M={M} jobs={nj} first={shapes[:3]}"
    if kind == "lnred":
        nj = min(nj, 6)
        jobs, pairs = [], []
        for _ in range(nj):
            rows, C = pick_dim(rng, 1, 3000), 64 * rng.randint(1, 20)
            x = (kc.rnd((rows, C), dt, P.dev, g) * 2 + 0.5).to(dt)
            dy = kc.rnd((rows, C), dt, P.dev, g)
            gamma, beta = 1 + 0.1 * kc.rndf((C,), P.dev, g), 0.1 * kc.rndf((C,), P.dev, g)
            y, st = torch.zeros_like(x), torch.zeros(rows, 2, device=P.dev)
            P.impl.ln_fwd(x, gamma, beta, y, st, rows, C, 1e-5)
            a = [torch.zeros_like(x), torch.ones(C, device=P.dev), torch.ones(C, device=P.dev)]
            b = [t.clone() for t in a]
            P.impl.ln_bwd(dy, x, st, gamma, None, a[0], a[1], a[2], rows, C, scratch=torch.full((K.LN_PARTIAL_ROWS * 2 * C,), float("nan"), device=P.dev))
            nblk = K.ln_bwd_blocks(rows, C)
            scr = torch.full((nblk * 2 * C,), float("nan"), device=P.dev)
            P.impl.ln_bwd(dy, x, st, gamma, None, b[0], b[1], b[2], rows, C, scratch=scr, defer_reduce=True)
            jobs.append((scr, b[1], b[2], nblk, C))
            pairs.append((a, b))
        P.impl.ln_param_reduce_batch(jobs)
        same = all(torch.equal(u, v) for a, b in pairs for u, v in zip(a, b))
        return desc, 0.0 if same else 1.0, 0.0
    Xs = [kc.rndf((M, Kd), P.dev, g) for N, Kd in shapes]
    Ws = [kc.rnd((N, Kd), dt, P.dev, g, Kd ** -0.5) for N, Kd in shapes]
    bs = [kc.rndf((N,), P.dev, g) if rng.random() < 0.5 else None for N, Kd in shapes]
    dYs = [kc.rndf((M, N), P.dev, g) for N, Kd in shapes]
    fl = [(rng.random() < 0.3, rng.random() < 0.5) for _ in shapes]
    if kind == "outer":
        ones_col = [rng.random() < 0.3 for _ in shapes]
        single = [torch.ones(N, 1 if oc else Kd, device=P.dev) for (N, Kd), oc in zip(shapes, ones_col)]
        batch = [t.clone() for t in single]
        ones = torch.ones(M, 1, device=P.dev)
        for i, (N, Kd) in enumerate(shapes):
            P.impl.outer_acc(dYs[i], ones if ones_col[i] else Xs[i], single[i], M, N, 1 if ones_col[i] else Kd, 0.25)
        P.impl.outer_acc_batch([(dYs[i], None if ones_col[i] else Xs[i], batch[i], N, 1 if ones_col[i] else Kd, 0.25) for i, (N, Kd) in enumerate(shapes)], M)
    else:
        trans = int(kind == "nn")
        single = [torch.ones(M, Kd if trans else N, device=P.dev) for N, Kd in shapes]
        batch = [t.clone() for t in single]
        for i, (N, Kd) in enumerate(shapes):
            if trans:
                P.impl.small_linear(dYs[i], Ws[i], None, single[i], M, N, Kd, Kd, 1, 0, int(fl[i][1]))
            else:
                P.impl.small_linear(Xs[i], Ws[i], bs[i], single[i], M, N, Kd, Kd, 0, int(fl[i][0]), int(fl[i][1]))
        P.impl.small_linear_batch([((dYs[i], Ws[i], None, batch[i], N, Kd, Kd, False, fl[i][1]) if trans else
                                    (Xs[i], Ws[i], bs[i], batch[i], N, Kd, Kd, fl[i][0], fl[i][1])) for i, (N, Kd) in enumerate(shapes)], M, trans)
    return desc, 0.0 if all(torch.equal(a, b) for a, b in zip(single, batch)) else 1.0, 0.0


def fuzz_rows(P, dt, rng, g):
    """row-vector / column-sum / transpose / concat kernels on ragged sizes"""
    rows, C = pick_dim(rng, 1, 2500), 8 * rng.randint(1, 80)
    ng = rng.randint(1, 6)
    x = kc.rnd((rows, C), dt, P.dev, g)
    vec = kc.rndf((ng, 2 * C), P.dev, g)
    rpg, mod = ((rows + ng - 1) // ng, 0) if rng.random() < 0.5 else (0, ng)
    desc = f"rows rows={rows} C={C} groups={ng} rpg={rpg} mod={mod}"
    o1, o2 = P.run("add_rowvec", lambda o: ((x, vec[:, C:], o["y"], rows, C, 2 * C, rpg, mod), {}), dict(y=torch.zeros_like(x)))
    e = kc.relerr(o1["y"], o2["y"])
    acc = rng.randint(0, 1)
    scr = torch.full((K.colsum_slabs(rows, rpg, mod) * ng * C,), float("nan"), device=P.dev) if rng.random() < 0.7 else None
    o1, o2 = P.run("colsum", lambda o: ((x, o["s"], rows, C, C, ng, rpg, mod), dict(accumulate=acc, scratch=scr)), dict(s=torch.ones(ng, C, device=P.dev)))
    e = max(e, 2 * kc.relerr(o1["s"], o2["s"]))
    r, c = pick_dim(rng, 1, 400), pick_dim(rng, 1, 400)
    xx = kc.rnd((r, c + 8), dt, P.dev, g)
    ldo = (r + 63) // 64 * 64
    o1, o2 = P.run("transpose", lambda o: ((xx, c + 8, o["t"], ldo, r, c), {}), dict(t=torch.full((c, ldo), 3.0, dtype=dt, device=P.dev)))
    e = max(e, 1e3 * kc.relerr(o1["t"], o2["t"]))
    wf = kc.rndf((r, c), P.dev, g)
    o1, o2 = P.run("cast_transpose_from_f32", lambda o: ((wf, o["t"], r, c), {}), dict(t=torch.zeros(c, r, dtype=dt, device=P.dev)))
    e = max(e, 1e3 * kc.relerr(o1["t"], o2["t"]))
    ca, cb = 8 * rng.randint(1, 40), 8 * rng.randint(1, 40)
    a2, b2 = kc.rnd((rows, ca), dt, P.dev, g), kc.rnd((rows, cb), dt, P.dev, g)
    o1, o2 = P.run("concat2", lambda o: ((a2, ca, b2, cb, o["c"], rows), {}), dict(c=torch.zeros(rows, ca + cb, dtype=dt, device=P.dev)))
    e = max(e, 1e3 * kc.relerr(o1["c"], o2["c"]))
    cat = o2["c"]
    o1, o2 = P.run("split2", lambda o: ((cat, o["a"], ca, o["b"], cb, rows), {}), dict(a=torch.zeros(rows, ca, dtype=dt, device=P.dev), b=torch.zeros(rows, cb, dtype=dt, device=P.dev)))
    e = max(e, 1e3 * kc.relerr(o1["a"], o2["a"]), 1e3 * kc.relerr(o1["b"], o2["b"]))
    return desc, e, kc.tol_for(dt)


def fuzz_optim(P, dt, rng, g):
    n = 4 * rng.randint(1, 6000)
    p, gr = kc.rndf((n,), P.dev, g), kc.rndf((n,), P.dev, g, 100.0)
    m, v = kc.rndf((n,), P.dev, g, 0.1), kc.rndf((n,), P.dev, g).abs()
    found = rng.random() < 0.25
    if found:
        gr[rng.randrange(n)] = float("inf") if rng.random() < 0.5 else float("nan")
    sched = rng.choice(([0, 0, 0, 0, 0, 0, 0], [1, 10, 0, 0, 0, 0, 1], [2, 2, 40, 0, 0, 0, 2], [3, 1, 20, 0.5, 0, 0, 1], [4, 1, 20, 3, 0, 0, 2], [5, 2, 30, 0, 2.0, 1e-2, 1]))
    st0 = torch.tensor([rng.randint(0, 9), 1024.0, rng.randint(0, 9), 0, 1, 1, 1, 0, 1.0] + sched, dtype=torch.float32, device=P.dev)
    outs = dict(st=st0, p=p, m=m, v=v, pa=torch.zeros(n, dtype=dt, device=P.dev))

    def seq(be, o):
        be.check_finite(gr, n, o["st"])
        be.optim_prep(o["st"], 0.9, 0.999, 2.0, 0.5, 7, 1)
        be.adamw(o["p"], gr, o["m"], o["v"], n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 0.5, o["st"], o["pa"])
    o1 = {k_: t.clone() for k_, t in outs.items()}
    o2 = {k_: t.clone() for k_, t in outs.items()}
    seq(P.impl, o1)
    seq(P.ref, o2)
    desc = f"optim n={n} found_inf={found} sched={sched}"
    e = max(kc.relerr(o1["st"], o2["st"]), kc.relerr(o1["p"], o2["p"]), kc.relerr(o1["m"], o2["m"]), kc.relerr(o1["v"], o2["v"])) * 100
    e = max(e, kc.relerr(o1["pa"], o2["pa"]))
    sh, w = kc.rndf((n + 3,), P.dev, g), kc.rndf((n + 3,), P.dev, g)
    o1, o2 = P.run("ema_lerp", lambda o: ((o["s"], w, n + 3, 0.013), {}), dict(s=sh))
    e = max(e, 1e3 * kc.relerr(o1["s"], o2["s"]))
    return desc, e, kc.tol_for(dt)


def fuzz_gemm_gn(P, dt, rng, g):
    """svdx_gemm_gn / svdx_gemm_finalize_gn: the result against the emulation, the statistics against a statistics pass over the tensor the
    launch itself wrote (same rounded values: the 64-bit fixed-point sums must agree to the float decode's last bits)."""
    from svd_xtend_amd.ops import _tile_launched, gn_tile_ok
    v = rng.choice((4, 6, 7, 8, 16, 17, 18, 20, 21, 22, 23, 24, 25, 26))
    cg = rng.choice((4, 8, 10, 20, 40, 80))
    N = cg * rng.choice((8, 16, 32))
    n_s, rows = rng.randint(1, 6), pick_dim(rng, 8, 400)
    Kd = 64 * rng.randint(1, 6)
    M, G = n_s * rows, N // cg
    split = rng.random() < 0.35
    A, B = kc.rnd((M, Kd), dt, P.dev, g), kc.rnd((N, Kd), dt, P.dev, g, Kd ** -0.5)
    bias, R, rv = kc.rndf((N,), P.dev, g), kc.rnd((M, N), dt, P.dev, g), kc.rndf((n_s, N), P.dev, g)
    mode = rng.choice(("plain", "bias_res", "rowvec"))
    ep = dict(bias=bias, res=R, ldres=N) if mode == "bias_res" else dict(bias=bias, rowvec=rv, rv_ld=N, rv_rpg=rows) if mode == "rowvec" else {}
    st1, st2 = (torch.zeros(K.GN_REPLICAS, n_s, G, K.GN_STAT_FLOATS, device=P.dev) for _ in range(2))
    c1, c2 = (torch.zeros(M, N, dtype=dt, device=P.dev) for _ in range(2))
    if split:
        sk = rng.choice([x for x in (2, 3, 4, 8) if x <= max(2, Kd // 64)])
        if sk > Kd // 64 or N * rows < 1024:
            raise K.SvdxError("not a split-K statistics case")
        for be, c, st in ((P.impl, c1, st1), (P.ref, c2, st2)):
            acc = torch.zeros(sk, M, N, device=P.dev)
            be.gemm(A, B, acc, M, N, Kd, Kd, Kd, N, out_mode=K.OUT_F32_SLAB, split_k=sk, variant=v)
            be.gemm_finalize(acc, sk, M * N, c, M, N, N, gn=(st, rows, cg), **ep)
        desc = f"gemm_finalize_gn v{v} n_s={n_s} rows={rows} N={N} cg={cg} K={Kd} sk={sk} {mode}"
    else:
        if not gn_tile_ok(_tile_launched(v, M, N), N, rows, cg):
            raise K.SvdxError("tile does not take the statistics path")
        P.impl.gemm(A, B, c1, M, N, Kd, Kd, Kd, N, variant=v, gn=(st1, rows, cg), **ep)
        P.ref.gemm(A, B, c2, M, N, Kd, Kd, Kd, N, variant=v, gn=(st2, rows, cg), **ep)
        desc = f"gemm_gn v{v} n_s={n_s} rows={rows} N={N} cg={cg} K={Kd} {mode}"
    own = torch.zeros_like(st1)
    P.ref.gn_stats(c1, own, n_s, rows, N, G, prezeroed=1)
    e_st = kc.relerr(emul.gn_decode(st1, n_s, G, rows * cg, 0).view(-1, 2), emul.gn_decode(own, n_s, G, rows * cg, 0).view(-1, 2))
    return desc, max(kc.relerr(c1, c2), e_st * (kc.tol_for(dt) / 1e-4)), kc.tol_for(dt)


def fuzz_gradfin(P, dt, rng, g):
    """svdx_grad_finalize_batch: random job tables (slices, sizes, store / accumulate, with and without column sums) against one
    svdx_gemm_finalize launch per job -- BIT equality (the same additions in the same order)."""
    jobs, singles = [], []
    for _ in range(rng.randint(1, 60)):
        sk, rows_, cols = rng.randint(1, 9), rng.randint(1, 40), 4 * rng.randint(1, 60)
        n = rows_ * cols
        acc = kc.rndf((sk, n), P.dev, g)
        store = rng.random() < 0.5
        d0 = kc.rndf((n,), P.dev, g)
        if rng.random() < 0.4:
            cs, co0 = kc.rndf((sk, rows_), P.dev, g), kc.rndf((rows_,), P.dev, g)
        else:
            cs, co0 = None, None
        jobs.append((acc, sk, n, d0, cs, co0, store, rows_, cols))
    outs = []
    for be, batched in ((P.impl, True), (P.impl, False)):
        ds = [j[3].clone() for j in jobs]
        cos = [None if j[5] is None else j[5].clone() for j in jobs]
        if batched:
            be.grad_finalize_batch([(acc, sk, n, d, n, cs, co, store) for (acc, sk, n, _d, cs, _c, store, _r, _k), d, co in zip(jobs, ds, cos)])
        else:
            for (acc, sk, n, _d, cs, _c, store, rows_, cols), d, co in zip(jobs, ds, cos):
                be.gemm_finalize(acc, sk, n, d, rows_, cols, cols, accumulate_f32=2 if store else 1, dtype=dt, colsum_slabs=cs, colsum_out=co)
        outs.append((ds, cos))
    same = all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0])) and all(a is None or torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1]))
    # ... and the emulation agrees to rounding
    ds = [j[3].clone() for j in jobs]
    cos = [None if j[5] is None else j[5].clone() for j in jobs]
    P.ref.grad_finalize_batch([(acc, sk, n, d, n, cs, co, store) for (acc, sk, n, _d, cs, _c, store, _r, _k), d, co in zip(jobs, ds, cos)])
    e = max(kc.relerr(a, b) for a, b in zip(outs[0][0], ds))
    return f"gradfin {len(jobs)} jobs", (e if same else float("inf")), 1e-5


def fuzz_attn_small(P, dt, rng, g):
    """svdx_attn_small_fwd: any head dimension up to 128 (real channels d inside dp packed columns), any S, wide rows either side."""
    n, heads, S = rng.randint(1, 2), rng.randint(1, 3), pick_dim(rng, 1, 300)
    dp = 8 * rng.randint(1, 16)
    d = dp if rng.random() < 0.6 else rng.randint(max(1, dp - 7), dp)
    ld, ld_o = 3 * heads * dp + 8 * rng.randint(0, 2), heads * dp + 4 * rng.randint(0, 2)
    qkv = kc.rnd((n * S, ld), dt, P.dev, g, rng.choice((0.5, 1.5, 3.0)))
    if d < dp:                                   # the padding columns of every head hold zeros (what the packed projection produces)
        cols = torch.arange(3 * heads * dp) % dp >= d
        qkv[:, :3 * heads * dp][:, cols] = 0
    o1, o2 = P.run("attn_small_fwd", lambda o: ((qkv, o["y"], n, S, heads, d, dp, ld, ld_o, d ** -0.5), {}),
                   dict(y=torch.ones(n * S, ld_o, dtype=dt, device=P.dev)))
    return f"attn_small n={n} heads={heads} S={S} d={d} dp={dp} ld={ld} ld_o={ld_o}", kc.relerr(o1["y"], o2["y"]), kc.tol_for(dt)


FAMILIES = {"gemm": fuzz_gemm, "gemm_gn": fuzz_gemm_gn, "gradfin": fuzz_gradfin, "gather": fuzz_gather, "tn": fuzz_tn, "geglu": fuzz_geglu, "norm": fuzz_norm, "lnbwd": fuzz_lnbwd,
            "attn": fuzz_attn, "attn_small": fuzz_attn_small, "tattn": fuzz_tattn, "tsa": fuzz_tsa, "small": fuzz_small, "batch": fuzz_batch, "rows": fuzz_rows, "optim": fuzz_optim}


def run(P, dt, families, n, seed, verbose=True):
    bad = []
    for fam in families:
        rng = random.Random(f"{fam}-{seed}")
        g = torch.Generator().manual_seed(seed)
        t0 = time.time()
        for i in range(n):
            state = rng.getstate()
            try:
                desc, e, tol = FAMILIES[fam](P, dt, rng, g)
            except K.SvdxError as ex:           # an argument check refused the combination: not a wrong result
                if verbose and os.environ.get("SVDX_FUZZ_SHOW_REFUSED"):
                    print(f"  refused: {ex}")
                continue
            if not (e <= tol and math.isfinite(e)):
                bad.append((fam, i, desc, e, tol))
                if verbose:
                    print(f"  BAD [{fam} #{i}] {desc}: err {e:.3g} > {tol:.3g}", flush=True)
            del state
        if verbose:
            print(f"{fam:8s} {str(dt):15s} {n} cases, {sum(1 for b in bad if b[0] == fam)} bad, {time.time() - t0:.1f}s", flush=True)
    return bad


if __name__ == "__main__":
    from backend import SimBackend
    args = sys.argv[1:]
    n = int(args[args.index("--n") + 1]) if "--n" in args else 60
    seed = int(args[args.index("--seed") + 1]) if "--seed" in args else 0
    fams = [a for a in args if a in FAMILIES] or list(FAMILIES)
    P = kc.Pair(SimBackend(), torch.device("cpu"))
    bad = run(P, torch.bfloat16 if "--bf16" in args else torch.float16, fams, n, seed)
    sys.exit(1 if bad else 0)
