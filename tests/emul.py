"""Plain-PyTorch emulation of every libsvdx entry point (svd_xtend_amd/kernels.py interface).

TEST INFRASTRUCTURE: (1) the fp32 reference each HIP kernel is compared against in the `-m gpu` tests,
(2) a CPU stand-in so the host-side orchestration (explicit forward/backward schedules, packing, flat
buffers, DP) can be checked against the oracle without a GPU.  Pointer semantics are reproduced with
as_strided on the tensor's storage, so views with offsets behave exactly like `base + offset` pointers.
"""
import math

import torch

from svd_xtend_amd import kernels as K

GN_REPLICAS = K.GN_REPLICAS


def V(t, rows, cols, ld):
    return torch.as_strided(t, (rows, cols), (ld, 1), t.storage_offset())


def V1(t, n):
    return torch.as_strided(t, (n,), (1,), t.storage_offset())


def gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x * 0.7071067811865476))


def gelu_grad(x):
    return 0.5 * (1.0 + torch.erf(x * 0.7071067811865476)) + x * torch.exp(-0.5 * x * x) * 0.3989422804014327


def silu(x):
    return x * torch.sigmoid(x)


def silu_grad(x):
    s = torch.sigmoid(x)
    return s * (1 + x * (1 - s))


def gather_rows(A, g: K.Gather, M):
    """Effective [M, taps*cin] A operand of the implicit GEMM (float)."""
    dev = A.device
    m = torch.arange(M, device=dev)
    cols = []
    if g.mode in (K.GATHER_CONV3X3, K.GATHER_CONV3X3_DGRAD2, K.GATHER_CONV3X3_PAD0):
        x = m % g.wo
        y = (m // g.wo) % g.ho
        n = m // (g.wo * g.ho)
        for dy in range(3):
            for dx in range(3):
                if g.mode in (K.GATHER_CONV3X3, K.GATHER_CONV3X3_PAD0):
                    pad = 1 if g.mode == K.GATHER_CONV3X3 else 0
                    ys, xs = y * g.stride + dy - pad, x * g.stride + dx - pad
                    valid = (ys >= 0) & (ys < g.hi) & (xs >= 0) & (xs < g.wi)
                    if g.ups:
                        row = (n * (g.hi // 2) + ys.clamp(min=0) // 2) * (g.wi // 2) + xs.clamp(min=0) // 2
                    else:
                        row = (n * g.hi + ys) * g.wi + xs
                else:
                    y2, x2 = y + 1 - dy, x + 1 - dx
                    valid = (y2 >= 0) & (y2 % 2 == 0) & (y2 // 2 < g.hi) & (x2 >= 0) & (x2 % 2 == 0) & (x2 // 2 < g.wi)
                    row = (n * g.hi + y2 // 2) * g.wi + x2 // 2
                cols.append((row, valid))
        nsrc = g.n_img * (g.hi // 2 if g.ups else g.hi) * (g.wi // 2 if g.ups else g.wi)
    elif g.mode == K.GATHER_TEMPORAL3:
        p = m % g.hw
        t = (m // g.hw) % g.t
        b = m // (g.hw * g.t)
        for dt in range(3):
            ts = t + dt - 1
            valid = (ts >= 0) & (ts < g.t)
            cols.append(((b * g.t + ts) * g.hw + p, valid))
        nsrc = g.n_img * g.t * g.hw
    else:
        raise ValueError(g.mode)
    src = V(A, nsrc, g.cin, g.lda).float()
    out = []
    for row, valid in cols:
        r = src[row.clamp(0, nsrc - 1)]
        out.append(torch.where(valid[:, None], r, torch.zeros_like(r)))
    return torch.cat(out, 1)


def gn_fixed_scales(cnt: int, mode: int):
    """include/svdx.h GroupNorm statistics: 64-bit fixed point, scale 2^k per (kind, count) -- csrc/norm.hip gn_fixed_scales."""
    lg = 0
    while (1 << lg) < cnt:
        lg += 1
    b0, b1 = (16, 24) if mode == 0 else (18, 26)
    return min(max(62 - b0 - lg, 0), 40), min(max(62 - b1 - lg, 0), 40)


def gn_view(stats, n_s, G):
    """int64 [replicas, n_s, G, 2] view of a statistics buffer (a float tensor of K.GN_STAT_FLOATS floats per entry)."""
    return V1(stats, GN_REPLICAS * n_s * G * K.GN_STAT_FLOATS).view(torch.int64).view(GN_REPLICAS, n_s, G, 2)


def gn_decode(stats, n_s, G, cnt, mode):
    k0, k1 = gn_fixed_scales(cnt, mode)
    tot = gn_view(stats, n_s, G).sum(0)
    return torch.stack([tot[..., 0].double() * 2.0 ** -k0, tot[..., 1].double() * 2.0 ** -k1], -1).float()


def gn_encode_add(stats, n_s, G, cnt, mode, s0, s1):
    k0, k1 = gn_fixed_scales(cnt, mode)
    v = gn_view(stats, n_s, G)
    v[0, ..., 0] += torch.round(s0.double() * 2.0 ** k0).to(torch.int64)
    v[0, ..., 1] += torch.round(s1.double() * 2.0 ** k1).to(torch.int64)


class EmuBackend:
    # ---- GEMM family ----
    def gemm(self, A, B, C, M, N, Kd, lda, ldb, ldc, bias=None, rowvec=None, rv_ld=0, rv_rpg=0, rv_mod=0,
             res=None, ldres=0, gather=None, out_mode=K.OUT_ACT, alpha=1.0, split_k=1, variant=0, epilogue=0, aux_in=None,
             aux_out=None, aux_dim=0, dual=None, gn=None):
        assert Kd % 64 == 0, "GEMM K must be a multiple of 64"
        assert gn is None or (dual is None and split_k == 1 and epilogue == 0 and out_mode == K.OUT_ACT)
        if gather is None or gather.mode == K.GATHER_PLAIN:
            a = V(A, M, Kd, lda).float()
        else:
            a = gather_rows(A, gather, M)
            assert a.shape[1] == Kd
        b = V(B, N, Kd, ldb).float()
        v = a @ b.t()
        if dual is not None:
            A2, B2, K2, lda2, ldb2 = dual[:5]
            seg = dual[5] if len(dual) > 5 else 0
            assert K2 % 64 == 0 and split_k == 1 and epilogue == K.EPI_NONE
            b2 = V(B2, N, K2, ldb2).float()
            if seg:
                assert N % seg == 0 and seg % 128 == 0 or seg % 160 == 0
                a2 = V(A2, M, (N // seg) * K2, lda2).float()
                v = v + torch.cat([a2[:, j * K2:(j + 1) * K2] @ b2[j * seg:(j + 1) * seg].t() for j in range(N // seg)], 1)
            else:
                v = v + V(A2, M, K2, lda2).float() @ b2.t()
        v = alpha * v
        if bias is not None:
            v = v + V1(bias, N)[None]
        if rowvec is not None:
            m = torch.arange(M, device=A.device)
            gi = (m % rv_mod) if rv_mod else (m // rv_rpg)
            ng = int(gi.max()) + 1
            v = v + V(rowvec, ng, N, rv_ld)[gi]
        if res is not None:
            v = v + V(res, M, N, ldres).float()
        if epilogue == K.EPI_GEGLU_FWD:
            Fd = aux_dim
            pre = (alpha * (a @ b.t()) + (V1(bias, N)[None] if bias is not None else 0)).to(C.dtype)
            V(C, M, N, ldc).copy_(pre)
            pf = pre.float()
            V(aux_out, M, Fd, Fd).copy_((pf[:, :Fd] * gelu(pf[:, Fd:])).to(C.dtype))
            return
        if epilogue == K.EPI_GEGLU_BWD:
            Fd = aux_dim
            dh = (alpha * (a @ b.t()) + (V1(bias, N)[None] if bias is not None else 0)).to(C.dtype).float()
            pf = V(aux_in, M, 2 * Fd, 2 * Fd).float()
            o = V(C, M, 2 * Fd, ldc)
            o[:, :Fd] = (dh * gelu(pf[:, Fd:])).to(C.dtype)
            o[:, Fd:] = (dh * pf[:, :Fd] * gelu_grad(pf[:, Fd:])).to(C.dtype)
            return
        if out_mode == K.OUT_F32_SLAB:
            assert bias is None and rowvec is None and res is None and C.dtype == torch.float32 and ldc == N
            ksz = (Kd // 64 + split_k - 1) // split_k * 64
            for z in range(split_k):
                sl = torch.as_strided(C, (M, N), (N, 1), C.storage_offset() + z * M * N)
                sl.copy_(alpha * (a[:, z * ksz:(z + 1) * ksz] @ b[:, z * ksz:(z + 1) * ksz].t()))
            return
        c = V(C, M, N, ldc)
        if out_mode in (K.OUT_F32_ATOMIC, K.OUT_F32_ADD):
            assert C.dtype == torch.float32
            c += v
        else:
            assert C.dtype == torch.float32 or out_mode == K.OUT_ACT
            c.copy_(v.to(C.dtype))
            if gn is not None:           # svdx_gemm_gn: statistics of the ROUNDED tensor, added to the (zeroed) buffer
                stats, rows, cg = gn
                assert M % rows == 0 and N % cg == 0
                xf = c.float().reshape(M // rows, rows, N // cg, cg)
                gn_encode_add(stats, M // rows, N // cg, rows * cg, 0, xf.sum((1, 3)), (xf * xf).sum((1, 3)))

    def gemm_tn(self, A, B, C, R, N, Kd, lda, ldb, ldc, out_mode=K.OUT_F32_ADD, split_k=1, a_colsum=None, stages=0, found_inf=None):
        a, b = V(A, R, N, lda).float(), V(B, R, Kd, ldb).float()
        if a_colsum is not None and out_mode != K.OUT_F32_SLAB:
            V1(a_colsum, N).add_(a.sum(0))
        if out_mode == K.OUT_F32_SLAB:
            assert ldc == Kd
            rt = (R + 63) // 64
            per = (rt + split_k - 1) // split_k * 64
            for z in range(split_k):
                sl = torch.as_strided(C, (N, Kd), (Kd, 1), C.storage_offset() + z * N * Kd)
                sl.copy_(a[z * per:(z + 1) * per].t() @ b[z * per:(z + 1) * per])
                if a_colsum is not None:                  # float[split_k][N]: slice z stores its partial column sums
                    V(a_colsum, split_k, N, N)[z].copy_(a[z * per:(z + 1) * per].sum(0))
            return
        v = a.t() @ b
        c = V(C, N, Kd, ldc)
        if out_mode == K.OUT_F32:
            c.copy_(v)
        else:
            c += v
        if found_inf is not None and not torch.isfinite(c).all():
            found_inf[0] = 1.0

    def gemm_finalize(self, acc, nsplit, slab_stride, C, M, N, ldc, bias=None, rowvec=None, rv_ld=0, rv_rpg=0, rv_mod=0,
                      res=None, ldres=0, accumulate_f32=False, dtype=None, colsum_slabs=None, colsum_out=None, gn=None):
        assert gn is None or (not accumulate_f32 and colsum_slabs is None)
        if colsum_slabs is not None:
            n = colsum_out.numel()
            V1(colsum_out, n).add_(V(colsum_slabs, nsplit, n, n).sum(0))
        v = torch.zeros(M, N, device=acc.device)
        for z in range(nsplit):
            v = v + torch.as_strided(acc, (M, N), (N, 1), acc.storage_offset() + z * slab_stride)
        if bias is not None:
            v = v + V1(bias, N)[None]
        if rowvec is not None:
            m = torch.arange(M, device=acc.device)
            gi = (m % rv_mod) if rv_mod else (m // rv_rpg)
            v = v + V(rowvec, int(gi.max()) + 1, N, rv_ld)[gi]
        if res is not None:
            v = v + V(res, M, N, ldres).float()
        if int(accumulate_f32) == 1:
            V(C, M, N, ldc).add_(v)
        else:                                        # 0: activation store, 2: float store
            V(C, M, N, ldc).copy_(v.to(C.dtype))
            if gn is not None:                       # svdx_gemm_finalize_gn: statistics of the rounded tensor
                stats, rows, cg = gn
                assert M % rows == 0 and N % cg == 0 and N * rows >= 1024 and N // cg <= 64
                xf = V(C, M, N, ldc).float().reshape(M // rows, rows, N // cg, cg)
                gn_encode_add(stats, M // rows, N // cg, rows * cg, 0, xf.sum((1, 3)), (xf * xf).sum((1, 3)))

    def grad_finalize_batch(self, jobs):
        for job in jobs:
            acc, nsplit, stride, dst, count, cs, co, store = job[:8]
            found = job[8] if len(job) > 8 else None
            v = torch.zeros(count, device=acc.device)
            for z in range(nsplit):
                v = v + torch.as_strided(acc, (count,), (1,), acc.storage_offset() + z * stride)
            d = V1(dst, count)
            if store:
                d.copy_(v)
            else:
                d.add_(v)
            if found is not None and not torch.isfinite(d).all():
                found[0] = 1.0
            if cs is not None:
                n = co.numel()
                V1(co, n).add_(V(cs, nsplit, n, n).sum(0))

    def small_linear(self, X, W, bias, Y, M, N, Kd, ldw, trans=0, silu_in=0, accumulate=0):
        w = V(W, N, Kd, ldw).float()
        if trans == 0:
            x = V(X, M, Kd, Kd)
            if silu_in:
                x = silu(x)
            y = x @ w.t()
            if bias is not None:
                y = y + V1(bias, N)[None]
            out = V(Y, M, N, N)
        else:
            y = V(X, M, N, N) @ w
            out = V(Y, M, Kd, Kd)
        if accumulate:
            out += y
        else:
            out.copy_(y)

    def outer_acc(self, dY, X, dW, M, N, Kd, scale=1.0):
        V(dW, N, Kd, Kd).add_(scale * (V(dY, M, N, N).t() @ V(X, M, Kd, Kd)))

    def small_linear_batch(self, jobs, M, trans=0):
        for X, W, b, Y, N, Kd, ldw, si, acc in jobs:
            self.small_linear(X, W, b, Y, M, N, Kd, ldw, trans, int(bool(si)), int(bool(acc)))

    def outer_acc_batch(self, jobs, M):
        for dY, X, dW, N, Kd, sc in jobs:
            x = X if X is not None else torch.ones(M, 1, dtype=torch.float32, device=dY.device)
            self.outer_acc(dY, x, dW, M, N, Kd, sc)

    def timestep_embed(self, t, out, n, dim):
        half = dim // 2
        f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
        e = V1(t, n)[:, None] * f[None]
        V(out, n, dim, dim).copy_(torch.cat([torch.cos(e), torch.sin(e)], 1))

    # ---- GroupNorm ----
    @staticmethod
    def _gn_parts(x, stats, n_s, rows, C, G, eps):
        xf = V(x, n_s * rows, C, C).float().view(n_s, rows, G, C // G)
        cnt = rows * (C // G)
        st = gn_decode(stats, n_s, G, cnt, 0)
        mean = st[..., 0] / cnt
        var = (st[..., 1] / cnt - mean * mean).clamp(min=0)
        rstd = torch.rsqrt(var + eps)
        return xf, mean[:, None, :, None], rstd[:, None, :, None], cnt

    def gn_stats(self, x, stats, n_s, rows, C, G, prezeroed=0):
        xf = V(x, n_s * rows, C, C).float().view(n_s, rows, G, C // G)
        if not prezeroed:
            gn_view(stats, n_s, G).zero_()
        # the replica split is the kernel's business; only the (fixed-point) sum is specified
        gn_encode_add(stats, n_s, G, rows * (C // G), 0, xf.sum((1, 3)), (xf * xf).sum((1, 3)))

    def gn_apply(self, x, stats, gamma, beta, y, n_s, rows, C, G, eps, silu_):
        xf, mean, rstd, _ = self._gn_parts(x, stats, n_s, rows, C, G, eps)
        z = ((xf - mean) * rstd).reshape(n_s, rows, C) * V1(gamma, C) + V1(beta, C)
        if silu_:
            z = silu(z)
        V(y, n_s * rows, C, C).copy_(z.reshape(n_s * rows, C).to(y.dtype))

    def _gn_dz(self, dy, x, stats, gamma, beta, n_s, rows, C, G, eps, silu_):
        xf, mean, rstd, cnt = self._gn_parts(x, stats, n_s, rows, C, G, eps)
        xhat = (xf - mean) * rstd
        dz = V(dy, n_s * rows, C, C).float().view(n_s, rows, C)
        if silu_:
            z = xhat.reshape(n_s, rows, C) * V1(gamma, C) + V1(beta, C)
            dz = dz * silu_grad(z)
        dzg = (dz * V1(gamma, C)).view(n_s, rows, G, C // G)
        return xhat, dzg, rstd, cnt

    def gn_bwd_stats(self, dy, x, stats, gamma, beta, bstats, n_s, rows, C, G, eps, silu_, prezeroed=0):
        xhat, dzg, _, _ = self._gn_dz(dy, x, stats, gamma, beta, n_s, rows, C, G, eps, silu_)
        if not prezeroed:
            gn_view(bstats, n_s, G).zero_()
        gn_encode_add(bstats, n_s, G, rows * (C // G), 1, dzg.sum((1, 3)), (dzg * xhat).sum((1, 3)))

    def gn_bwd_apply(self, dy, x, stats, bstats, gamma, beta, add, dx, n_s, rows, C, G, eps, silu_):
        xhat, dzg, rstd, cnt = self._gn_dz(dy, x, stats, gamma, beta, n_s, rows, C, G, eps, silu_)
        bs = gn_decode(bstats, n_s, G, cnt, 1)
        s1 = bs[..., 0][:, None, :, None]
        s2 = bs[..., 1][:, None, :, None]
        d = rstd * (dzg - (s1 + xhat * s2) / cnt)
        d = d.reshape(n_s * rows, C)
        if add is not None:
            d = d + V(add, n_s * rows, C, C).float()
        V(dx, n_s * rows, C, C).copy_(d.to(dx.dtype))

    # ---- LayerNorm ----
    def ln_fwd(self, x, gamma, beta, y, stats, rows, C, eps):
        xf = V(x, rows, C, C).float()
        mean = xf.mean(1, keepdim=True)
        var = xf.var(1, unbiased=False, keepdim=True)
        rstd = torch.rsqrt(var + eps)
        st = V(stats, rows, 2, 2)
        st[:, 0:1] = mean
        st[:, 1:2] = rstd
        V(y, rows, C, C).copy_((((xf - mean) * rstd) * V1(gamma, C) + V1(beta, C)).to(y.dtype))

    def ln_bwd(self, dy, x, stats, gamma, add, dx, dgamma, dbeta, rows, C, scratch=None, add2=None, add2_scale=1.0, defer_reduce=False):
        xf = V(x, rows, C, C).float()
        st = V(stats, rows, 2, 2)
        xhat = (xf - st[:, 0:1]) * st[:, 1:2]
        d = V(dy, rows, C, C).float()
        dg = d * V1(gamma, C)
        out = st[:, 1:2] * (dg - dg.mean(1, keepdim=True) - xhat * (dg * xhat).mean(1, keepdim=True))
        if add is not None:
            out = out + V(add, rows, C, C).float()
        if add2 is not None:
            out = out + add2_scale * V(add2, rows, C, C).float()
        V(dx, rows, C, C).copy_(out.to(dx.dtype))
        if defer_reduce:                       # partial rows [nblk][2C] for ln_param_reduce_batch: the totals in row 0, zeros below
            assert dgamma is not None and dbeta is not None and scratch is not None
            pt = V(scratch, K.ln_bwd_blocks(rows, C), 2 * C, 2 * C)
            pt.zero_()
            pt[0, :C] = (d * xhat).sum(0)
            pt[0, C:] = d.sum(0)
            return
        if dgamma is not None:
            V1(dgamma, C).add_((d * xhat).sum(0))
        if dbeta is not None:
            V1(dbeta, C).add_(d.sum(0))

    def ln_param_reduce_batch(self, jobs):
        for pt, dg, db, nblk, C in jobs:
            t = V(pt, nblk, 2 * C, 2 * C).sum(0)
            V1(dg, C).add_(t[:C])
            V1(db, C).add_(t[C:])

    # ---- spatial attention ----
    @staticmethod
    def _hv(t, nb, S, heads, ld):
        return torch.as_strided(t, (nb, heads, S, 64), (S * ld, 64, ld, 1), t.storage_offset())

    def attn_fwd(self, q, k, v, o, lse, nb, heads, S, ld, ld_o, scale):
        qf, kf, vf = (self._hv(t, nb, S, heads, ld).float() for t in (q, k, v))
        s = (qf @ kf.transpose(2, 3)) * scale
        l = torch.logsumexp(s, -1)
        p = torch.exp(s - l[..., None])
        V1(lse, nb * heads * S).view(nb, heads, S).copy_(l)
        self._hv(o, nb, S, heads, ld_o).copy_((p @ vf).to(o.dtype))

    def attn_bwd_prep(self, o, d_o, D, nb, heads, S, ld_o):
        V1(D, nb * heads * S).view(nb, heads, S).copy_(
            (self._hv(o, nb, S, heads, ld_o).float() * self._hv(d_o, nb, S, heads, ld_o).float()).sum(-1))

    def _attn_bwd_common(self, q, k, v, d_o, lse, D, nb, heads, S, ld, ld_o, scale):
        qf, kf, vf = (self._hv(t, nb, S, heads, ld).float() for t in (q, k, v))
        dof = self._hv(d_o, nb, S, heads, ld_o).float()
        l = V1(lse, nb * heads * S).view(nb, heads, S)
        Dv = V1(D, nb * heads * S).view(nb, heads, S)
        p = torch.exp((qf @ kf.transpose(2, 3)) * scale - l[..., None])
        dp = dof @ vf.transpose(2, 3)
        ds = p * (dp - Dv[..., None])
        return qf, kf, vf, dof, p, ds

    def attn_bwd_dkv(self, q, k, v, d_o, lse, D, dk, dv, nb, heads, S, ld, ld_o, ld_d, scale):
        qf, kf, vf, dof, p, ds = self._attn_bwd_common(q, k, v, d_o, lse, D, nb, heads, S, ld, ld_o, scale)
        self._hv(dv, nb, S, heads, ld_d).copy_((p.transpose(2, 3) @ dof).to(dv.dtype))
        self._hv(dk, nb, S, heads, ld_d).copy_(((ds.transpose(2, 3) @ qf) * scale).to(dk.dtype))

    def attn_bwd_dq(self, q, k, v, d_o, lse, D, dq, nb, heads, S, ld, ld_o, ld_d, scale):
        qf, kf, vf, dof, p, ds = self._attn_bwd_common(q, k, v, d_o, lse, D, nb, heads, S, ld, ld_o, scale)
        self._hv(dq, nb, S, heads, ld_d).copy_(((ds @ kf) * scale).to(dq.dtype))

    # ---- temporal attention ----
    @staticmethod
    def _tv(t, B, T, HW, heads, ld):
        # -> [B, HW, heads, T, 64]
        return torch.as_strided(t, (B, HW, heads, T, 64), (T * HW * ld, ld, 64, HW * ld, 1), t.storage_offset())

    def tattn_fwd(self, q, k, v, o, B, T, HW, heads, ld, ld_o, scale):
        qf, kf, vf = (self._tv(t, B, T, HW, heads, ld).float() for t in (q, k, v))
        p = torch.softmax((qf @ kf.transpose(-1, -2)) * scale, -1)
        self._tv(o, B, T, HW, heads, ld_o).copy_((p @ vf).to(o.dtype))

    def tattn_bwd(self, q, k, v, d_o, dq, dk, dv, B, T, HW, heads, ld, ld_o, ld_d, scale):
        qf, kf, vf = (self._tv(t, B, T, HW, heads, ld).float() for t in (q, k, v))
        dof = self._tv(d_o, B, T, HW, heads, ld_o).float()
        p = torch.softmax((qf @ kf.transpose(-1, -2)) * scale, -1)
        dp = dof @ vf.transpose(-1, -2)
        ds = p * (dp - (dp * p).sum(-1, keepdim=True))
        self._tv(dv, B, T, HW, heads, ld_d).copy_((p.transpose(-1, -2) @ dof).to(dv.dtype))
        self._tv(dq, B, T, HW, heads, ld_d).copy_(((ds @ kf) * scale).to(dq.dtype))
        self._tv(dk, B, T, HW, heads, ld_d).copy_(((ds.transpose(-1, -2) @ qf) * scale).to(dk.dtype))

    # ---- elementwise ----
    def geglu_fwd(self, pre, out, M, F):
        p = V(pre, M, 2 * F, 2 * F).float()
        V(out, M, F, F).copy_((p[:, :F] * gelu(p[:, F:])).to(out.dtype))

    def geglu_bwd(self, dout, pre, dpre, M, F):
        p = V(pre, M, 2 * F, 2 * F).float()
        d = V(dout, M, F, F).float()
        o = V(dpre, M, 2 * F, 2 * F)
        o[:, :F] = (d * gelu(p[:, F:])).to(o.dtype)
        o[:, F:] = (d * p[:, :F] * gelu_grad(p[:, F:])).to(o.dtype)

    def add(self, a, b, out, n):
        V1(out, n).copy_((V1(a, n).float() + V1(b, n).float()).to(out.dtype))

    def blend(self, a, b, mix, out, n):
        al = torch.sigmoid(V1(mix, 1))
        V1(out, n).copy_((al * V1(a, n).float() + (1 - al) * V1(b, n).float()).to(out.dtype))

    def blend_bwd(self, dy, mix, da, db, n):
        al = torch.sigmoid(V1(mix, 1))
        d = V1(dy, n).float()
        if da is not None:
            V1(da, n).copy_((al * d).to(da.dtype))
        V1(db, n).copy_(((1 - al) * d).to(db.dtype))

    @staticmethod
    def _gidx(rows, rpg, mod, dev):
        m = torch.arange(rows, device=dev)
        return (m % mod) if mod else (m // rpg)

    def add_rowvec(self, x, vec, out, rows, C, rv_ld, rpg, mod):
        gi = self._gidx(rows, rpg, mod, x.device)
        v = V(vec, int(gi.max()) + 1, C, rv_ld)
        V(out, rows, C, C).copy_((V(x, rows, C, C).float() + v[gi]).to(out.dtype))

    def colsum(self, x, out, rows, C, ldx, n_groups, rpg, mod, accumulate=0, scratch=None):
        gi = self._gidx(rows, rpg, mod, x.device)
        o = V(out, n_groups, C, C)
        if not accumulate:
            o.zero_()
        o.index_add_(0, gi, V(x, rows, C, ldx).float())

    def transpose(self, inp, ld_in, out, ld_out, rows, cols):
        o = V(out, cols, ld_out, ld_out)
        o.zero_()
        o[:, :rows] = V(inp, rows, cols, ld_in).t()

    def concat2(self, a, Ca, b, Cb, out, rows):
        o = V(out, rows, Ca + Cb, Ca + Cb)
        o[:, :Ca] = V(a, rows, Ca, Ca)
        o[:, Ca:] = V(b, rows, Cb, Cb)

    def split2(self, inp, a, Ca, b, Cb, rows):
        i = V(inp, rows, Ca + Cb, Ca + Cb)
        V(a, rows, Ca, Ca).copy_(i[:, :Ca])
        V(b, rows, Cb, Cb).copy_(i[:, Ca:])

    def sum2x2(self, inp, out, n_img, h, w, C):
        i = V(inp, n_img * 4 * h * w, C, C).float().view(n_img, h, 2, w, 2, C)
        V(out, n_img * h * w, C, C).copy_(i.sum((2, 4)).reshape(-1, C).to(out.dtype))

    def cast_from_f32(self, inp, out, n):
        V1(out, n).copy_(V1(inp, n).to(out.dtype))

    def cast_transpose_from_f32(self, inp, out, R, Cc):
        V(out, Cc, R, R).copy_(V(inp, R, Cc, Cc).t().to(out.dtype))

    def nchw_to_rows(self, inp, out, n_img, C, H, W, ld, mul=1.0):
        o = V(out, n_img * H * W, ld, ld)
        o.zero_()
        i = V1(inp, n_img * C * H * W).view(n_img, C, H * W)
        o[:, :C] = (i * mul).permute(0, 2, 1).reshape(-1, C).to(out.dtype)

    def rows_to_nchw(self, inp, out, n_img, C, H, W, ld):
        i = V(inp, n_img * H * W, C, ld).float().view(n_img, H * W, C)
        V1(out, n_img * C * H * W).view(n_img, C, H * W).copy_(i.permute(0, 2, 1))

    def zero(self, t):
        t.zero_()

    def tsa_fwd(self, x, gamma, beta, eps, wqkv, wo, bo, cvec, rv_ld, rv_rpg, rv_mod, n1, stats, qkv, o, h1, B, T, HW, C, heads, scale):
        """The four launches the fused op replaces, with their roundings (n, qkv, o are stored in the activation dtype and re-read)."""
        M = B * T * HW
        dt = x.dtype
        xf = V(x, M, C, C).float()
        mean = xf.mean(1, keepdim=True)
        var = ((xf - mean) ** 2).mean(1, keepdim=True)
        rstd = torch.rsqrt(var + eps)
        V(stats, M, 2, 2).copy_(torch.cat([mean, rstd], 1))
        n = ((xf - mean) * rstd * V1(gamma, C) + V1(beta, C)).to(dt)
        if n1 is not None:
            V(n1, M, C, C).copy_(n)
        q3 = (n.float() @ V(wqkv, 3 * C, C, C).float().t()).to(dt)
        V(qkv, M, 3 * C, 3 * C).copy_(q3)
        self.tattn_fwd(qkv, qkv[:, C:], qkv[:, 2 * C:], o, B, T, HW, heads, 3 * C, C, scale)
        v = V(o, M, C, C).float() @ V(wo, C, C, C).float().t() + V1(bo, C)[None]
        if cvec is not None:
            m = torch.arange(M, device=x.device)
            gi = (m % rv_mod) if rv_mod else (m // rv_rpg)
            v = v + V(cvec, int(gi.max()) + 1, C, rv_ld)[gi]
        V(h1, M, C, C).copy_((v + xf).to(dt))

    def patch_rows(self, inp, out, n_img, C, H, W, kh, kw, stride, pad, ho, wo, ldk, mul=1.0):
        import torch.nn.functional as F
        x = V1(inp, n_img * C * H * W).view(n_img, C, H, W) * mul
        cols = F.unfold(x, (kh, kw), padding=pad, stride=stride)                  # [n, C*kh*kw, ho*wo], k = (c*kh + dy)*kw + dx
        o = V(out, n_img * ho * wo, ldk, ldk)
        o.zero_()
        o[:, :C * kh * kw] = cols.permute(0, 2, 1).reshape(n_img * ho * wo, C * kh * kw).to(out.dtype)

    def softmax_rows(self, inp, out, rows, cols, cols_out, ld_in, ld_out, scale):
        x = V(inp, rows, cols, ld_in).float() * scale
        o = V(out, rows, cols_out, ld_out)
        o.zero_()
        o[:, :cols] = torch.softmax(x, -1).to(out.dtype)

    def blur_axis(self, inp, out, planes, H, W, taps, axis):
        import torch.nn.functional as F
        x = V1(inp, planes * H * W).view(planes, 1, H, W)
        half = (taps.numel() - 1) // 2
        if axis == 0:
            y = F.conv2d(F.pad(x, (half, half, 0, 0), mode="reflect"), taps.view(1, 1, 1, -1))
        else:
            y = F.conv2d(F.pad(x, (0, 0, half, half), mode="reflect"), taps.view(1, 1, -1, 1))
        V1(out, planes * H * W).copy_(y.reshape(-1))

    def bicubic_affine(self, inp, out, n_img, C, H, W, ho, wo, scale, shift):
        import torch.nn.functional as F
        x = V1(inp, n_img * C * H * W).view(n_img, C, H, W)
        y = F.interpolate(x, size=(ho, wo), mode="bicubic", align_corners=True)
        V1(out, n_img * C * ho * wo).copy_((y * scale.view(1, C, 1, 1) + shift.view(1, C, 1, 1)).reshape(-1))

    def attn_small_fwd(self, qkv, out, n_img, S, heads, d, dp, ld, ld_o, scale):
        x = V(qkv, n_img * S, 3 * heads * dp, ld).float().view(n_img, S, 3, heads, dp)
        q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))                  # [n, heads, S, dp]; channels >= d are padding
        p = torch.softmax((q[..., :d] @ k[..., :d].transpose(-1, -2)) * scale, -1)
        o = torch.zeros(n_img, heads, S, dp, device=qkv.device)
        o[..., :d] = p @ v[..., :d]
        V(out, n_img * S, heads * dp, ld_o).copy_(o.permute(0, 2, 1, 3).reshape(n_img * S, heads * dp).to(out.dtype))

    def act_rows(self, inp, out, n, act=0):
        x = V1(inp, n).float()
        V1(out, n).copy_((gelu(x) if act == 0 else x * torch.sigmoid(1.702 * x)).to(out.dtype))

    # ---- loss / optimizer ----
    def edm_loss(self, pred, ld, noisy, target, sigma, loss, dpred, B, T, C, HW, opt_state, scratch=None):
        p = V(pred, B * T * HW, C, ld).float().view(B, T, HW, C).permute(0, 1, 3, 2)
        nz = V1(noisy, B * T * C * HW).view(B, T, C, HW)
        tg = V1(target, B * T * C * HW).view(B, T, C, HW)
        s = V1(sigma, B)[:, None, None, None]
        c_out = -s / torch.sqrt(s * s + 1)
        c_skip = 1 / (s * s + 1)
        wgt = (1 + s * s) / (s * s)
        diff = c_out * p + c_skip * nz - tg
        norm = 1.0 / (B * T * C * HW)
        V1(loss, 1).add_((wgt * diff * diff).sum() * norm)
        d = opt_state[1] * 2 * wgt * diff * c_out * norm
        V(dpred, B * T * HW, C, dpred.shape[1]).copy_(d.permute(0, 1, 3, 2).reshape(-1, C).to(dpred.dtype))

    def check_finite(self, g, n, opt_state):
        if not torch.isfinite(V1(g, n)).all():
            opt_state[3] = 1.0

    def check_finite_spans(self, g, spans, n_spans, opt_state):
        for off, cnt in spans[:n_spans].view(-1, 2).tolist():
            if not torch.isfinite(V1(g, off + cnt)[off:off + cnt]).all():
                opt_state[3] = 1.0

    def optim_prep(self, st, beta1, beta2, growth, backoff, interval, dynamic):
        found = bool(st[3] > 0)
        inv = 1.0 / float(st[1])
        scale, tracker, step = float(st[1]), float(st[2]), float(st[0])
        if dynamic:
            if found:
                scale, tracker = scale * backoff, 0.0
            else:
                tracker += 1
                if tracker >= interval:
                    scale, tracker = scale * growth, 0.0
        st[8] = self._lr_lambda(st, step * max(1.0, float(st[15])))
        if not found:
            step += 1
        st[:8].copy_(torch.tensor([step, scale, tracker, 0.0, inv, 1 - beta1 ** step, 1 - beta2 ** step,
                                   1.0 if found else 0.0], dtype=torch.float32))

    @staticmethod
    def _lr_lambda(st, n):
        kind, warm, total, cycles, power, end_ratio = (float(x) for x in st[9:15])
        kind = int(kind)
        if kind == 0:
            return 1.0
        if kind == 6:
            nr = min(int(st[10]), K.SCHED_MAX_RULES)
            for i in range(nr):
                if n < float(st[16 + 2 * i]):
                    return float(st[17 + 2 * i])
            return float(st[16 + 2 * nr])
        if kind == 5:
            if n < warm:
                return n / max(1.0, warm)
            if n > total:
                return end_ratio
            return (1 - end_ratio) * (1 - (n - warm) / (total - warm)) ** power + end_ratio
        if n < warm:
            return n / max(1.0, warm)
        if kind == 1:
            return 1.0
        if kind == 2:
            return max(0.0, (total - n) / max(1.0, total - warm))
        prog = (n - warm) / max(1.0, total - warm)
        if kind == 3:
            return max(0.0, 0.5 * (1 + math.cos(math.pi * cycles * 2 * prog)))
        if prog >= 1:
            return 0.0
        return max(0.0, 0.5 * (1 + math.cos(math.pi * ((cycles * prog) % 1.0))))

    def ema_lerp(self, shadow, p, n, one_minus_decay):
        S, P = V1(shadow, n), V1(p, n)
        S.sub_(one_minus_decay * (S - P))

    def allreduce_grads(self, peer_bufs, rank, n, phase):
        """peer_bufs: the ranks' float tensors themselves (the emulation has no address space to map)"""
        world = len(peer_bufs)
        per = -(-(-(-n // world)) // 4) * 4
        if phase == 0:
            lo, hi = rank * per, min(n, rank * per + per)
            if hi > lo:
                acc = V1(peer_bufs[0], n)[lo:hi].clone()
                for q in range(1, world):
                    acc += V1(peer_bufs[q], n)[lo:hi]
                V1(peer_bufs[rank], n)[lo:hi] = acc
        elif phase == 1:
            for q in range(world):
                lo, hi = q * per, min(n, q * per + per)
                if q != rank and hi > lo:
                    V1(peer_bufs[rank], n)[lo:hi] = V1(peer_bufs[q], n)[lo:hi]

    def zero_spans(self, base, spans, n_spans):
        for off, cnt in spans[:n_spans].view(-1, 2).tolist():
            V1(base, off + cnt)[off:off + cnt].zero_()

    @staticmethod
    def _adamw_update(P, G, Mm, Vv, gm, lr, beta1, beta2, eps, wd, ss, bc2_sqrt, param_mode):
        """in-place update of (P, Mm, Vv) views.  param_mode 1 (SVDX_PARAMS_BF16_REFERENCE): torch.optim.AdamW's op sequence on bf16 TENSORS
        (the reference's LoRA recipe, train_svd_lora.py:666-674) -- done here with torch's own bf16 ops, results written back as floats."""
        if not param_mode:
            gg = G * gm
            P.mul_(1 - lr * wd)
            Mm.mul_(beta1).add_(gg, alpha=1 - beta1)
            Vv.mul_(beta2).addcmul_(gg, gg, value=1 - beta2)
            P.addcdiv_(Mm, Vv.sqrt() / bc2_sqrt + eps, value=-ss)
            return
        pb, mb, vb = P.to(torch.bfloat16), Mm.to(torch.bfloat16), Vv.to(torch.bfloat16)
        gb = (G * gm).to(torch.bfloat16)
        pb.mul_(1 - lr * wd)
        mb.lerp_(gb, 1 - beta1)
        vb.mul_(beta2).addcmul_(gb, gb, value=1 - beta2)
        denom = (vb.sqrt() / bc2_sqrt).add_(eps)
        pb.addcdiv_(mb, denom, value=-ss)
        P.copy_(pb.float()); Mm.copy_(mb.float()); Vv.copy_(vb.float())

    def adamw_tiled(self, p, g, m, v, tiles, n_tiles, lr, beta1, beta2, eps, wd, grad_mul, st, p_act, pt_act, param_mode=0):
        if float(st[7]) > 0:
            return
        lr = lr * float(st[8])
        gm, ss, bc2 = float(st[4]) * grad_mul, lr / float(st[5]), math.sqrt(float(st[6]))
        for off, ld, rows, cols, wt_off, ldwt in tiles[:n_tiles].view(-1, 6).tolist():
            def T2(t):
                return torch.as_strided(t, (rows, cols), (ld, 1), t.storage_offset() + off)
            P, G, Mm, Vv = T2(p), T2(g), T2(m), T2(v)
            self._adamw_update(P, G, Mm, Vv, gm, lr, beta1, beta2, eps, wd, ss, bc2, param_mode)
            if p_act is not None:
                T2(p_act).copy_(P.to(p_act.dtype))
            if wt_off >= 0:
                torch.as_strided(pt_act, (cols, rows), (ldwt, 1), pt_act.storage_offset() + wt_off).copy_(P.t().to(pt_act.dtype))

    def adamw(self, p, g, m, v, n, lr, beta1, beta2, eps, wd, grad_mul, st, p_act, param_mode=0):
        if float(st[7]) > 0:
            return
        lr = lr * float(st[8])
        P, G, Mm, Vv = V1(p, n), V1(g, n), V1(m, n), V1(v, n)
        self._adamw_update(P, G, Mm, Vv, float(st[4]) * grad_mul, lr, beta1, beta2, eps, wd, lr / float(st[5]), math.sqrt(float(st[6])), param_mode)
        if p_act is not None:
            V1(p_act, n).copy_(P.to(p_act.dtype))
