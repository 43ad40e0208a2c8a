"""Pins the torch emulation of the kernel interface (the fp32 reference every HIP kernel is compared with) to
PyTorch's own operators: implicit-GEMM gathers == conv2d/conv3d and their autograd data-grads, attention ==
SDPA + autograd, GroupNorm/LayerNorm/GEGLU forward+backward == autograd, AdamW == torch.optim.AdamW."""
import torch
import torch.nn.functional as F

import emul
from svd_xtend_amd import kernels as K

E = emul.EmuBackend()
torch.manual_seed(0)


def rows(x):   # NCHW -> [(n,y,x), C]
    n, c, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous()


def unrows(r, n, h, w):
    return r.reshape(n, h, w, -1).permute(0, 3, 1, 2)


def pack_fwd(W):       # [co,ci,3,3] -> [co, 9*ci] with k = tap*ci + c
    co, ci = W.shape[:2]
    return W.reshape(co, ci, -1).permute(0, 2, 1).reshape(co, -1).contiguous()


def pack_dgrad(W, flip):
    co, ci = W.shape[:2]
    w = W.reshape(co, ci, -1)
    if flip:
        w = w.flip(2)
    return w.permute(1, 2, 0).reshape(ci, -1).contiguous()


def test_conv3x3_fwd_and_dgrad_all_modes():
    n, ci, co, h, w = 2, 64, 64, 6, 10
    x = torch.randn(n, ci, h, w, requires_grad=True)
    W = torch.randn(co, ci, 3, 3) * 0.05
    for stride, ups in ((1, 0), (2, 0), (1, 1)):
        xin = F.interpolate(x, scale_factor=2.0, mode="nearest") if ups else x
        y = F.conv2d(xin, W, stride=stride, padding=1)
        ho, wo = y.shape[2:]
        M = n * ho * wo
        out = torch.zeros(M, co)
        g = K.Gather(K.GATHER_CONV3X3, n_img=n, hi=h * (2 if ups else 1), wi=w * (2 if ups else 1), ho=ho, wo=wo, cin=ci,
                     stride=stride, ups=ups, lda=ci)
        E.gemm(rows(x.detach()), pack_fwd(W), out, M, co, 9 * ci, ci, 9 * ci, co, gather=g)
        assert torch.allclose(unrows(out, n, ho, wo), y, atol=1e-4), (stride, ups)
        dy = torch.randn_like(y)
        (dx_ref,) = torch.autograd.grad(y, x, dy)
        if stride == 2:
            gd = K.Gather(K.GATHER_CONV3X3_DGRAD2, n_img=n, hi=ho, wi=wo, ho=h, wo=w, cin=co, lda=co)
            dx = torch.zeros(n * h * w, ci)
            E.gemm(rows(dy), pack_dgrad(W, False), dx, n * h * w, ci, 9 * co, co, 9 * co, ci, gather=gd)
        else:
            gd = K.Gather(K.GATHER_CONV3X3, n_img=n, hi=ho, wi=wo, ho=ho, wo=wo, cin=co, stride=1, lda=co)
            dxh = torch.zeros(n * ho * wo, ci)
            E.gemm(rows(dy), pack_dgrad(W, True), dxh, n * ho * wo, ci, 9 * co, co, 9 * co, ci, gather=gd)
            if ups:
                dx = torch.zeros(n * h * w, ci)
                E.sum2x2(dxh, dx, n, h, w, ci)
            else:
                dx = dxh
        assert torch.allclose(unrows(dx, n, h, w), dx_ref, atol=1e-4), (stride, ups)


def test_temporal_conv_fwd_and_dgrad():
    B, T, C, h, w = 2, 5, 64, 3, 4
    x = torch.randn(B, C, T, h, w, requires_grad=True)
    W = torch.randn(C, C, 3, 1, 1) * 0.05
    y = F.conv3d(x, W, padding=(1, 0, 0))
    xr = x.detach().permute(0, 2, 3, 4, 1).reshape(-1, C).contiguous()      # rows (b,t,y,x)
    M = B * T * h * w
    g = K.Gather(K.GATHER_TEMPORAL3, n_img=B, cin=C, t=T, hw=h * w, lda=C)
    out = torch.zeros(M, C)
    E.gemm(xr, pack_fwd(W), out, M, C, 3 * C, C, 3 * C, C, gather=g)
    assert torch.allclose(out.reshape(B, T, h, w, C).permute(0, 4, 1, 2, 3), y, atol=1e-4)
    dy = torch.randn_like(y)
    (dx_ref,) = torch.autograd.grad(y, x, dy)
    dyr = dy.permute(0, 2, 3, 4, 1).reshape(-1, C).contiguous()
    dx = torch.zeros(M, C)
    E.gemm(dyr, pack_dgrad(W, True), dx, M, C, 3 * C, C, 3 * C, C, gather=g)
    assert torch.allclose(dx.reshape(B, T, h, w, C).permute(0, 4, 1, 2, 3), dx_ref, atol=1e-4)


def test_groupnorm_fwd_bwd_2d_and_3d():
    for n_s, rws, C, silu in ((3, 20, 64, True), (2, 35, 96, False)):
        x = torch.randn(n_s, rws, C, requires_grad=True)
        gamma, beta = torch.randn(C), torch.randn(C)
        y_ref = F.group_norm(x.permute(0, 2, 1), 32, gamma, beta, 1e-5).permute(0, 2, 1)
        if silu:
            y_ref = F.silu(y_ref)
        xf = x.detach().reshape(-1, C).contiguous()
        st, y = torch.zeros(8, n_s, 32, 4), torch.zeros(n_s * rws, C)
        E.gn_stats(xf, st, n_s, rws, C, 32)
        E.gn_apply(xf, st, gamma, beta, y, n_s, rws, C, 32, 1e-5, silu)
        assert torch.allclose(y.view(n_s, rws, C), y_ref, atol=1e-4)
        dy = torch.randn(n_s, rws, C)
        (dx_ref,) = torch.autograd.grad(y_ref, x, dy)
        bs, dx = torch.zeros(8, n_s, 32, 4), torch.zeros(n_s * rws, C)
        add = torch.randn(n_s * rws, C)
        E.gn_bwd_stats(dy.reshape(-1, C), xf, st, gamma, beta, bs, n_s, rws, C, 32, 1e-5, silu)
        E.gn_bwd_apply(dy.reshape(-1, C), xf, st, bs, gamma, beta, add, dx, n_s, rws, C, 32, 1e-5, silu)
        assert torch.allclose(dx - add, dx_ref.reshape(-1, C), atol=2e-4)


def test_layernorm_fwd_bwd():
    R, C = 37, 128
    x = torch.randn(R, C, requires_grad=True)
    gamma, beta = torch.randn(C, requires_grad=True), torch.randn(C, requires_grad=True)
    y_ref = F.layer_norm(x, (C,), gamma, beta, 1e-5)
    y, st = torch.zeros(R, C), torch.zeros(R, 2)
    E.ln_fwd(x.detach(), gamma.detach(), beta.detach(), y, st, R, C, 1e-5)
    assert torch.allclose(y, y_ref, atol=1e-5)
    dy = torch.randn(R, C)
    dx_ref, dg_ref, db_ref = torch.autograd.grad(y_ref, (x, gamma, beta), dy)
    dx, dg, db = torch.zeros(R, C), torch.zeros(C), torch.zeros(C)
    E.ln_bwd(dy, x.detach(), st, gamma.detach(), None, dx, dg, db, R, C)
    assert torch.allclose(dx, dx_ref, atol=1e-4) and torch.allclose(dg, dg_ref, atol=1e-4) and torch.allclose(db, db_ref, atol=1e-4)


def test_spatial_attention_fwd_bwd_vs_sdpa():
    nb, heads, S = 2, 2, 40
    C = heads * 64
    qkv = torch.randn(nb * S, 3 * C, requires_grad=True)
    q, k, v = (qkv[:, i * C:(i + 1) * C].reshape(nb, S, heads, 64).transpose(1, 2) for i in range(3))
    o_ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(nb * S, C)
    d = qkv.detach()
    o, lse = torch.zeros(nb * S, C), torch.zeros(nb * heads * S)
    E.attn_fwd(d, d[:, C:], d[:, 2 * C:], o, lse, nb, heads, S, 3 * C, C, 0.125)
    assert torch.allclose(o, o_ref, atol=1e-5)
    d_o = torch.randn(nb * S, C)
    (dqkv_ref,) = torch.autograd.grad(o_ref, qkv, d_o)
    D = torch.zeros(nb * heads * S)
    E.attn_bwd_prep(o, d_o, D, nb, heads, S, C)
    dqkv = torch.zeros(nb * S, 3 * C)
    E.attn_bwd_dkv(d, d[:, C:], d[:, 2 * C:], d_o, lse, D, dqkv[:, C:], dqkv[:, 2 * C:], nb, heads, S, 3 * C, C, 3 * C, 0.125)
    E.attn_bwd_dq(d, d[:, C:], d[:, 2 * C:], d_o, lse, D, dqkv, nb, heads, S, 3 * C, C, 3 * C, 0.125)
    assert torch.allclose(dqkv, dqkv_ref, atol=1e-4)


def test_temporal_attention_vs_reference_permutation():
    """(B*T,HW,C) rows addressed in place == diffusers' permute to (B*HW,T,C) + SDPA."""
    B, T, HW, heads = 2, 5, 6, 2
    C = heads * 64
    qkv = torch.randn(B * T * HW, 3 * C, requires_grad=True)

    def perm(x):   # rows (b,t,p) -> [B*HW, heads, T, 64]
        return x.reshape(B, T, HW, heads, 64).permute(0, 2, 3, 1, 4).reshape(B * HW, heads, T, 64)
    q, k, v = (perm(qkv[:, i * C:(i + 1) * C]) for i in range(3))
    o_ref = F.scaled_dot_product_attention(q, k, v).reshape(B, HW, heads, T, 64).permute(0, 3, 1, 2, 4).reshape(B * T * HW, C)
    d = qkv.detach()
    o = torch.zeros(B * T * HW, C)
    E.tattn_fwd(d, d[:, C:], d[:, 2 * C:], o, B, T, HW, heads, 3 * C, C, 0.125)
    assert torch.allclose(o, o_ref, atol=1e-5)
    d_o = torch.randn_like(o)
    (ref,) = torch.autograd.grad(o_ref, qkv, d_o)
    dqkv = torch.zeros_like(d)
    E.tattn_bwd(d, d[:, C:], d[:, 2 * C:], d_o, dqkv, dqkv[:, C:], dqkv[:, 2 * C:], B, T, HW, heads, 3 * C, C, 3 * C, 0.125)
    assert torch.allclose(dqkv, ref, atol=1e-4)


def test_geglu_and_blend_grads():
    M, Fd = 9, 16
    pre = torch.randn(M, 2 * Fd, requires_grad=True)
    a, gte = pre.chunk(2, -1)
    y_ref = a * F.gelu(gte)
    y = torch.zeros(M, Fd)
    E.geglu_fwd(pre.detach(), y, M, Fd)
    assert torch.allclose(y, y_ref, atol=1e-6)
    dy = torch.randn(M, Fd)
    (ref,) = torch.autograd.grad(y_ref, pre, dy)
    dpre = torch.zeros(M, 2 * Fd)
    E.geglu_bwd(dy, pre.detach(), dpre, M, Fd)
    assert torch.allclose(dpre, ref, atol=1e-5)


def test_adamw_and_scaler_match_torch():
    n = 64
    p0, g = torch.randn(n), torch.randn(n)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p_ref], lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    p, m, v = p0.clone(), torch.zeros(n), torch.zeros(n)
    st = torch.tensor([0, 8.0, 0, 0, 1, 1, 1, 0, 1.0] + [0.0] * 7, dtype=torch.float32)
    for it in range(3):
        p_ref.grad = g.clone() * (it + 1)
        opt.step()
        gs = g * (it + 1) * float(st[1]) * 2          # grads arrive multiplied by loss scale and summed over 2 ranks
        E.check_finite(gs, n, st)
        E.optim_prep(st, 0.9, 0.999, 2.0, 0.5, 2, 1)
        E.adamw(p, gs, m, v, n, 1e-2, 0.9, 0.999, 1e-8, 1e-2, 0.5, st, None)
    assert torch.allclose(p, p_ref.detach(), atol=1e-6)
    assert float(st[0]) == 3 and float(st[1]) == 16.0      # grew once after 2 clean steps
    bad = g.clone()
    bad[3] = float("nan")
    E.check_finite(bad, n, st)
    E.optim_prep(st, 0.9, 0.999, 2.0, 0.5, 2, 1)
    before = p.clone()
    E.adamw(p, bad, m, v, n, 1e-2, 0.9, 0.999, 1e-8, 1e-2, 0.5, st, None)
    assert torch.equal(p, before) and float(st[1]) == 8.0 and float(st[0]) == 3    # skipped, scale halved
