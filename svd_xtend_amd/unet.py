"""MI355X-native `UNetSpatioTemporalConditionModel`: drop-in for the class the reference trains
(/root/reference/src/unet_spatio_temporal_condition.py:32-490, used at /root/reference/train_svd.py:1021).

Same constructor arguments, `forward(sample, timestep, encoder_hidden_states, added_time_ids, return_dict)`
contract and diffusers state-dict key names.  The arithmetic is NOT torch: every block below
holds its parameters under the diffusers names and runs an explicit, hand-written forward and backward made
of libsvdx (HIP, gfx950) kernel launches -- see svd_xtend_amd/ops.py and include/svdx.h.  Block internals
follow diffusers' unet_3d_blocks / transformer_temporal / attention / resnet modules as restated in
SURVEY.md 8(a) rows a3-a10.

Internal layout is rows = (b, t, y, x), channels last; temporal ops address frames by stride (no
(B*T,HW,C) <-> (B*HW,T,C) transposes); cross-attention has KV length 1 so it reduces to a per-clip row
vector `to_out(to_v(ctx))`.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import kernels as K
from .lora import LoraLinear, base_linear, inject
from .ops import _ones as ops_ones
from .ops import (ConvOp, GroupNormOp, LayerNormOp, LinearOp, LoraOp, Runtime, SmallLinearOp, SmallLoraOp, choose_geglu_variant, geglu_candidates, choose_split, flatten_trainables, tuned_call,
                  rup)

HEAD_DIM = 64


class Geom:
    """Shape of the activation rows at one resolution level."""

    def __init__(self, B: int, T: int, h: int, w: int):
        self.B, self.T, self.h, self.w = B, T, h, w
        self.N = B * T
        self.HW = h * w
        self.M = self.N * self.HW


# ==================================================================================================
# holders for diffusers sub-module names
# ==================================================================================================
class _TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim, out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.linear_2 = nn.Linear(dim, out_dim or dim)

    def build(self):
        self.l1 = SmallLinearOp(self.linear_1.weight, self.linear_1.bias)
        self.l2 = SmallLinearOp(self.linear_2.weight, self.linear_2.bias)

    def pack(self, rt):
        self.l1.pack(rt)
        self.l2.pack(rt)

    def fwd(self, rt, x, M, out=None, accumulate=False):
        h = self.l1.fwd(rt, x, M)
        return self.l2.fwd(rt, h, M, silu_in=True, out=out, accumulate=accumulate)


class _AlphaBlender(nn.Module):
    def __init__(self):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([0.5], dtype=torch.float32))


class _GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)


class _FeedForward(nn.Module):
    """net.0.proj (GEGLU) / net.2; fwd saves (pre, g) for the backward."""

    def __init__(self, dim, dim_out=None):
        super().__init__()
        self.dim, self.inner = dim, dim * 4
        self.net = nn.ModuleList([_GEGLU(dim, self.inner), nn.Dropout(0.0), nn.Linear(self.inner, dim_out or dim)])

    def build(self):
        self.p1 = LinearOp([self.net[0].proj.weight], [self.net[0].proj.bias])
        self.p2 = LinearOp([self.net[2].weight], [self.net[2].bias])

    def pack(self, rt):
        self.p1.pack(rt)
        self.p2.pack(rt)

    def refresh(self, rt):
        for op in (self.p1, self.p2):        # frozen projections (config 5: only adapters train) keep the copies packed at prepare()
            if op.trainable:
                op.refresh(rt)

    def _fusable(self, rt, M):
        F = self.inner
        return (rt.gemm_variant == 4 and rt.fuse_geglu and F % 128 == 0 and choose_split(rt, M, 2 * F, self.dim, 2 * F) == 1
                and choose_split(rt, M, F, self.p2.N, 2 * F) == 1)

    def fwd(self, rt, x, M, res):
        F = self.inner
        g = rt.empty(M, F)
        if self._fusable(rt, M):
            # GEGLU fused into the projection GEMM: one launch emits pre [M,2F] (saved for the backward) and h = a * gelu(gate)
            pre = rt.empty(M, 2 * F)
            tuned_call(rt, ("geglu_fwd", M, F, self.dim), lambda: geglu_candidates(M, 2 * F, self.dim),
                       lambda: choose_geglu_variant(M, 2 * F, self.dim),
                       lambda v: rt.k.gemm(x, self.p1.w, pre, M, 2 * F, self.dim, self.dim, self.dim, 2 * F, bias=self.p1.b, variant=v,
                                           epilogue=K.EPI_GEGLU_FWD, aux_out=g, aux_dim=F))
        else:
            pre = self.p1.fwd(rt, x, M)
            rt.k.geglu_fwd(pre, g, M, F)
        y = self.p2.fwd(rt, g, M, res=res)
        return y, pre, g

    def fwd_ln(self, rt, ln, x, M, res, need_n: bool):
        """LayerNorm `ln` + this feed-forward.  Returns (y, pre, g, n, stats).  (Round 2's one-launch LayerNorm + GEGLU band kernel was
        removed in round 3: with eight-wave tiles under the GEGLU epilogue the two launches are 0.15 ms/step faster.)"""
        n, st = ln.fwd(rt, x, M)
        y, pre, g = self.fwd(rt, n, M, res)
        return y, pre, g, n, st

    def bwd(self, rt, dy, x_saved, pre, g, M):
        """returns d(input of p1); accumulates weight grads when trainable."""
        k = rt.k
        F = self.inner
        dpre = rt.empty(M, 2 * F)
        if self._fusable(rt, M):
            # d(h) = dy W2 never reaches HBM: the data-grad GEMM's epilogue applies the GEGLU backward and writes d(pre)
            tuned_call(rt, ("geglu_bwd", M, F, self.p2.N), lambda: geglu_candidates(M, F, self.p2.N, fwd=False),
                       lambda: choose_geglu_variant(M, F, self.p2.N, fwd=False),
                       lambda v: k.gemm(dy, self.p2.wt, dpre, M, F, self.p2.N, self.p2.N, self.p2.N, 2 * F, variant=v,
                                        epilogue=K.EPI_GEGLU_BWD, aux_in=pre, aux_dim=F))
        else:
            dg = self.p2.bwd_dx(rt, dy, M)
            k.geglu_bwd(dg, pre, dpre, M, F)
            del dg
        if self.p2.trainable:
            self.p2.bwd_dw(rt, dy, g, M)
        dx = self.p1.bwd_dx(rt, dpre, M)
        if self.p1.trainable:
            self.p1.bwd_dw(rt, dpre, x_saved, M)
        return dx


class HipAttnProcessor:
    """What stands where diffusers keeps an attention-processor object (`Attention.processor`): attention here is always the
    library's own kernels (flash-style spatial attention, MFMA temporal attention, the KV-length-1 cross-attention short cut), so
    there is exactly one kind of processor and it carries no state.  It exists so that the processor plumbing of the reference class
    (/root/reference/src/unet_spatio_temporal_condition.py:248-321) can be called by code written against it."""

    def __repr__(self):
        return "HipAttnProcessor()"


class _Attention(nn.Module):
    def __init__(self, dim, heads, cross_dim=None):
        super().__init__()
        self.processor = HipAttnProcessor()
        inner = heads * HEAD_DIM
        kv = cross_dim if cross_dim is not None else dim
        self.heads, self.dim, self.cross = heads, dim, cross_dim is not None
        self.to_q = nn.Linear(dim, inner, bias=False)
        self.to_k = nn.Linear(kv, inner, bias=False)
        self.to_v = nn.Linear(kv, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, dim, bias=True), nn.Dropout(0.0)])

    def get_processor(self):
        return self.processor

    def set_processor(self, processor):
        if not isinstance(processor, HipAttnProcessor):
            raise ValueError(f"attention runs in libsvdx: the only processor is HipAttnProcessor, got {type(processor).__name__}")
        self.processor = processor

    def build(self):
        q, kk, v, o = (base_linear(m) for m in (self.to_q, self.to_k, self.to_v, self.to_out[0]))
        is_l = [isinstance(m, LoraLinear) for m in (self.to_q, self.to_k, self.to_v, self.to_out[0])]
        self.qkv_lora = self.o_lora = self.v_lora = None
        if self.cross:
            self.v = SmallLinearOp(v.weight, None)
            self.o = SmallLinearOp(o.weight, o.bias)
            # KV length 1: to_q / to_k (and their adapters) never influence the output -> zero gradient, nothing to run
            self.v_lora = SmallLoraOp(self.to_v) if is_l[2] else None
            self.o_lora = SmallLoraOp(self.to_out[0]) if is_l[3] else None
        else:
            self.qkv = LinearOp([q.weight, kk.weight, v.weight])
            self.o = LinearOp([o.weight], [o.bias])
            if any(is_l[:3]):
                if not all(is_l[:3]):
                    raise NotImplementedError("LoRA on a subset of to_q/to_k/to_v is not supported (the projection is fused)")
                self.qkv_lora = LoraOp([self.to_q, self.to_k, self.to_v])
            self.o_lora = LoraOp([self.to_out[0]]) if is_l[3] else None
        self.loras = [l for l in (self.qkv_lora, self.o_lora, self.v_lora) if l is not None]
        self.lora_trainable = any(l.trainable for l in self.loras)

    def pack(self, rt):
        if self.cross:
            self.v.pack(rt)
            self.o.pack(rt)
        else:
            self.qkv.pack(rt)
            self.o.pack(rt)
        for l in self.loras:
            l.pack(rt)

    def refresh(self, rt):
        """After an optimizer step: re-pack whatever is trainable (base projections in config 2-4, adapters in config 5)."""
        for op in ((self.v, self.o) if self.cross else (self.qkv, self.o)):
            if op.trainable:
                op.refresh(rt)
        for l in self.loras:
            if l.trainable:
                l.refresh(rt)

    # KV-length-1 cross attention: softmax over one key == 1, so attn2(x, ctx) = to_out(to_v(ctx)) for every
    # query row; to_q / to_k (and the LayerNorm feeding to_q) receive exactly zero gradient.
    _pre = None       # (out, saved) left by UNet..._cross_precompute for the next cross_vec of this sweep

    def cross_vec(self, rt, ctx, Bn):
        if self._pre is not None:                    # all blocks' to_v / to_out ran in two launches at the head of the sweep
            pre, self._pre = self._pre, None
            return pre
        v = self.v.fwd(rt, ctx, Bn)
        va = self.v_lora.fwd(rt, ctx, v, Bn) if self.v_lora is not None else None
        out = self.o.fwd(rt, v, Bn)
        oa = self.o_lora.fwd(rt, v, out, Bn) if self.o_lora is not None else None
        return out, (v, va, oa)

    @property
    def cross_trainable(self):
        return self.o.trainable or self.v.trainable or self.lora_trainable

    def cross_batchable(self) -> bool:
        """the table-driven launches carry no adapter scale: peft's lora_alpha == r (scale 1) is what the reference uses"""
        return all(l is None or l.s == 1.0 for l in (self.v_lora, self.o_lora))

    def cross_vec_bwd(self, rt, dvec, saved, ctx, Bn):
        v, va, oa = saved
        if rt.batch_small and self.cross_batchable():
            # nothing downstream of the sweep reads these gradients: queue the chain and run every block's share together when the
            # sweep -- or the gradient bucket -- ends (Runtime.flush_deferred).  Transposed linears in dependency stages (d(v) from
            # to_out and its adapter's B | the adapter's A into d(v) | d(v) through to_v's adapter B), then ALL outer products.
            def nn(op, dy, dx, stage, acc=False):
                rt.defer_nn((dy, op.w, None, dx, op.N, op.Kdim, op.Kdim, 0, int(acc)), Bn, stage)

            def outer(op, dy, x):
                if op.trainable:
                    rt.defer_outer((dy, x, op.weight.grad, op.N, op.Kdim, 1.0), Bn)
                    if op.bias is not None:
                        rt.defer_outer((dy, None, op.bias.grad, op.N, 1, 1.0), Bn)
            ol = self.o_lora if self.o_lora is not None and self.o_lora.trainable else None
            vl = self.v_lora if self.v_lora is not None and self.v_lora.trainable else None
            need_dv = self.v.trainable or vl is not None
            outer(self.o, dvec, v)
            dv = rt.f32(Bn, self.o.Kdim) if need_dv else None
            if need_dv:
                nn(self.o, dvec, dv, 0)
            if ol is not None:                       # y += B (A v): d(Av) = dvec B, dB += dvec^T (Av), dA += d(Av)^T v, d(v) += d(Av) A
                dxa = rt.f32(Bn, ol.b.Kdim)
                outer(ol.b, dvec, oa)
                nn(ol.b, dvec, dxa, 0)
                outer(ol.a, dxa, v)
                if need_dv:
                    nn(ol.a, dxa, dv, 1, acc=True)
            if need_dv:
                outer(self.v, dv, ctx)
            if vl is not None:
                dxa = rt.f32(Bn, vl.b.Kdim)
                outer(vl.b, dv, va)
                nn(vl.b, dv, dxa, 2)
                outer(vl.a, dxa, ctx)
            return
        need_dv = self.v.trainable or (self.v_lora is not None and self.v_lora.trainable)
        dv = self.o.bwd(rt, dvec, v, Bn, need_dx=need_dv)
        if self.o_lora is not None and self.o_lora.trainable:
            dv2 = self.o_lora.bwd(rt, dvec, v, oa, Bn, need_dx=need_dv)
            if need_dv:
                dv.add_(dv2)                        # [B, C] floats
        if self.v.trainable:
            self.v.bwd(rt, dv, ctx, Bn, need_dx=False)
        if self.v_lora is not None and self.v_lora.trainable:
            self.v_lora.bwd(rt, dv, ctx, va, Bn, need_dx=False)


def _proj_fwd(rt, lin, lora, x, M, **kw):
    """Projection with an optional LoRA adapter: the adapter term is the base GEMM's second operand pair.  Returns (y, xs)."""
    if lora is None:
        return lin.fwd(rt, x, M, **kw), None
    xs = lora.fwd_xs(rt, x, M)
    return lin.fwd(rt, x, M, dual=lora.fwd_dual(xs), **kw), xs


def _proj_bwd_dx(rt, lin, lora, dy, lddy, x, xs, M, need_dx=True, colsum_to=None):
    """Adapter gradients (when there is an adapter) and d(input) of the adapted projection, the adapter's share riding on the
    base data-grad GEMM.  colsum_to: see LoraOp.bwd."""
    dxa = lora.bwd(rt, dy, lddy, x, xs, M, colsum_to=colsum_to) if lora is not None else None
    if not need_dx:
        return None
    return lin.bwd_dx(rt, dy, M, dual=lora.bwd_dual(dxa) if lora is not None else None)


def _zeroed_vec(rt, n):
    """n zeroed floats (a slice of the sweep's pre-zeroed statistics arena when it has room)"""
    t, zeroed = rt.take_zeroed(n)
    if not zeroed:
        rt.k.zero(t)
    return t


def _dvec_from_lora(rt, attn1, attn2, g) -> bool:
    """config 5: d(cross-attention vector) = colsum(d(h)) rides on the dB GEMM of attn1.to_out's adapter (one clip per rank)"""
    return (rt.dvec_from_dw and attn2.cross_trainable and g.B == 1 and attn1.o_lora is not None and attn1.o_lora.trainable
            and attn1.o_lora.can_colsum())


# ==================================================================================================
# transformer blocks
# ==================================================================================================
class BasicTransformerBlock(nn.Module):
    """Spatial block (diffusers attention.BasicTransformerBlock): self-attn over HW, KV-1 cross-attn, GEGLU FF."""

    _chunk_size, _chunk_dim = None, 0

    def set_chunk_feed_forward(self, chunk_size, dim: int = 0):
        self._chunk_size, self._chunk_dim = chunk_size, dim           # recorded only (UNet...Model.enable_forward_chunking)

    def __init__(self, dim, heads, cross_dim):
        super().__init__()
        self.dim, self.heads = dim, heads
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = _Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = _Attention(dim, heads, cross_dim)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = _FeedForward(dim)

    def build(self):
        for m in (self.attn1, self.attn2, self.ff):
            m.build()
        self.ln1, self.ln3 = LayerNormOp(self.norm1), LayerNormOp(self.norm3)
        if any(p.requires_grad for n, p in self.named_parameters() if ".lora_" not in n):
            raise NotImplementedError("spatial transformer blocks are frozen on this path (train_svd.py:761-766); only LoRA "
                                      "adapters on their attention projections may train (train_svd_lora.py:659-674)")
        self.trainable = any(p.requires_grad for p in self.parameters())

    def refresh(self, rt):
        self.attn1.refresh(rt)
        self.attn2.refresh(rt)

    def pack(self, rt):
        for m in (self.attn1, self.attn2, self.ff):
            m.pack(rt)

    def fwd(self, rt: Runtime, h, g: Geom, ctx):
        k, C, M, S = rt.k, self.dim, g.M, g.HW
        n1, st1 = self.ln1.fwd(rt, h, M)
        qkv, xs_qkv = _proj_fwd(rt, self.attn1.qkv, self.attn1.qkv_lora, n1, M)
        if not (self.attn1.qkv_lora is not None and self.attn1.qkv_lora.trainable):
            n1 = None
        o = rt.empty(M, C)
        lse = rt.f32(g.N * self.heads * S)
        k.attn_fwd(qkv, qkv[:, C:], qkv[:, 2 * C:], o, lse, g.N, self.heads, S, 3 * C, C, HEAD_DIM ** -0.5)
        cvec, cv = self.attn2.cross_vec(rt, ctx, g.B)
        h2, xs_o = _proj_fwd(rt, self.attn1.o, self.attn1.o_lora, o, M, res=h, rowvec=cvec, rv_ld=C, rv_rpg=g.T * g.HW)
        h3, pre, _, _, st3 = self.ff.fwd_ln(rt, self.ln3, h2, M, res=h2, need_n=False)
        self.sv = (h, st1, qkv, o, lse, h2, st3, pre, n1, xs_qkv, xs_o, cv, ctx)
        return h3

    def bwd(self, rt: Runtime, dh3, g: Geom, need_dx: bool = True):
        k, C, M, S = rt.k, self.dim, g.M, g.HW
        h, st1, qkv, o, lse, h2, st3, pre, n1, xs_qkv, xs_o, cv, ctx = self.sv
        self.sv = None
        dn3 = self.ff.bwd(rt, dh3, None, pre, None, M)
        dh2 = self.ln3.bwd(rt, dn3, h2, st3, M, add=dh3)
        del dn3, pre, h2
        cs_lora = _dvec_from_lora(rt, self.attn1, self.attn2, g)
        if self.attn2.cross_trainable and not cs_lora:   # adapters on the cross-attention's to_v / to_out (per-clip vectors)
            dvec = rt.f32(g.B, C)
            k.colsum(dh2, dvec, M, C, C, g.B, g.T * g.HW, 0, scratch=rt.f32(K.colsum_slabs(M, g.T * g.HW, 0) * g.B * C))
            self.attn2.cross_vec_bwd(rt, dvec, cv, ctx, g.B)
        dvec = _zeroed_vec(rt, C) if cs_lora else None
        d_o = _proj_bwd_dx(rt, self.attn1.o, self.attn1.o_lora, dh2, C, o, xs_o, M, colsum_to=dvec)
        if cs_lora:
            self.attn2.cross_vec_bwd(rt, dvec.view(1, C), cv, ctx, 1)
        D = rt.f32(g.N * self.heads * S)
        k.attn_bwd_prep(o, d_o, D, g.N, self.heads, S, C)
        dqkv = rt.empty(M, 3 * C)
        q_, k_, v_ = qkv, qkv[:, C:], qkv[:, 2 * C:]
        scale = HEAD_DIM ** -0.5
        k.attn_bwd_dkv(q_, k_, v_, d_o, lse, D, dqkv[:, C:], dqkv[:, 2 * C:], g.N, self.heads, S, 3 * C, C, 3 * C, scale)
        k.attn_bwd_dq(q_, k_, v_, d_o, lse, D, dqkv, g.N, self.heads, S, 3 * C, C, 3 * C, scale)
        del d_o
        lora_q = self.attn1.qkv_lora is not None and self.attn1.qkv_lora.trainable
        if not need_dx and not lora_q:
            return None
        dn1 = _proj_bwd_dx(rt, self.attn1.qkv, self.attn1.qkv_lora, dqkv, 3 * C, n1, xs_qkv, M, need_dx=need_dx)
        del dqkv
        if not need_dx:
            return None
        return self.ln1.bwd(rt, dn1, h, st1, M, add=dh2)


class TemporalBasicTransformerBlock(nn.Module):
    """diffusers attention.TemporalBasicTransformerBlock -- the trainable set of train_svd.py:761-766.
    Rows stay in (b,t,p) order; only the frame-axis attention looks across rows (strided)."""

    _chunk_size, _chunk_dim = None, 0

    def set_chunk_feed_forward(self, chunk_size, dim: int = 0):
        self._chunk_size, self._chunk_dim = chunk_size, dim           # recorded only (UNet...Model.enable_forward_chunking)

    def __init__(self, dim, heads, cross_dim):
        super().__init__()
        self.dim, self.heads = dim, heads
        self.norm_in = nn.LayerNorm(dim)
        self.ff_in = _FeedForward(dim, dim_out=dim)
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = _Attention(dim, heads)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = _Attention(dim, heads, cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = _FeedForward(dim)

    def build(self):
        for m in (self.ff_in, self.attn1, self.attn2, self.ff):
            m.build()
        self.ln0, self.ln1, self.ln3 = LayerNormOp(self.norm_in), LayerNormOp(self.norm1), LayerNormOp(self.norm3)
        self.trainable = any(p.requires_grad for p in self.parameters())

    def pack(self, rt):
        for m in (self.ff_in, self.attn1, self.attn2, self.ff):
            m.pack(rt)

    def refresh(self, rt):
        for m in (self.ff_in, self.attn1, self.attn2, self.ff):
            m.refresh(rt)

    @staticmethod
    def _rv(g: Geom):
        # diffusers builds time_context in (HW, B) order but the block flattens (B, HW): row r of the block
        # sees the context of clip r % B (identical to the intended clip only at B == 1).  Reproduced as is.
        if g.B == 1:
            return dict(rv_rpg=g.M, rv_mod=0)
        assert g.HW % g.B == 0, "time_context ordering quirk needs HW % B == 0"
        return dict(rv_rpg=0, rv_mod=g.B)

    def _tsa_fused(self, rt: Runtime, g: Geom) -> bool:
        """norm1 -> attn1 -> residual as ONE launch (csrc/tsa.hip): needs T <= 16, C <= 320, no adapters on attn1, and pays when a
        band of pixels fills most of the kernel's 144-row tile (the 64x40 level of the benched shape: 10 pixels x 14 frames)."""
        if not (rt.fuse_tsa and hasattr(rt.k, "tsa_fwd") and self.attn1.qkv_lora is None and self.attn1.o_lora is None):
            return False
        if self.dim > K.TSA_MAX_C or self.dim % 64 or g.T > K.TSA_MAX_T:
            return False
        if g.M * 3 * self.dim * 2 >= 2 ** 31:                     # the kernel addresses q/k/v with 32-bit buffer offsets: larger batches take the unfused path
            return False
        return K.tsa_pixels_per_band(g.T, g.HW) * g.T >= 96

    def fwd(self, rt: Runtime, x, g: Geom, tctx):
        k, C, M = rt.k, self.dim, g.M
        keep_n = self.trainable and self.ff.p1.trainable            # the weight gradients of the feed-forwards read the normalised rows
        h, pre0, g0, n0, st0 = self.ff_in.fwd_ln(rt, self.ln0, x, M, res=x, need_n=keep_n)
        cvec, cv = self.attn2.cross_vec(rt, tctx, g.B)
        # the temporal self-attention OP (north_star's kernel): norm1 -> q/k/v -> attention over the frames -> out-projection + residual
        with rt.region("temporal_self_attention.fwd", M=M, C=C, T=g.T):
            if self._tsa_fused(rt, g):
                n1 = rt.empty(M, C) if self.trainable else None
                st1, qkv, o, h1 = rt.f32(M, 2), rt.empty(M, 3 * C), rt.empty(M, C), rt.empty(M, C)
                rv = self._rv(g)
                k.tsa_fwd(h, self.norm1.weight.data, self.norm1.bias.data, self.norm1.eps, self.attn1.qkv.w, self.attn1.o.w, self.attn1.o.b,
                          cvec, C, rv["rv_rpg"], rv["rv_mod"], n1, st1, qkv, o, h1, g.B, g.T, g.HW, C, self.heads, HEAD_DIM ** -0.5)
                xs_qkv = xs_o = None
            else:
                n1, st1 = self.ln1.fwd(rt, h, M)
                qkv, xs_qkv = _proj_fwd(rt, self.attn1.qkv, self.attn1.qkv_lora, n1, M)
                o = rt.empty(M, C)
                k.tattn_fwd(qkv, qkv[:, C:], qkv[:, 2 * C:], o, g.B, g.T, g.HW, self.heads, 3 * C, C, HEAD_DIM ** -0.5)
                h1, xs_o = _proj_fwd(rt, self.attn1.o, self.attn1.o_lora, o, M, res=h, rowvec=cvec, rv_ld=C, **self._rv(g))
        out, pre, gg, n3, st3 = self.ff.fwd_ln(rt, self.ln3, h1, M, res=h1, need_n=keep_n)
        if not self.trainable:
            n0 = g0 = n1 = n3 = gg = None
        elif not self.ff.p1.trainable:               # adapters only (config 5): the feed-forwards are frozen
            n0 = g0 = n3 = gg = None
        self.sv = (x, st0, n0, pre0, g0, h, st1, n1, qkv, o, cv, tctx, h1, st3, n3, pre, gg, xs_qkv, xs_o)
        return out

    def bwd(self, rt: Runtime, dout, g: Geom, need_dx: bool = True, add: Optional[torch.Tensor] = None, add_scale: float = 1.0):
        """returns d(input) + add_scale * add (the fan-in is folded into the last LayerNorm backward)."""
        k, C, M = rt.k, self.dim, g.M
        x, st0, n0, pre0, g0, h, st1, n1, qkv, o, cv, tctx, h1, st3, n3, pre, gg, xs_qkv, xs_o = self.sv
        self.sv = None
        dn3 = self.ff.bwd(rt, dout, n3, pre, gg, M)
        dh1 = self.ln3.bwd(rt, dn3, h1, st3, M, add=dout)
        del dn3, pre, gg, n3, h1
        # d(cross-attention vector) = the sum of d(h1) over the rows of each clip.  With one clip that is colsum(d(h1)), which the
        # attn1.to_out weight-gradient GEMM computes anyway (its bias gradient, four MFMAs per K-half on a fragment of ones): the
        # GEMM leaves it in `dvec`, and `dvec` is then added to the bias gradient by a skinny job -- no pass over d(h1) of its own
        dvec_from_dw = (rt.dvec_from_dw and self.attn2.cross_trainable and g.B == 1 and self.attn1.o.trainable
                        and self.attn1.o.b_grad is not None)
        cs_lora = not dvec_from_dw and _dvec_from_lora(rt, self.attn1, self.attn2, g)
        if self.attn2.cross_trainable and not (dvec_from_dw or cs_lora):
            rv = self._rv(g)
            dvec = rt.f32(g.B, C)
            k.colsum(dh1, dvec, M, C, C, g.B, rv["rv_rpg"], rv["rv_mod"],
                     scratch=rt.f32(K.colsum_slabs(M, rv["rv_rpg"], rv["rv_mod"]) * g.B * C))
            self.attn2.cross_vec_bwd(rt, dvec, cv, tctx, g.B)
        dvec = _zeroed_vec(rt, C) if (dvec_from_dw or cs_lora) else None
        # ... to the norm1 backward; the skinny cross-attention gradient launches in between belong to attn2
        with rt.region("temporal_self_attention.bwd", M=M, C=C, T=g.T):
            d_o = _proj_bwd_dx(rt, self.attn1.o, self.attn1.o_lora, dh1, C, o, xs_o, M, colsum_to=dvec if cs_lora else None)
            if cs_lora:
                self.attn2.cross_vec_bwd(rt, dvec.view(1, C), cv, tctx, 1)
            if dvec_from_dw:
                self.attn1.o.bwd_dw(rt, dh1, o, M, colsum_to=dvec)
                if rt.batch_small:
                    rt.defer_outer((dvec, None, self.attn1.o.b_grad, C, 1, 1.0), 1)
                else:
                    k.outer_acc(dvec.view(1, C), ops_ones(rt), self.attn1.o.b_grad.view(C, 1), 1, C, 1, 1.0)
                self.attn2.cross_vec_bwd(rt, dvec.view(1, C), cv, tctx, 1)
            elif self.attn1.o.trainable:
                self.attn1.o.bwd_dw(rt, dh1, o, M)
            dqkv = rt.empty(M, 3 * C)
            k.tattn_bwd(qkv, qkv[:, C:], qkv[:, 2 * C:], d_o, dqkv, dqkv[:, C:], dqkv[:, 2 * C:], g.B, g.T, g.HW,
                        self.heads, 3 * C, C, 3 * C, HEAD_DIM ** -0.5)
            del d_o, qkv, o
            dn1 = _proj_bwd_dx(rt, self.attn1.qkv, self.attn1.qkv_lora, dqkv, 3 * C, n1, xs_qkv, M)
            if self.attn1.qkv.trainable:
                self.attn1.qkv.bwd_dw(rt, dqkv, n1, M)
            del dqkv, n1
            dh = self.ln1.bwd(rt, dn1, h, st1, M, add=dh1)
        del dn1, dh1, h
        dn0 = self.ff_in.bwd(rt, dh, n0, pre0, g0, M)
        if not need_dx and not self.ln0.trainable:
            return None
        return self.ln0.bwd(rt, dn0, x, st0, M, add=dh, add2=add, add2_scale=add_scale)


class TransformerSpatioTemporalModel(nn.Module):
    """diffusers transformer_temporal.TransformerSpatioTemporalModel."""

    def __init__(self, heads, in_channels, num_layers, cross_dim):
        super().__init__()
        assert in_channels == heads * HEAD_DIM, "SVD uses head_dim 64 everywhere"
        self.C, self.heads = in_channels, heads
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, in_channels)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(in_channels, heads, cross_dim) for _ in range(num_layers)])
        self.temporal_transformer_blocks = nn.ModuleList(
            [TemporalBasicTransformerBlock(in_channels, heads, cross_dim) for _ in range(num_layers)])
        self.time_pos_embed = _TimestepEmbedding(in_channels, in_channels * 4, out_dim=in_channels)
        self.time_mixer = _AlphaBlender()
        self.proj_out = nn.Linear(in_channels, in_channels)
        self.need_dx = True
        self._pos_cache = None

    def build(self):
        self.gn = GroupNormOp(self.norm, silu=False)
        self.pin = LinearOp([self.proj_in.weight], [self.proj_in.bias])
        self.pout = LinearOp([self.proj_out.weight], [self.proj_out.bias])
        self.time_pos_embed.build()
        for b in list(self.transformer_blocks) + list(self.temporal_transformer_blocks):
            b.build()
        for n, p in self.named_parameters():
            if p.requires_grad and "temporal_transformer_blocks" not in n and ".lora_" not in n:
                raise NotImplementedError(f"{n}: only temporal_transformer_blocks.* or LoRA adapters may be trainable this round")

    def has_trainable(self):
        return any(p.requires_grad for p in self.parameters())

    def pack(self, rt):
        self._pos_cache = None
        self.alpha = float(torch.sigmoid(self.time_mixer.mix_factor.data.float()))      # frozen (build() checks)
        self.pin.pack(rt)
        self.pout.pack(rt)
        self.time_pos_embed.pack(rt)
        for b in list(self.transformer_blocks) + list(self.temporal_transformer_blocks):
            b.pack(rt)

    def refresh(self, rt):
        for b in list(self.transformer_blocks) + list(self.temporal_transformer_blocks):
            if b.trainable:
                b.refresh(rt)

    def first_norm(self, g: Geom):
        """(GroupNormOp, samples, rows per sample) of the norm that reads this module's input"""
        return self.gn, g.N, g.HW

    def fwd(self, rt: Runtime, x, g: Geom, ctx, pre=None, want_out=None):
        """x [M, C]; ctx float [B, cross_dim] (the CLIP embed of each clip; identical for all its frames, so
        the first-frame `time_context` of the temporal blocks is the same tensor).
        pre: GroupNorm statistics of x left by the GEMM that wrote it (GroupNormOp.fwd); want_out: `gn` operand for the GEMM that writes
        the result (GroupNormOp.want of the consumer).  Returns (out, statistics of out taken)."""
        k, C, M = rt.k, self.C, g.M
        xn, st = self.gn.fwd(rt, x, g.N, g.HW, pre=pre)
        h = self.pin.fwd(rt, xn, M)
        del xn
        # frame position embedding e[t] = time_pos_embed(Timesteps(C)(arange(T))), one row vector per frame.  It depends only on
        # (frozen) weights and the clip geometry, so it is computed once per (B, T) and reused by every later step.
        key = (g.B, g.T)
        if self._pos_cache is None or self._pos_cache[0] != key:
            tpos = torch.arange(g.T, dtype=torch.float32, device=rt.dev).repeat(g.B)
            fe = rt.f32(g.N, C)
            k.timestep_embed(tpos, fe, g.N, C)
            self._pos_cache = (key, self.time_pos_embed.fwd(rt, fe, g.N))
        e = self._pos_cache[1]
        for blk, tblk in zip(self.transformer_blocks, self.temporal_transformer_blocks):
            h = blk.fwd(rt, h, g, ctx)
            hm = rt.empty(M, C)
            k.add_rowvec(h, e, hm, M, C, C, g.HW, 0)
            hm = tblk.fwd(rt, hm, g, ctx)
            h2 = rt.empty(M, C)
            k.blend(h, hm, self.time_mixer.mix_factor.data, h2, M * C)
            h = h2
        if want_out is not None:
            out, took = self.pout.fwd(rt, h, M, res=x, gn=want_out)
        else:
            out, took = self.pout.fwd(rt, h, M, res=x), False
        self.sv = (x, st)
        return out, took

    def bwd(self, rt: Runtime, dout, g: Geom):
        k, C, M = rt.k, self.C, g.M
        x, st = self.sv
        self.sv = None
        dh = self.pout.bwd_dx(rt, dout, M)
        nl = len(self.transformer_blocks)
        for i in reversed(range(nl)):
            blk, tblk = self.transformer_blocks[i], self.temporal_transformer_blocks[i]
            last = (i == 0) and not self.need_dx
            stop = last and not blk.trainable        # nothing trainable at or before the spatial block: the sweep ends here
            dhm = rt.empty(M, C)
            k.blend_bwd(dh, self.time_mixer.mix_factor.data, None, dhm, M * C)       # (1-a) * dout for the temporal branch
            # d(h + e) = dhm_in ; the spatial output h feeds both the blend (a * dout) and the temporal block: that fan-in is
            # folded into the temporal block's last LayerNorm backward (add2 = dout, scale a)
            dh = tblk.bwd(rt, dhm, g, need_dx=not stop, add=None if stop else dh, add_scale=self.alpha)
            del dhm
            if stop:
                blk.sv = None
                return None
            dh = blk.bwd(rt, dh, g, need_dx=not last)    # with adapters (config 5) the spatial block has gradients of its own
            if last:
                return None
        dxn = self.pin.bwd_dx(rt, dh, M)
        return self.gn.bwd(rt, dxn, x, st, g.N, g.HW, add=dout)


# ==================================================================================================
# resnet blocks
# ==================================================================================================
class _ResnetHalf(nn.Module):
    """ResnetBlock2D (temporal=False) or TemporalResnetBlock (temporal=True) of diffusers resnet.py."""

    def __init__(self, cin, cout, temb_channels, eps, temporal):
        super().__init__()
        self.cin, self.cout, self.temporal = cin, cout, temporal
        self.norm1 = nn.GroupNorm(32, cin, eps=eps)
        if temporal:
            self.conv1 = nn.Conv3d(cin, cout, (3, 1, 1), padding=(1, 0, 0))
        else:
            self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, cout)
        self.norm2 = nn.GroupNorm(32, cout, eps=eps)
        if temporal:
            self.conv2 = nn.Conv3d(cout, cout, (3, 1, 1), padding=(1, 0, 0))
        else:
            self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = None
        if cin != cout:
            self.conv_shortcut = nn.Conv3d(cin, cout, 1) if temporal else nn.Conv2d(cin, cout, 1)
        self.temb_slice = (0, 0)    # (offset, ld) into the batched time_emb_proj output

    def build(self):
        kind = "t3" if self.temporal else "3x3"
        self.gn1, self.gn2 = GroupNormOp(self.norm1, True), GroupNormOp(self.norm2, True)
        self.c1 = ConvOp(self.conv1.weight, self.conv1.bias, kind)
        self.c2 = ConvOp(self.conv2.weight, self.conv2.bias, kind)
        self.sc = ConvOp(self.conv_shortcut.weight, self.conv_shortcut.bias, "1x1") if self.conv_shortcut is not None else None
        if self.time_emb_proj.weight.requires_grad:
            raise NotImplementedError("time_emb_proj is frozen on this path")

    def pack(self, rt, need_dx, out_scale: float = 1.0):
        self.c1.pack(rt, need_dx)
        self.c2.pack(rt, need_dx, out_scale=out_scale)
        if self.sc is not None:
            self.sc.pack(rt, need_dx)

    def _ns(self, g: Geom):
        # GroupNorm sample: a frame for the 2-D block, the whole clip (all frames) for the 3-D block
        return (g.B, g.T * g.HW) if self.temporal else (g.N, g.HW)

    def fwd(self, rt: Runtime, x, g: Geom, temb_all, pre=None, want_out=None):
        """pre / want_out: GroupNorm statistics from the producing GEMM's store loop -- of x (already taken) and of the result (asked
        of conv2) -- see TransformerSpatioTemporalModel.fwd.  Returns (out, statistics of out taken)."""
        n_s, rows = self._ns(g)
        a1, st1 = self.gn1.fwd(rt, x, n_s, rows, pre=pre)
        off, ld = self.temb_slice
        ni = g.B if self.temporal else g.N
        w2 = self.gn2.want(rt, n_s, rows)
        h1, _, _ = self.c1.fwd(rt, a1, ni, g.h, g.w, T=g.T, rowvec=temb_all[:, off:], rv_ld=ld, rv_rpg=g.T * g.HW, gn=w2)
        del a1
        a2, st2 = self.gn2.fwd(rt, h1, n_s, rows, pre=None if w2 is None else (w2[0], self.c1.took_gn))
        sc = x if self.sc is None else self.sc.fwd(rt, x, ni, g.h, g.w, T=g.T)[0]
        out, _, _ = self.c2.fwd(rt, a2, ni, g.h, g.w, T=g.T, res=sc, gn=want_out)
        self.sv = (x, st1, h1, st2)
        return out, (want_out is not None and self.c2.took_gn)

    def bwd(self, rt: Runtime, dout, g: Geom, add=None):
        """returns dx (+ add).  The identity/1x1 shortcut gradient is folded into the last GroupNorm backward."""
        x, st1, h1, st2 = self.sv
        self.sv = None
        n_s, rows = self._ns(g)
        ni = g.B if self.temporal else g.N
        da2 = self.c2.bwd_dx(rt, dout, ni, g.h, g.w, T=g.T)
        dh1 = self.gn2.bwd(rt, da2, h1, st2, n_s, rows)
        del da2, h1
        da1 = self.c1.bwd_dx(rt, dh1, ni, g.h, g.w, T=g.T)
        del dh1
        if self.sc is None:
            dsc = dout
        else:
            dsc = self.sc.bwd_dx(rt, dout, ni, g.h, g.w, T=g.T)
        if add is not None:
            tmp = rt.empty(g.M, self.cin)
            rt.k.add(dsc, add, tmp, g.M * self.cin)
            dsc = tmp
        return self.gn1.bwd(rt, da1, x, st1, n_s, rows, add=dsc)


class SpatioTemporalResBlock(nn.Module):
    def __init__(self, cin, cout, temb_channels, eps):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.spatial_res_block = _ResnetHalf(cin, cout, temb_channels, eps, temporal=False)
        self.temporal_res_block = _ResnetHalf(cout, cout, temb_channels, eps, temporal=True)
        self.time_mixer = _AlphaBlender()
        self.need_dx = True

    def build(self):
        self.spatial_res_block.build()
        self.temporal_res_block.build()
        if self.time_mixer.mix_factor.requires_grad:
            raise NotImplementedError("mix_factor is frozen on this path")

    def has_trainable(self):
        return False

    def pack(self, rt):
        # AlphaBlender folded away.  The temporal half has an identity shortcut, t = s + H(s) with H ending in conv2, so
        #   out = a*s + (1-a)*t = s + (1-a)*H(s),   a = sigmoid(mix_factor)   (frozen: train_svd.py:761-766)
        # i.e. the temporal half itself with conv2's weights and bias scaled by (1-a): no blend pass forward, and backward
        # d s = dout + H'^T((1-a) dout) is the temporal half's own backward on dout (scaled data-grad weights).
        assert self.temporal_res_block.conv_shortcut is None
        one_minus_a = 1.0 - float(torch.sigmoid(self.time_mixer.mix_factor.data.float()))
        self.spatial_res_block.pack(rt, self.need_dx)
        self.temporal_res_block.pack(rt, self.need_dx, out_scale=one_minus_a)

    def first_norm(self, g: Geom):
        return self.spatial_res_block.gn1, g.N, g.HW

    def fwd(self, rt: Runtime, x, g: Geom, temb_all, pre=None, want_out=None):
        t = self.temporal_res_block
        ws = t.gn1.want(rt, *t._ns(g))            # the temporal half's first norm reads the spatial half's output (clip-wide groups)
        s, took = self.spatial_res_block.fwd(rt, x, g, temb_all, pre=pre, want_out=ws)
        return t.fwd(rt, s, g, temb_all, pre=None if ws is None else (ws[0], took), want_out=want_out)

    def bwd(self, rt: Runtime, dout, g: Geom):
        if not self.need_dx:
            self.spatial_res_block.sv = self.temporal_res_block.sv = None
            return None
        ds_total = self.temporal_res_block.bwd(rt, dout, g)
        return self.spatial_res_block.bwd(rt, ds_total, g)


class Downsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=1)
        self.need_dx = True

    def build(self):
        self.op = ConvOp(self.conv.weight, self.conv.bias, "3x3", stride=2)

    def pack(self, rt):
        self.op.pack(rt, self.need_dx)


class Upsample2D(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)
        self.need_dx = True

    def build(self):
        self.op = ConvOp(self.conv.weight, self.conv.bias, "3x3", ups=True)

    def pack(self, rt):
        self.op.pack(rt, self.need_dx)


# ==================================================================================================
# block containers (diffusers unet_3d_blocks); they only hold names -- the UNet drives a flat step list
# ==================================================================================================
class _Block(nn.Module):
    def __init__(self, resnets, attentions=None, downsample=None, upsample=None):
        super().__init__()
        if attentions is not None:
            self.attentions = nn.ModuleList(attentions)
        self.resnets = nn.ModuleList(resnets)
        if downsample is not None:
            self.downsamplers = nn.ModuleList([downsample])
        if upsample is not None:
            self.upsamplers = nn.ModuleList([upsample])
        self.has_cross_attention = attentions is not None


class UNetSpatioTemporalConditionOutput(SimpleNamespace):
    pass


class FrozenConfig(dict):
    """diffusers' FrozenDict in miniature: `config.addition_time_embed_dim` (train_svd.py:887) and
    `model.register_to_config(**other.config)` (train_svd.py:723) both work."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


CONFIG_NAME = "config.json"
WEIGHTS_NAME = "diffusion_pytorch_model{variant}.safetensors"


class _UNetFn(torch.autograd.Function):
    """Thin autograd boundary so `loss.backward()` in a host script reaches the hand-written backward."""

    @staticmethod
    def forward(ctx, model, sample, timestep, ehs, added_time_ids, _anchor):
        ctx.model = model
        return model._forward_impl(sample, timestep, ehs, added_time_ids)

    @staticmethod
    def backward(ctx, d_out):
        ctx.model._backward_impl(d_out.contiguous())
        return None, None, None, None, None, None


class UNetSpatioTemporalConditionModel(nn.Module):
    """Constructor signature of src/unet_spatio_temporal_condition.py:71-96."""

    _supports_gradient_checkpointing = True

    def __init__(self, sample_size=None, in_channels: int = 8, out_channels: int = 4,
                 down_block_types=("CrossAttnDownBlockSpatioTemporal",) * 3 + ("DownBlockSpatioTemporal",),
                 up_block_types=("UpBlockSpatioTemporal",) + ("CrossAttnUpBlockSpatioTemporal",) * 3,
                 block_out_channels=(320, 640, 1280, 1280), addition_time_embed_dim: int = 256,
                 projection_class_embeddings_input_dim: int = 768, layers_per_block=2, cross_attention_dim=1024,
                 transformer_layers_per_block=1, num_attention_heads=(5, 10, 20, 20), num_frames: int = 25):
        super().__init__()
        n = len(down_block_types)
        if len(up_block_types) != n or len(block_out_channels) != n:
            raise ValueError("down_block_types, up_block_types and block_out_channels must have the same length")
        self.config = FrozenConfig(
            sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
            block_out_channels=tuple(block_out_channels), addition_time_embed_dim=addition_time_embed_dim,
            projection_class_embeddings_input_dim=projection_class_embeddings_input_dim,
            layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim,
            transformer_layers_per_block=transformer_layers_per_block, num_attention_heads=num_attention_heads,
            num_frames=num_frames)
        heads = (num_attention_heads,) * n if isinstance(num_attention_heads, int) else tuple(num_attention_heads)
        cross = (cross_attention_dim,) * n if isinstance(cross_attention_dim, int) else tuple(cross_attention_dim)
        layers = [layers_per_block] * n if isinstance(layers_per_block, int) else list(layers_per_block)
        tl = [transformer_layers_per_block] * n if isinstance(transformer_layers_per_block, int) \
            else list(transformer_layers_per_block)
        if len(set(cross)) != 1:
            raise NotImplementedError("a single cross_attention_dim is assumed")
        ch = block_out_channels
        temb = ch[0] * 4
        self.in_channels, self.out_channels = in_channels, out_channels
        self.cin_pad = rup(in_channels, 64)
        self.cout_pad = rup(out_channels, 64)

        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.time_embedding = _TimestepEmbedding(ch[0], temb)
        self.add_embedding = _TimestepEmbedding(projection_class_embeddings_input_dim, temb)

        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()      # registered before mid_block, as in the reference (:146-147): same parameter order
        out_c = ch[0]
        for i, t in enumerate(down_block_types):
            in_c, out_c = out_c, ch[i]
            final = i == n - 1
            if t == "CrossAttnDownBlockSpatioTemporal":
                res = [SpatioTemporalResBlock(in_c if j == 0 else out_c, out_c, temb, 1e-6) for j in range(layers[i])]
                att = [TransformerSpatioTemporalModel(heads[i], out_c, tl[i], cross[i]) for _ in range(layers[i])]
            elif t == "DownBlockSpatioTemporal":
                res = [SpatioTemporalResBlock(in_c if j == 0 else out_c, out_c, temb, 1e-5) for j in range(layers[i])]
                att = None
            else:
                raise ValueError(f"{t} does not exist.")
            self.down_blocks.append(_Block(res, att, downsample=None if final else Downsample2D(out_c)))

        mid_c = ch[-1]
        self.mid_block = _Block([SpatioTemporalResBlock(mid_c, mid_c, temb, 1e-5) for _ in range(2)],
                                [TransformerSpatioTemporalModel(heads[-1], mid_c, tl[-1], cross[-1])])

        rch, rheads, rlayers, rcross, rtl = (list(reversed(v)) for v in (ch, heads, layers, cross, tl))
        out_c = rch[0]
        for i, t in enumerate(up_block_types):
            final = i == n - 1
            prev_c, out_c = out_c, rch[i]
            in_c = rch[min(i + 1, n - 1)]
            nl = rlayers[i] + 1
            res = []
            for j in range(nl):
                skip = in_c if j == nl - 1 else out_c
                rin = prev_c if j == 0 else out_c
                res.append(SpatioTemporalResBlock(rin + skip, out_c, temb, 1e-6))
            if t == "CrossAttnUpBlockSpatioTemporal":
                att = [TransformerSpatioTemporalModel(rheads[i], out_c, rtl[i], rcross[i]) for _ in range(nl)]
            elif t == "UpBlockSpatioTemporal":
                att = None
            else:
                raise ValueError(f"{t} does not exist.")
            self.up_blocks.append(_Block(res, att, upsample=None if final else Upsample2D(out_c)))

        self.conv_norm_out = nn.GroupNorm(num_channels=ch[0], num_groups=32, eps=1e-5)
        self.conv_out = nn.Conv2d(ch[0], out_channels, 3, padding=1)

        self.add_embedding_in = projection_class_embeddings_input_dim
        self.rt: Optional[Runtime] = None
        self._anchor = None
        self.gradient_checkpointing = False

    # ---- reference-script surface ------------------------------------------------------
    def register_to_config(self, **kwargs) -> None:                     # train_svd.py:723
        self.config = FrozenConfig({**self.config, **kwargs})

    @classmethod
    def load_config(cls, path, return_unused_kwargs: bool = False, subfolder: Optional[str] = None, **unused):
        """`config.json` of a diffusers model folder, split into constructor arguments and everything else (EMAModel keeps its
        settings there, train_svd.py:699-700 / :710-711)."""
        import json
        import os
        folder = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(folder, CONFIG_NAME)) as f:
            raw = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        names = cls.__init__.__code__.co_varnames
        cfg = {k: v for k, v in raw.items() if k in names}
        rest = {k: v for k, v in raw.items() if k not in names}
        return (cfg, rest) if return_unused_kwargs else cfg

    @classmethod
    def from_config(cls, config, **unused):
        """A randomly initialised model from a config mapping (EMAModel.save_pretrained rebuilds the module tree this way)."""
        names = cls.__init__.__code__.co_varnames
        return cls(**{k: v for k, v in dict(config).items() if k in names and not k.startswith("_")})

    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = None, variant: Optional[str] = None, torch_dtype=None,
                        **unused):
        """diffusers folder layout (train_svd.py:651-656, 721): `<path>/<subfolder>/config.json` +
        `diffusion_pytorch_model[.<variant>].safetensors`, diffusers key names, strict load.  Local folders only."""
        import json
        import os

        from safetensors.torch import load_file
        folder = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(folder, CONFIG_NAME)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        ctor = {k: v for k, v in cfg.items() if k in cls.__init__.__code__.co_varnames}
        model = cls(**ctor)
        wpath = os.path.join(folder, WEIGHTS_NAME.format(variant=f".{variant}" if variant else ""))
        if not os.path.exists(wpath) and variant:                         # diffusers falls back to the un-suffixed file
            wpath = os.path.join(folder, WEIGHTS_NAME.format(variant=""))
        sd = load_file(wpath)
        model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)   # float masters; kernels use packed copies
        if torch_dtype is not None and torch_dtype != torch.float32:
            model._requested_dtype = torch_dtype
        return model

    def save_pretrained(self, folder, variant: Optional[str] = None, **unused) -> None:    # train_svd.py:703, 1088-1090
        import json
        import os

        from safetensors.torch import save_file
        os.makedirs(folder, exist_ok=True)
        cfg = {"_class_name": "UNetSpatioTemporalConditionModel", "_svd_xtend_amd": True}
        cfg.update({k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()})
        with open(os.path.join(folder, CONFIG_NAME), "w") as f:
            json.dump(cfg, f, indent=2)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if variant == "fp16":
            sd = {k: v.half() for k, v in sd.items()}
        save_file(sd, os.path.join(folder, WEIGHTS_NAME.format(variant=f".{variant}" if variant else "")))

    def add_adapter(self, adapter_config, adapter_name: str = "default") -> int:
        """train_svd_lora.py:671 (`unet.add_adapter(LoraConfig(...))`, diffusers -> peft): wrap every targeted attention
        projection (lora.py) and freeze its base layer; the adapters are the new trainables.  Call before `prepare()` /
        `Trainer(...)`.  Returns the number of wrapped layers (256 for the SVD UNet with the reference's target list)."""
        if adapter_name != "default":
            raise NotImplementedError("a single adapter named 'default' is supported")
        if self.rt is not None:
            raise RuntimeError("add_adapter must be called before prepare()")
        return inject(self, adapter_config)

    def enable_gradient_checkpointing(self):      # train_svd.py:732 -- 288 GB HBM: not needed, accepted as no-op
        self.gradient_checkpointing = False

    def enable_xformers_memory_efficient_attention(self, *a, **k):   # train_svd.py:690 -- own attention kernels
        return None

    # ---- attention-processor plumbing and feed-forward chunking of the reference class (src/...:248-321, 328-355) ----------------
    # Never called by the training scripts; kept so that code written against the reference class runs.  Same names, argument
    # meaning and error behaviour; the only processor that exists here is HipAttnProcessor.
    @property
    def attn_processors(self):
        """{"<module path>.processor": processor} of every attention layer (64 at the SVD topology), keyed like the reference's."""
        return {f"{name}.processor": m.get_processor() for name, m in self.named_modules() if hasattr(m, "get_processor")}

    def set_attn_processor(self, processor):
        """One processor for every attention layer, or a dict keyed like `attn_processors` (its length must equal the layer count)."""
        layers = [(name, m) for name, m in self.named_modules() if hasattr(m, "set_processor")]
        if isinstance(processor, dict):
            if len(processor) != len(layers):
                raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not match the"
                                 f" number of attention layers: {len(layers)}. Please make sure to pass {len(layers)} processor classes.")
            for name, m in layers:
                m.set_processor(processor.pop(f"{name}.processor"))
        else:
            for _, m in layers:
                m.set_processor(processor)

    def set_default_attn_processor(self):
        self.set_attn_processor(HipAttnProcessor())

    def enable_forward_chunking(self, chunk_size=None, dim: int = 0) -> None:
        """The reference chunks every feed-forward over `dim` (0 = batch, 1 = sequence) to bound its peak memory.  The feed-forwards
        here are row-wise GEMM chains on buffers sized for 288 GB of HBM: results do not depend on chunking, so the request is validated
        and recorded on the transformer blocks (`_chunk_size`, `_chunk_dim`, the attributes diffusers' blocks keep), nothing else."""
        if dim not in (0, 1):
            raise ValueError(f"Make sure to set `dim` to either 0 or 1, not {dim}")
        chunk_size = chunk_size or 1
        for m in self.modules():
            if hasattr(m, "set_chunk_feed_forward"):
                m.set_chunk_feed_forward(chunk_size=chunk_size, dim=dim)

    # ---- build / pack ---------------------------------------------------------------------------------
    def _steps(self):
        """Flat forward schedule: (kind, module, level_delta)."""
        steps = []
        for blk in self.down_blocks:
            for j, r in enumerate(blk.resnets):
                steps.append(("res", r))
                if blk.has_cross_attention:
                    steps.append(("attn", blk.attentions[j]))
                steps.append(("push", None))
            if hasattr(blk, "downsamplers"):
                steps.append(("down", blk.downsamplers[0]))
                steps.append(("push", None))
        steps.append(("res", self.mid_block.resnets[0]))
        steps.append(("attn", self.mid_block.attentions[0]))
        steps.append(("res", self.mid_block.resnets[1]))
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                steps.append(("pop_cat", r))
                steps.append(("res", r))
                if blk.has_cross_attention:
                    steps.append(("attn", blk.attentions[j]))
            if hasattr(blk, "upsamplers"):
                steps.append(("up", blk.upsamplers[0]))
        return steps

    def to(self, *args, **kwargs):
        """`unet.to(device, dtype=weight_dtype)` (train_svd_lora.py:669, train_svd.py:739): the float masters stay fp32 (the
        kernels run on their packed 16-bit copies); a half / bfloat16 request only selects the activation dtype."""
        device, dtype, non_blocking, _ = torch._C._nn._parse_to(*args, **kwargs)
        if dtype is not None and dtype.is_floating_point and dtype != torch.float32:
            self._requested_dtype = dtype
            return super().to(device=device, non_blocking=non_blocking) if device is not None else self
        return super().to(*args, **kwargs)

    def half(self):
        return self.to(dtype=torch.float16)

    def bfloat16(self):
        return self.to(dtype=torch.bfloat16)

    def prepare(self, dtype: Optional[torch.dtype] = None) -> "UNetSpatioTemporalConditionModel":
        """Build operator objects and pack weights into kernel layouts.  Call after weights are loaded, after
        `requires_grad` flags are final and (for training) after the Trainer has installed flat grads.  dtype: activation
        dtype (default: what `.to(dtype=...)` / `from_pretrained(torch_dtype=...)` asked for, else float16)."""
        dtype = dtype or getattr(self, "_requested_dtype", None) or torch.float16
        dev = next(self.parameters()).device
        if any(p.requires_grad and p.grad is None for p in self.parameters()):
            # stand-alone use (host script keeps its own optimizer): give the trainables flat, adjacent storage
            self._flat = flatten_trainables(self)
        self.rt = rt = Runtime(dtype, dev)
        if getattr(self, "_flat", None) is not None and self._flat[2] > 0:
            rt.p_flat = self._flat[3]
            rt.w16_flat = torch.empty(rt.p_flat.numel(), dtype=dtype, device=dev)
            rt.k.cast_from_f32(rt.p_flat, rt.w16_flat, rt.p_flat.numel())
            rt.wt16_flat = torch.empty(rt.p_flat.numel(), dtype=dtype, device=dev)     # transposed twins (ops.LinearOp.pack)
        self.steps = self._steps()
        self.grads_ready_cb = None       # callable(module) invoked by backward_rows after each transformer block (gradient overlap)
        self.grads_ready_flush = True    # the callback consumes the block's gradients: run the queued skinny gradient launches first
        # which modules need an input gradient: only those executed after the first trainable parameter
        seen = False
        skip_flags = [seen]                       # conv_in output
        for kind, m in self.steps:
            if kind in ("res", "attn", "down", "up"):
                m.need_dx = seen
                if kind == "attn" and m.has_trainable():
                    seen = True
            elif kind == "push":
                skip_flags.append(seen)
        self._skip_needs_grad = skip_flags
        self._any_trainable = seen
        for p_name, p in self.named_parameters():
            if p.requires_grad and "temporal_transformer_blocks" not in p_name and ".lora_" not in p_name:
                raise NotImplementedError(f"{p_name}: only temporal_transformer_blocks.* or LoRA adapters may be trainable this round")

        self.time_embedding.build()
        self.add_embedding.build()
        self.time_embedding.pack(rt)
        self.add_embedding.pack(rt)
        self.cin_op = ConvOp(self.conv_in.weight, self.conv_in.bias, "3x3", cin_pad=self.cin_pad)
        self.cin_op.pack(rt, need_dx=False)
        self.gn_out = GroupNormOp(self.conv_norm_out, silu=True)
        self.cout_op = ConvOp(self.conv_out.weight, self.conv_out.bias, "3x3", cout_pad=self.cout_pad)
        self.cout_op.pack(rt, need_dx=True)
        halves = []
        for kind, m in self.steps:
            if kind in ("res", "attn", "down", "up"):
                m.build()
                m.pack(rt)
            if kind == "res":
                halves += [m.spatial_res_block, m.temporal_res_block]
        # one batched time_emb_proj for all resnet halves: [sum(Cout), temb] (the input silu(emb) is shared)
        off = 0
        for hf in halves:
            hf.temb_slice = (off, None)
            off += hf.cout
        for hf in halves:
            hf.temb_slice = (hf.temb_slice[0], off)
        self._temb_total = off
        wcat = torch.cat([hf.time_emb_proj.weight.data for hf in halves], 0).contiguous()
        self._temb_b = torch.cat([hf.time_emb_proj.bias.data for hf in halves], 0).contiguous()
        self._temb_w = rt.empty(off, wcat.shape[1])
        rt.k.cast_from_f32(wcat, self._temb_w, wcat.numel())
        self._anchor = torch.zeros((), device=dev, requires_grad=True)
        return self

    def refresh_trainable(self, masters_changed_on_host: bool = True) -> None:
        """Re-pack the low-precision copies of trainable weights after an optimizer step.  `masters_changed_on_host`
        (torch optimizer route): one cast of the whole flat buffer; the Trainer's AdamW kernel writes it directly."""
        if masters_changed_on_host and self.rt.p_flat is not None:
            self.rt.k.cast_from_f32(self.rt.p_flat, self.rt.w16_flat, self.rt.p_flat.numel())
        for kind, m in self.steps:
            if kind == "attn":
                m.refresh(self.rt)

    # ---- forward / backward ---------------------------------------------------------------------------
    def forward(self, sample, timestep, encoder_hidden_states, added_time_ids, return_dict: bool = True):
        if self.rt is None:
            self.prepare()
        if torch.is_grad_enabled() and self._any_trainable:
            out = _UNetFn.apply(self, sample, timestep, encoder_hidden_states, added_time_ids, self._anchor)
        else:
            out = self._forward_impl(sample, timestep, encoder_hidden_states, added_time_ids)
            self._drop_saved()
        if not return_dict:
            return (out,)
        return UNetSpatioTemporalConditionOutput(sample=out)

    def _drop_saved(self):
        for kind, m in self.steps:
            if kind == "res":
                m.spatial_res_block.sv = m.temporal_res_block.sv = None
            elif kind == "attn":
                m.sv = None
                for b in list(m.transformer_blocks) + list(m.temporal_transformer_blocks):
                    b.sv = None
        self._fwd_state = None
        self._clear_cross_pre()

    def _clear_cross_pre(self):
        """Drop cross-attention vectors a previous sweep precomputed and never consumed (a forward that raised, a knob that
        flipped between sweeps): the next cross_vec must not read results of an older context."""
        for kind, m in self.steps:
            if kind == "attn":
                for b in list(m.transformer_blocks) + list(m.temporal_transformer_blocks):
                    b.attn2._pre = None

    def _forward_impl(self, sample, timestep, ehs, added_time_ids):
        out_rows = self.forward_rows(sample, timestep, ehs, added_time_ids)
        B, T = sample.shape[:2]
        g = self._fwd_state["g0"]
        out = torch.empty(B, T, self.out_channels, g.h, g.w, dtype=torch.float32, device=self.rt.dev)
        self.rt.k.rows_to_nchw(out_rows, out, g.N, self.out_channels, g.h, g.w, self.out_channels)
        return out

    def forward_rows(self, sample, timestep, ehs, added_time_ids):
        """Forward up to the channels-last prediction rows [B*T*h*w, out_channels] (activation dtype)."""
        rt = self.rt
        k = rt.k
        rt.begin_pass(0)
        self._clear_cross_pre()
        B, T, Cin, h, w = sample.shape
        mult = 2 ** sum(1 for kind, _ in self.steps if kind == "down")
        if h % mult or w % mult:
            raise ValueError(f"latent height/width must be multiples of {mult} ({mult.bit_length() - 1} stride-2 levels; SURVEY.md 0.8)")
        g = Geom(B, T, h, w)
        dev = rt.dev
        # 1. time + added-id embeddings (float, skinny path)  src/unet_spatio_temporal_condition.py:386-416
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([float(timestep)], dtype=torch.float32, device=dev)
        ts = timestep.to(device=dev, dtype=torch.float32).reshape(-1).expand(B).contiguous()
        C0 = self.config.block_out_channels[0]
        t_emb = rt.f32(B, C0)
        k.timestep_embed(ts, t_emb, B, C0)
        emb = self.time_embedding.fwd(rt, t_emb, B)
        ids = added_time_ids.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
        ad = self.config.addition_time_embed_dim
        id_emb = rt.f32(ids.numel(), ad)
        k.timestep_embed(ids, id_emb, ids.numel(), ad)
        if id_emb.numel() != B * self.add_embedding_in:
            raise ValueError(f"Model expects an added time embedding vector of length {self.add_embedding_in}, "
                             f"but a vector of {id_emb.numel() // B} was created.")
        self.add_embedding.fwd(rt, id_emb.view(B, -1), B, out=emb, accumulate=True)
        # all 44 time_emb_proj(silu(emb)) at once; emb is identical for every frame of a clip (:423)
        temb_all = rt.f32(B, self._temb_total)
        k.small_linear(emb, self._temb_w, self._temb_b, temb_all, B, self._temb_total, emb.shape[1], emb.shape[1],
                       0, 1, 0)
        ctx = ehs.to(device=dev, dtype=torch.float32).reshape(B, -1).contiguous()     # [B, 1, D] -> [B, D]
        if rt.batch_small:
            self._cross_precompute(rt, ctx, B)

        # 2. conv_in on channels-last rows (input channels zero-padded to a multiple of 64)
        x0 = rt.empty(g.M, self.cin_pad)
        k.nchw_to_rows(sample.reshape(g.N, Cin, h, w).to(torch.float32).contiguous(), x0, g.N, Cin, h, w,
                       self.cin_pad, 1.0)
        # GroupNorm statistics ride on the GEMM that writes the normalised tensor (svdx_gemm_gn): `consumer(i, geom)` is the `gn` operand
        # for the producer at step i -- the first norm of the next module when that module reads the tensor as it stands (a skip
        # connection is concatenated first: its norm spans both inputs and takes its own pass), the output norm after the last step
        steps = self.steps

        def consumer(i, geom):
            j = i + 1
            while j < len(steps) and steps[j][0] == "push":
                j += 1
            if j == len(steps):
                return self.gn_out.want(rt, geom.N, geom.HW)
            if steps[j][0] in ("res", "attn"):
                op, n_s, rows = steps[j][1].first_norm(geom)
                return op.want(rt, n_s, rows)
            return None
        want = consumer(-1, g)
        x, _, _ = self.cin_op.fwd(rt, x0, g.N, h, w, gn=want)
        pre = None if want is None else (want[0], self.cin_op.took_gn)      # (statistics buffer, filled) of x for the next first norm
        del x0

        skips: List[Tuple[torch.Tensor, Geom]] = [(x, g)]
        geoms = [g]
        cur = g
        cat_info = []
        for i, (kind, m) in enumerate(steps):
            if kind == "res" or kind == "attn":
                want = consumer(i, cur)
                x, took = m.fwd(rt, x, cur, temb_all if kind == "res" else ctx, pre=pre, want_out=want)
                pre = None if want is None else (want[0], took)
            elif kind == "push":
                skips.append((x, cur))
            elif kind == "down" or kind == "up":
                ho, wo = m.op.out_hw(cur.h, cur.w)
                nxt = Geom(B, T, ho, wo)
                want = consumer(i, nxt)
                x, ho, wo = m.op.fwd(rt, x, cur.N, cur.h, cur.w, gn=want)
                pre = None if want is None else (want[0], m.op.took_gn)
                cur = nxt
                if kind == "down":
                    geoms.append(cur)
            elif kind == "pop_cat":
                pre = None
                s, sg = skips.pop()
                assert sg.h == cur.h and sg.w == cur.w
                Ca, Cb = x.shape[1], s.shape[1]
                cat = rt.empty(cur.M, Ca + Cb)
                k.concat2(x, Ca, s, Cb, cat, cur.M)
                cat_info.append((Ca, Cb))
                x = cat
        assert not skips
        # 3. out: GroupNorm + SiLU + conv_out
        a, st = self.gn_out.fwd(rt, x, cur.N, cur.HW, pre=pre)
        y, _, _ = self.cout_op.fwd(rt, a, cur.N, cur.h, cur.w)
        self._fwd_state = dict(g0=g, x_last=x, st_last=st, cat_info=cat_info)
        return y

    def _cross_precompute(self, rt, ctx, B):
        """The KV-length-1 cross-attention of EVERY transformer block, up front: out_i = to_out_i(to_v_i(ctx)) depends on nothing but
        the clip's CLIP embed, so the 2 x 32 skinny linears of a sweep run as two table-driven launches (svdx_small_linear_batch)
        instead of 64 launches of ~5 us strung between the GEMMs.  With adapters on the value path (config 5: y += B (A x)) the chain
        has four dependency stages -- [v = W_v ctx, A_v ctx] [v += B_v ..] [out = W_o v + b, A_o v] [out += B_o ..] -- one launch each
        instead of six per block."""
        atts = [blk.attn2 for kind, m in self.steps if kind == "attn"
                for blk in list(m.transformer_blocks) + list(m.temporal_transformer_blocks)]
        stages = [[], [], [], []]

        def lin(op, x, stage, y=None):
            acc = y is not None
            y = y if acc else rt.f32(B, op.N)
            stages[stage].append((x, op.w, None if op.bias is None else op.bias.data, y, op.N, op.Kdim, op.Kdim, 0, int(acc)))
            return y
        for a in atts:
            if not a.cross_batchable():
                continue
            v = lin(a.v, ctx, 0)
            va = oa = None
            if a.v_lora is not None:
                va = lin(a.v_lora.a, ctx, 0)
                lin(a.v_lora.b, va, 1, y=v)
            o = lin(a.o, v, 2)
            if a.o_lora is not None:
                oa = lin(a.o_lora.a, v, 2)
                lin(a.o_lora.b, oa, 3, y=o)
            a._pre = (o, (v, va, oa))
        for jobs in stages:
            if jobs:
                rt.k.small_linear_batch(jobs, B, 0)

    def _backward_impl(self, d_out):
        """d_out: float [B,T,out_channels,h,w] (gradient of `.sample`)."""
        rt = self.rt
        g = self._fwd_state["g0"]
        dy = rt.empty(g.M, self.cout_pad)
        rt.k.nchw_to_rows(d_out.reshape(g.N, self.out_channels, g.h, g.w), dy, g.N, self.out_channels, g.h, g.w,
                          self.cout_pad, 1.0)
        self.backward_rows(dy)

    def backward_rows(self, dy):
        """dy: [B*T*h*w, rup(out_channels,64)] channels-last gradient of the prediction rows (zero padded)."""
        rt = self.rt
        rt.grad_overwrite, rt.grads_fresh = rt.grads_fresh, False
        rt.drop_deferred()                      # leftovers of a sweep that raised
        try:
            return self._backward_rows(dy)
        finally:
            rt.grad_overwrite = False

    def _backward_rows(self, dy):
        rt = self.rt
        k = rt.k
        fs = self._fwd_state
        self._fwd_state = None
        rt.begin_pass(1)
        g0 = fs["g0"]
        B, T = g0.B, g0.T
        cur = g0
        da = self.cout_op.bwd_dx(rt, dy, cur.N, cur.h, cur.w)
        dx = self.gn_out.bwd(rt, da, fs["x_last"], fs["st_last"], cur.N, cur.HW)
        del da
        cat_info = list(fs["cat_info"])
        # geometry per resolution level, replayed from the forward schedule
        level_geoms = [g0]
        gg = g0
        for kind, m in self.steps:
            if kind == "down":
                ho, wo = m.op.out_hw(gg.h, gg.w)
                gg = Geom(B, T, ho, wo)
                level_geoms.append(gg)
        lvl = 0
        n_skips = 1 + sum(1 for kind, _ in self.steps if kind == "push")
        # forward pops skips LIFO: the i-th pop_cat (forward order) consumed skip n_skips-1-i.  Walking the
        # schedule backwards meets pop_cats with i descending, i.e. skip indices ascending from 0, and then the
        # pushes with skip indices descending -- so a plain stack pairs them up.
        pops_left = sum(1 for kind, _ in self.steps if kind == "pop_cat")
        skip_grads: List[Optional[torch.Tensor]] = []
        for kind, m in reversed(self.steps):
            if kind == "push":
                sg = skip_grads.pop()
                if sg is not None:
                    if dx is None:
                        dx = sg
                    else:
                        tmp = rt.empty(*dx.shape)
                        k.add(dx, sg, tmp, dx.numel())
                        dx = tmp
                continue
            if kind == "up":
                lvl += 1
            elif kind == "down":
                lvl -= 1
            if dx is None:
                cur = level_geoms[lvl]
                if kind == "res":
                    m.spatial_res_block.sv = m.temporal_res_block.sv = None
                elif kind == "attn":
                    m.sv = None
                    for b in list(m.transformer_blocks) + list(m.temporal_transformer_blocks):
                        b.sv = None
                continue
            if kind == "res" or kind == "attn":
                dx = m.bwd(rt, dx, cur)
                if kind == "attn" and self.grads_ready_cb is not None:
                    if self.grads_ready_flush:
                        rt.flush_deferred()         # the queued skinny gradient launches of this block belong to its bucket
                    self.grads_ready_cb(m)          # this block's weight gradients are final: the trainer may start reducing them
            elif kind == "up":
                low = level_geoms[lvl]
                dx = m.op.bwd_dx(rt, dx, low.N, low.h, low.w) if m.need_dx else None
            elif kind == "down":
                hi = level_geoms[lvl]
                dx = m.op.bwd_dx(rt, dx, hi.N, hi.h, hi.w) if m.need_dx else None
            elif kind == "pop_cat":
                Ca, Cb = cat_info.pop()
                skip_no = n_skips - pops_left
                pops_left -= 1
                da_ = rt.empty(cur.M, Ca)
                db_ = rt.empty(cur.M, Cb)
                k.split2(dx, da_, Ca, db_, Cb, cur.M)
                dx = da_
                skip_grads.append(db_ if self._skip_needs_grad[skip_no] else None)
            cur = level_geoms[lvl]
        # the conv_in skip (index 0) and conv_in itself carry no trainable parameters upstream
        rt.flush_deferred()
        return None
