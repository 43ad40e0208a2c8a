"""MI355X-native encoder half of `AutoencoderKLTemporalDecoder` -- the step right before the UNet in the reference loop:
`tensor_to_vae_latent` (/root/reference/train_svd.py:283-291), called on the 14 frames of a clip (:948) and on its noised
conditioning frame (:957-960); the VAE is frozen (:659) and loaded from `<model>/vae` (:649-650).  SURVEY.md 8(f) rank 1.

Same surface as the diffusers class for what the train script touches: `from_pretrained(path, subfolder="vae", variant=...)`,
`.config.scaling_factor`, `.requires_grad_(False)`, `.to(device, dtype=...)`, `vae.encode(x).latent_dist.sample()` /
`.mode()` / `.mean` / `.logvar`; diffusers state-dict keys (`encoder.*`, `quant_conv.*`; the `decoder.*` half of a checkpoint is
loaded too: the validation sampler decodes).  The arithmetic is libsvdx kernel launches on channels-last rows [n*h*w, C]:

  conv_in (3 -> C0)          svdx_patch_rows (im2col of the 3-channel image, K = 27 padded to 64) + plain GEMM
  ResnetBlock2D              svdx_gn_stats/apply (+SiLU), implicit-GEMM 3x3 convs, the 1x1 shortcut / identity as the GEMM residual
  Downsample2D(padding=0)    the SVDX_GATHER_CONV3X3_PAD0 gather: stride 2 over F.pad(x, (0, 1, 0, 1)) without materialising the pad
  mid-block attention        one head of dim C over h*w tokens: fused q/k/v GEMM, per frame S = q k^T (GEMM, alpha = C^-1/2),
                             svdx_softmax_rows, O = P v (GEMM against the transposed v), out-projection + residual in the epilogue
  conv_out + quant_conv      ONE 3x3 conv: the 1x1 `quant_conv` is folded into conv_out's weights at prepare() (exact: both linear)

Forward only (no gradients exist on this path).  Frames are processed in chunks that keep every operand below the 2 GiB range of
the GEMM's buffer addressing (51 frames at 512x320, 14 at 1024x576).

The TEMPORAL DECODER half (`vae.decode(z, num_frames)`) is the last stage of the validation sampler (train_svd.py:1106-1137 through
StableVideoDiffusionPipeline.decode_latents; svd_xtend_amd/pipeline.py): diffusers' `TemporalDecoder` --
conv_in, a mid block (SpatioTemporalResBlock, single-head attention, SpatioTemporalResBlock), four up blocks of three
SpatioTemporalResBlocks (+ nearest-x2 upsample folded into the following conv's gather), GroupNorm + SiLU + conv_out, and the
Conv3d (3,1,1) `time_conv_out`.  A SpatioTemporalResBlock here has no time embedding and a "learned" AlphaBlender with
`switch_spatial_to_temporal_mix`: out = s + sigmoid(mix_factor) * H(s) with s the spatial resnet's output and H the temporal
resnet's residual branch -- the factor is folded into the temporal conv2 weights at prepare() (as in the UNet's resnets)."""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import kernels as K
from .ops import ConvOp, GroupNormOp, LinearOp, Runtime, gemm_act, rup
from .unet import FrozenConfig

CONFIG_NAME = "config.json"
WEIGHTS_NAME = "diffusion_pytorch_model{variant}.safetensors"


class _Resnet(nn.Module):
    """diffusers ResnetBlock2D without a time embedding (eps 1e-6, 32 groups, SiLU)."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(32, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def build(self, rt: Runtime) -> None:
        self.gn1, self.gn2 = GroupNormOp(self.norm1, True), GroupNormOp(self.norm2, True)
        self.c1 = ConvOp(self.conv1.weight, self.conv1.bias, "3x3")
        self.c2 = ConvOp(self.conv2.weight, self.conv2.bias, "3x3")
        self.sc = ConvOp(self.conv_shortcut.weight, self.conv_shortcut.bias, "1x1") if self.conv_shortcut is not None else None
        for op in (self.c1, self.c2, self.sc):
            if op is not None:
                op.pack(rt, need_dx=False)

    def fwd(self, rt: Runtime, x, n: int, h: int, w: int, pre=None, want_out=None):
        """pre: (stats, filled) when the launch that wrote x already took norm1's statistics (GroupNormOp.fwd); want_out: what the norm
        that reads THIS block's output wants from the launch that writes it (GroupNormOp.want) -- `self.took_out` says whether it got them.
        As in the UNet's resnets, the statistics passes over the full-resolution tensors ride on the producing GEMM's store loop."""
        a, _ = self.gn1.fwd(rt, x, n, h * w, pre=pre)
        w2 = self.gn2.want(rt, n, h * w)
        y, _, _ = self.c1.fwd(rt, a, n, h, w, gn=w2)
        a, _ = self.gn2.fwd(rt, y, n, h * w, pre=(w2[0], self.c1.took_gn) if w2 is not None else None)
        sc = x if self.sc is None else self.sc.fwd(rt, x, n, h, w)[0]
        out = self.c2.fwd(rt, a, n, h, w, res=sc, gn=want_out)[0]
        self.took_out = want_out is not None and self.c2.took_gn
        return out


class _Downsample(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def build(self, rt: Runtime) -> None:
        self.op = ConvOp(self.conv.weight, self.conv.bias, "3x3", stride=2, pad0=True)
        self.op.pack(rt, need_dx=False)


class _DownBlock(nn.Module):
    def __init__(self, cin: int, cout: int, layers: int, add_downsample: bool):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if i == 0 else cout, cout) for i in range(layers)])
        if add_downsample:
            self.downsamplers = nn.ModuleList([_Downsample(cout)])


class _Attention(nn.Module):
    """diffusers Attention as UNetMidBlock2D builds it for the VAE: ONE head of dim C, GroupNorm on the input, biased projections,
    residual connection."""

    def __init__(self, c: int):
        super().__init__()
        self.c = c
        self.group_norm = nn.GroupNorm(32, c, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def build(self, rt: Runtime) -> None:
        self.gn = GroupNormOp(self.group_norm, False)
        self.qkv = LinearOp([self.to_q.weight, self.to_k.weight, self.to_v.weight], [self.to_q.bias, self.to_k.bias, self.to_v.bias])
        self.o = LinearOp([self.to_out[0].weight], [self.to_out[0].bias])
        self.qkv.pack(rt, need_dx=False)
        self.o.pack(rt, need_dx=False)

    def fwd(self, rt: Runtime, x, n: int, h: int, w: int):
        k, C, S = rt.k, self.c, h * w
        M = n * S
        y, _ = self.gn.fwd(rt, x, n, S)
        qkv = self.qkv.fwd(rt, y, M)                               # [M, 3C]
        del y
        Sp = rup(S, 64)                                            # the P v GEMM reduces over the tokens: K granule 64
        att = rt.empty(M, C)
        s = rt.empty(S, Sp)                                        # scores, row pitch Sp (16-byte aligned rows for any S)
        p = rt.empty(S, Sp)
        vt = rt.empty(C, Sp)                                       # v^T; svdx_transpose zero-fills the columns S..Sp
        for f in range(n):
            q_f, k_f, v_f = qkv[f * S:], qkv[f * S:, C:], qkv[f * S:, 2 * C:]
            gemm_act(rt, q_f, k_f, s, S, S, C, 3 * C, 3 * C, Sp, alpha=C ** -0.5)      # softmax(q k^T / sqrt(C)): the scale rides on alpha
            k.softmax_rows(s, p, S, S, Sp, Sp, Sp, 1.0)
            k.transpose(v_f, 3 * C, vt, Sp, S, C)
            gemm_act(rt, p, vt, att[f * S:], S, C, Sp, Sp, Sp, C)
        return self.o.fwd(rt, att, M, res=x)


class _MidBlock(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.attentions = nn.ModuleList([_Attention(c)])
        self.resnets = nn.ModuleList([_Resnet(c, c), _Resnet(c, c)])


class _Encoder(nn.Module):
    """diffusers.models.autoencoders.vae.Encoder(double_z=True)."""

    def __init__(self, in_channels: int, latent_channels: int, block_out_channels: Sequence[int], layers_per_block: int):
        super().__init__()
        ch = list(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        cout = ch[0]
        for i, c in enumerate(ch):
            cin, cout = cout, c
            self.down_blocks.append(_DownBlock(cin, cout, layers_per_block, add_downsample=i != len(ch) - 1))
        self.mid_block = _MidBlock(ch[-1])
        self.conv_norm_out = nn.GroupNorm(32, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * latent_channels, 3, padding=1)


class _TemporalResnet(nn.Module):
    """diffusers TemporalResnetBlock(temb_channels=None): 3-D GroupNorm (a clip is one sample of T*h*w rows), Conv3d (3,1,1)."""

    def __init__(self, c: int, eps: float):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, c, eps=eps)
        self.conv1 = nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))
        self.norm2 = nn.GroupNorm(32, c, eps=eps)
        self.conv2 = nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))

    def build(self, rt: Runtime, out_scale: float) -> None:
        self.gn1, self.gn2 = GroupNormOp(self.norm1, True), GroupNormOp(self.norm2, True)
        self.c1 = ConvOp(self.conv1.weight, self.conv1.bias, "t3")
        self.c2 = ConvOp(self.conv2.weight, self.conv2.bias, "t3")
        self.c1.pack(rt, need_dx=False)
        self.c2.pack(rt, need_dx=False, out_scale=out_scale)

    def fwd(self, rt: Runtime, x, B: int, T: int, h: int, w: int):
        a, _ = self.gn1.fwd(rt, x, B, T * h * w)
        y, _, _ = self.c1.fwd(rt, a, B, h, w, T=T)
        a, _ = self.gn2.fwd(rt, y, B, T * h * w)
        return self.c2.fwd(rt, a, B, h, w, T=T, res=x)[0]


class _Mixer(nn.Module):
    def __init__(self):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([0.0]))


class _STResBlock(nn.Module):
    """SpatioTemporalResBlock(temb_channels=None, eps=1e-6, temporal_eps=1e-5, merge_strategy="learned", merge_factor=0.0,
    switch_spatial_to_temporal_mix=True)."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.spatial_res_block = _Resnet(cin, cout)
        self.temporal_res_block = _TemporalResnet(cout, 1e-5)
        self.time_mixer = _Mixer()

    def build(self, rt: Runtime) -> None:
        self.spatial_res_block.build(rt)
        # alpha = 1 - sigmoid(mix): out = alpha s + (1 - alpha)(s + H(s)) = s + sigmoid(mix) H(s)
        self.temporal_res_block.build(rt, float(torch.sigmoid(self.time_mixer.mix_factor.data.float()).item()))

    def fwd(self, rt: Runtime, x, B: int, T: int, h: int, w: int):
        s = self.spatial_res_block.fwd(rt, x, B * T, h, w)
        return self.temporal_res_block.fwd(rt, s, B, T, h, w)


class _Upsample(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def build(self, rt: Runtime) -> None:
        self.op = ConvOp(self.conv.weight, self.conv.bias, "3x3", ups=True)
        self.op.pack(rt, need_dx=False)


class _MidBlockDec(nn.Module):
    def __init__(self, c: int, layers: int):
        super().__init__()
        self.attentions = nn.ModuleList([_Attention(c)])
        self.resnets = nn.ModuleList([_STResBlock(c, c) for _ in range(layers)])


class _UpBlockDec(nn.Module):
    def __init__(self, cin: int, cout: int, layers: int, add_upsample: bool):
        super().__init__()
        self.resnets = nn.ModuleList([_STResBlock(cin if i == 0 else cout, cout) for i in range(layers)])
        if add_upsample:
            self.upsamplers = nn.ModuleList([_Upsample(cout)])


class _Decoder(nn.Module):
    """diffusers.models.autoencoders.autoencoder_kl_temporal_decoder.TemporalDecoder."""

    def __init__(self, in_channels: int, out_channels: int, block_out_channels: Sequence[int], layers_per_block: int):
        super().__init__()
        ch = list(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, ch[-1], 3, padding=1)
        self.mid_block = _MidBlockDec(ch[-1], layers_per_block)
        self.up_blocks = nn.ModuleList()
        rev = ch[::-1]
        cout = rev[0]
        for i, c in enumerate(rev):
            cin, cout = cout, c
            self.up_blocks.append(_UpBlockDec(cin, cout, layers_per_block + 1, add_upsample=i != len(ch) - 1))
        self.conv_norm_out = nn.GroupNorm(32, ch[0], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[0], out_channels, 3, padding=1)
        self.time_conv_out = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))


class DiagonalGaussianDistribution:
    """diffusers.models.autoencoders.vae.DiagonalGaussianDistribution over moments given as (mean, logvar) [n, c, h, w] floats."""

    def __init__(self, mean: torch.Tensor, logvar: torch.Tensor):
        self.mean = mean
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        eps = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * eps

    def mode(self) -> torch.Tensor:
        return self.mean


class AutoencoderKLTemporalDecoder(nn.Module):
    """Constructor arguments of the diffusers class that shape the ENCODER (everything else in a `vae/config.json` is accepted and
    kept in `.config`)."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, latent_channels: int = 4,
                 block_out_channels: Sequence[int] = (128, 256, 512, 512), layers_per_block: int = 2,
                 scaling_factor: float = 0.18215, **other):
        super().__init__()
        for c in block_out_channels:
            if c % 64:
                raise ValueError("block_out_channels must be multiples of 64 (the implicit-GEMM K granule)")
        self.config = FrozenConfig(in_channels=in_channels, out_channels=out_channels, latent_channels=latent_channels,
                                   block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                   scaling_factor=scaling_factor, **other)
        self.encoder = _Encoder(in_channels, latent_channels, block_out_channels, layers_per_block)
        self.decoder = _Decoder(latent_channels, out_channels, block_out_channels, layers_per_block)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.rt: Optional[Runtime] = None
        self._requested_dtype = None

    # ---- loading ------------------------------------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = None, variant: Optional[str] = None, torch_dtype=None, **unused):
        """`<path>/<subfolder>/config.json` + `diffusion_pytorch_model[.<variant>].safetensors` (train_svd.py:649-650), loaded
        strictly (encoder, decoder, quant_conv)."""
        from safetensors.torch import load_file
        folder = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(folder, CONFIG_NAME)) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        model = cls(**cfg)
        wpath = os.path.join(folder, WEIGHTS_NAME.format(variant=f".{variant}" if variant else ""))
        if not os.path.exists(wpath) and variant:
            wpath = os.path.join(folder, WEIGHTS_NAME.format(variant=""))
        sd = {k: v.float() for k, v in load_file(wpath).items()}
        model.load_state_dict(sd, strict=True)
        if torch_dtype is not None and torch_dtype != torch.float32:
            model._requested_dtype = torch_dtype
        return model

    def to(self, *args, **kwargs):
        """`vae.to(device, dtype=weight_dtype)` (train_svd.py:738): float masters stay fp32, a 16-bit dtype selects the activation type."""
        device, dtype, non_blocking, _ = torch._C._nn._parse_to(*args, **kwargs)
        if dtype is not None and dtype.is_floating_point and dtype != torch.float32:
            self._requested_dtype = dtype
            return super().to(device=device, non_blocking=non_blocking) if device is not None else self
        return super().to(*args, **kwargs)

    # ---- packing ------------------------------------------------------------------------------------------------------
    def prepare(self, dtype: Optional[torch.dtype] = None) -> "AutoencoderKLTemporalDecoder":
        dtype = dtype or self._requested_dtype or torch.float16
        dev = next(self.parameters()).device
        self.requires_grad_(False)
        self.rt = rt = Runtime(dtype, dev)
        enc = self.encoder
        cin, c0 = self.config.in_channels, self.config.block_out_channels[0]
        self.k_in = rup(cin * 9, 64)
        w_in = torch.zeros(c0, self.k_in, dtype=torch.float32, device=dev)
        w_in[:, :cin * 9] = enc.conv_in.weight.data.reshape(c0, cin * 9)           # k = (c*3 + dy)*3 + dx: svdx_patch_rows order
        self.w_in = rt.empty(c0, self.k_in)
        rt.k.cast_from_f32(w_in.reshape(-1), self.w_in, w_in.numel())
        for blk in enc.down_blocks:
            for r in blk.resnets:
                r.build(rt)
            if hasattr(blk, "downsamplers"):
                blk.downsamplers[0].build(rt)
        for r in enc.mid_block.resnets:
            r.build(rt)
        enc.mid_block.attentions[0].build(rt)
        self.gn_out = GroupNormOp(enc.conv_norm_out, True)
        # quant_conv (1x1, 2z -> 2z) after conv_out (3x3, C -> 2z): one convolution with weights Q W and bias Q b + q
        Q = self.quant_conv.weight.data.reshape(self.quant_conv.weight.shape[0], -1).float()
        Wf = torch.einsum("oc,cikl->oikl", Q, enc.conv_out.weight.data.float())
        bf = Q @ enc.conv_out.bias.data.float() + self.quant_conv.bias.data.float()
        self._folded = (nn.Parameter(Wf.contiguous(), requires_grad=False), nn.Parameter(bf.contiguous(), requires_grad=False))
        self.c_out = ConvOp(self._folded[0], self._folded[1], "3x3")
        self.c_out.pack(rt, need_dx=False)
        self._prepare_decoder(rt)
        return self

    def _prepare_decoder(self, rt: Runtime) -> None:
        dec, dev = self.decoder, rt.dev
        self.zpad = 64                                             # latent channels zero-padded to the K granule of the gather
        self.d_in = ConvOp(dec.conv_in.weight, dec.conv_in.bias, "3x3", cin_pad=self.zpad)
        self.d_in.pack(rt, need_dx=False)
        for r in dec.mid_block.resnets:
            r.build(rt)
        dec.mid_block.attentions[0].build(rt)
        for blk in dec.up_blocks:
            for r in blk.resnets:
                r.build(rt)
            if hasattr(blk, "upsamplers"):
                blk.upsamplers[0].build(rt)
        self.d_gn_out = GroupNormOp(dec.conv_norm_out, True)
        # conv_out (C0 -> 3) and time_conv_out (3 -> 3 over frames): output channels padded to 8 with zero filters; conv_out writes
        # its 8 columns into zeroed 64-wide rows, the input pitch of the temporal conv's gather
        co, c0 = self.config.out_channels, self.config.block_out_channels[0]
        self.opad = 8
        w = torch.zeros(self.opad, c0, 3, 3, dtype=torch.float32, device=dev)
        b = torch.zeros(self.opad, dtype=torch.float32, device=dev)
        w[:co], b[:co] = dec.conv_out.weight.data.float(), dec.conv_out.bias.data.float()
        wt = torch.zeros(self.opad, co, 3, 1, 1, dtype=torch.float32, device=dev)
        bt = torch.zeros(self.opad, dtype=torch.float32, device=dev)
        wt[:co], bt[:co] = dec.time_conv_out.weight.data.float(), dec.time_conv_out.bias.data.float()
        self._dec_padded = [nn.Parameter(t, requires_grad=False) for t in (w, b, wt, bt)]
        self.d_out = ConvOp(self._dec_padded[0], self._dec_padded[1], "3x3")
        self.d_out.pack(rt, need_dx=False)
        self.d_tout = ConvOp(self._dec_padded[2], self._dec_padded[3], "t3", cin_pad=self.zpad)
        self.d_tout.pack(rt, need_dx=False)

    # ---- forward ------------------------------------------------------------------------------------------------------
    def _moments_rows(self, x: torch.Tensor):
        """x float [n, cin, H, W] on the device -> (rows [n*h*w, 2z] in the activation dtype, h, w)."""
        rt, enc = self.rt, self.encoder
        k = rt.k
        n, cin, H, W = x.shape
        rt.begin_pass(0)
        a = rt.empty(n * H * W, self.k_in)
        k.patch_rows(x.contiguous(), a, n, cin, H, W, 3, 3, 1, 1, H, W, self.k_in)
        c0 = self.config.block_out_channels[0]
        y = rt.empty(n * H * W, c0)
        # the chain of stages; every producer (conv_in, a resnet's second convolution, a downsampling convolution) is told what the NEXT
        # stage's first GroupNorm wants, so that the statistics passes over the full-resolution tensors ride on the producing GEMMs
        blocks = list(enc.down_blocks)
        mid = enc.mid_block
        first = blocks[0].resnets[0].gn1.want(rt, n, H * W)
        took = gemm_act(rt, a, self.w_in, y, n * H * W, c0, self.k_in, self.k_in, self.k_in, c0, bias=enc.conv_in.bias.data, gn=first)
        pre = (first[0], took) if first is not None else None
        del a
        h, w = H, W
        for bi, blk in enumerate(blocks):
            rs = list(blk.resnets)
            down = blk.downsamplers[0].op if hasattr(blk, "downsamplers") else None
            after = (blocks[bi + 1].resnets[0] if bi + 1 < len(blocks) else mid.resnets[0]).gn1      # the norm behind this block
            for i, r in enumerate(rs):
                nxt = rs[i + 1].gn1.want(rt, n, h * w) if i + 1 < len(rs) else (after.want(rt, n, h * w) if down is None else None)
                y = r.fwd(rt, y, n, h, w, pre=pre, want_out=nxt)
                pre = (nxt[0], r.took_out) if nxt is not None else None
            if down is not None:
                ho, wo = down.out_hw(h, w)
                nxt = after.want(rt, n, ho * wo)
                y, h, w = down.fwd(rt, y, n, h, w, gn=nxt)
                pre = (nxt[0], down.took_gn) if nxt is not None else None
        y = mid.resnets[0].fwd(rt, y, n, h, w, pre=pre)
        y = mid.attentions[0].fwd(rt, y, n, h, w)
        want = self.gn_out.want(rt, n, h * w)
        y = mid.resnets[1].fwd(rt, y, n, h, w, want_out=want)
        a, _ = self.gn_out.fwd(rt, y, n, h * w, pre=(want[0], mid.resnets[1].took_out) if want is not None else None)
        return self.c_out.fwd(rt, a, n, h, w)[0], h, w

    def max_frames(self, H: int, W: int) -> int:
        """Frames per chunk that keep the largest activation (C0 channels at full resolution, or the padded im2col rows of conv_in)
        inside the 2 GiB reach of the GEMM's 32-bit buffer offsets."""
        per = H * W * max(self.config.block_out_channels[0], getattr(self, "k_in", 64)) * 2
        return max(1, (2 ** 31 - 1) // per)

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """x [n, 3, H, W] in [-1, 1] (H, W multiples of 8) -> `.latent_dist` over [n, latent_channels, H/8, W/8]."""
        if self.rt is None:
            self.prepare()
        rt = self.rt
        if x.ndim != 4 or x.shape[1] != self.config.in_channels:
            raise ValueError(f"expected [n, {self.config.in_channels}, H, W], got {tuple(x.shape)}")
        n, _, H, W = x.shape
        down = 2 ** (len(self.config.block_out_channels) - 1)
        if H % down or W % down:
            raise ValueError(f"height and width must be multiples of {down}")
        x = x.to(device=rt.dev, dtype=torch.float32)
        z2 = 2 * self.config.latent_channels
        out = torch.empty(n, z2, H // down, W // down, dtype=torch.float32, device=rt.dev)
        step = self.max_frames(H, W)
        for i in range(0, n, step):
            rows, h, w = self._moments_rows(x[i:i + step])
            m = min(step, n - i)
            rt.k.rows_to_nchw(rows, out[i:i + m], m, z2, h, w, z2)
        mean, logvar = torch.chunk(out, 2, dim=1)
        dist = DiagonalGaussianDistribution(mean, logvar)
        if not return_dict:
            return (dist,)
        return SimpleNamespace(latent_dist=dist)

    def max_decode_frames(self, h: int, w: int) -> int:
        """Frames per `decode` call that keep the largest decoder activation (the upsampled input of the last up block, or the
        64-wide rows in front of `time_conv_out`) inside the 2 GiB reach of the fast GEMM kernel's 32-bit buffer offsets; larger
        chunks decode correctly through the 64-bit-pointer kernel, more slowly."""
        ch = self.config.block_out_channels
        s = 2 ** (len(ch) - 1)
        per = max(ch[1] * (h * s) * (w * s), ch[-1] * (h * s // 2) * (w * s // 2), 64 * (h * s) * (w * s)) * 2
        return max(1, (2 ** 31 - 1) // per)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, num_frames: int = 1, image_only_indicator=None, return_dict: bool = True):
        """z [b*num_frames, latent_channels, h, w] (already divided by `scaling_factor`, as the pipeline does) -> `.sample`
        [b*num_frames, out_channels, 8h, 8w] floats.  `image_only_indicator` is accepted and unused: the decoder's blenders are
        "learned", not "learned_with_images"."""
        if self.rt is None:
            self.prepare()
        rt, dec = self.rt, self.decoder
        k = rt.k
        if z.ndim != 4 or z.shape[1] != self.config.latent_channels or z.shape[0] % num_frames:
            raise ValueError(f"expected [b*{num_frames}, {self.config.latent_channels}, h, w], got {tuple(z.shape)}")
        n, zc, h, w = z.shape
        # n > max_decode_frames(h, w) is fine: svdx_gemm takes its 64-bit-pointer kernel for any operand beyond the 2 GiB reach of the
        # buffer-addressed one (the reference's validation default, 8 frames of 72x128 latents, has 2.4 GB activations at full resolution)
        B, T = n // num_frames, num_frames
        rt.begin_pass(0)
        x0 = rt.empty(n * h * w, self.zpad)
        k.nchw_to_rows(z.to(device=rt.dev, dtype=torch.float32).contiguous(), x0, n, zc, h, w, self.zpad, 1.0)
        y, _, _ = self.d_in.fwd(rt, x0, n, h, w)
        del x0
        mid = dec.mid_block
        y = mid.resnets[0].fwd(rt, y, B, T, h, w)
        for resnet, attn in zip(mid.resnets[1:], mid.attentions):
            y = attn.fwd(rt, y, n, h, w)
            y = resnet.fwd(rt, y, B, T, h, w)
        for blk in dec.up_blocks:
            for r in blk.resnets:
                y = r.fwd(rt, y, B, T, h, w)
            if hasattr(blk, "upsamplers"):
                y, h, w = blk.upsamplers[0].op.fwd(rt, y, n, h, w)
        a, _ = self.d_gn_out.fwd(rt, y, n, h * w)
        del y
        rows = rt.empty(n * h * w, self.zpad)
        k.zero(rows)
        self.d_out.fwd(rt, a, n, h, w, ldc=self.zpad, out=rows)
        del a
        out_rows, _, _ = self.d_tout.fwd(rt, rows, B, h, w, T=T)    # [n*h*w, 8]
        co = self.config.out_channels
        out = torch.empty(n, co, h, w, dtype=torch.float32, device=rt.dev)
        k.rows_to_nchw(out_rows, out, n, co, h, w, self.opad)
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)


def tensor_to_vae_latent(t: torch.Tensor, vae: AutoencoderKLTemporalDecoder, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """train_svd.py:283-291 verbatim in meaning: [b, f, c, h, w] pixels -> [b, f, 4, h/8, w/8] latents x scaling_factor."""
    b, f = t.shape[:2]
    z = vae.encode(t.reshape(b * f, *t.shape[2:])).latent_dist.sample(generator)
    return z.reshape(b, f, *z.shape[1:]) * vae.config.scaling_factor
