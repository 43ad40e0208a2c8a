"""CPU oracle for the step that precedes the UNet in the reference loop: `tensor_to_vae_latent`
(/root/reference/train_svd.py:283-291, called at :948 and :959-960) = the ENCODER half of diffusers'
`AutoencoderKLTemporalDecoder` + `DiagonalGaussianDistribution.sample()` x `scaling_factor`.

TEST INFRASTRUCTURE ONLY (see oracle/unet.py header); the product counterpart is svd_xtend_amd/vae.py.  PARITY UNPINNED: diffusers
is not installed here; the encoder is restated from the published module (diffusers.models.autoencoders.vae.Encoder with
DownEncoderBlock2D x4, UNetMidBlock2D with one single-head attention, `quant_conv`), self-pinned only structurally
(tests/test_oracle_vae.py): the diffusers state-dict key set and the parameter count of the SD / SVD VAE encoder (34,163,592 + 72
for quant_conv).  tests/test_diffusers_pin.py compares it with the installed class the day diffusers is importable.

The second half of the file is the TEMPORAL DECODER (`vae.decode(z, num_frames)`), the last stage of the validation sampler
(/root/reference/train_svd.py:1106-1137 through StableVideoDiffusionPipeline.decode_latents): restated from
diffusers 0.26 `autoencoder_kl_temporal_decoder.TemporalDecoder` (MidBlockTemporalDecoder / UpBlockTemporalDecoder built from
`SpatioTemporalResBlock(temb_channels=None, eps=1e-6, temporal_eps=1e-5, merge_strategy="learned", merge_factor=0.0,
switch_spatial_to_temporal_mix=True)`, a Conv3d (3,1,1) `time_conv_out` after `conv_out`).  Equally unpinned.

Module / parameter names are diffusers' (`encoder.down_blocks.0.resnets.0.norm1.weight`, `encoder.mid_block.attentions.0.to_q.weight`,
`quant_conv.weight`, ...), so `state_dict()` keys match the `encoder.*` / `quant_conv.*` subset of a real SVD `vae/` checkpoint.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

SVD_VAE_CONFIG = dict(in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                      scaling_factor=0.18215)
TINY_VAE_CONFIG = dict(in_channels=3, latent_channels=4, block_out_channels=(32, 64, 64, 64), layers_per_block=1,
                       scaling_factor=0.18215)


class ResnetBlock2D(nn.Module):
    """diffusers.models.resnet.ResnetBlock2D without a time embedding (temb_channels=None), eps 1e-6, 32 groups, SiLU."""

    def __init__(self, cin: int, cout: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))              # dropout p = 0
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h                                       # output_scale_factor = 1


class Downsample2D(nn.Module):
    """diffusers Downsample2D(use_conv=True, padding=0): zero-pad one column on the right and one row at the bottom, then a
    3x3 stride-2 convolution without padding (key `downsamplers.0.conv`)."""

    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class DownEncoderBlock2D(nn.Module):
    def __init__(self, cin: int, cout: int, layers: int, add_downsample: bool):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout) for i in range(layers)])
        if add_downsample:
            self.downsamplers = nn.ModuleList([Downsample2D(cout)])

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if hasattr(self, "downsamplers"):
            x = self.downsamplers[0](x)
        return x


class Attention(nn.Module):
    """diffusers Attention as UNetMidBlock2D builds it for the VAE: one head of dim C, GroupNorm(32, eps 1e-6) on the input,
    biased q/k/v/out projections, residual connection, softmax in fp32 (upcast_softmax)."""

    def __init__(self, c: int, groups: int = 32, eps: float = 1e-6):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        res = x
        y = self.group_norm(x.view(b, c, h * w)).transpose(1, 2)              # [b, hw, c]
        q, k, v = self.to_q(y), self.to_k(y), self.to_v(y)
        p = torch.softmax((q @ k.transpose(1, 2)).float() * c ** -0.5, dim=-1).to(v.dtype)
        y = self.to_out[0](p @ v)
        return y.transpose(1, 2).reshape(b, c, h, w) + res                      # rescale_output_factor = 1


class UNetMidBlock2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.attentions = nn.ModuleList([Attention(c)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c), ResnetBlock2D(c, c)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class Encoder(nn.Module):
    """diffusers.models.autoencoders.vae.Encoder(double_z=True)."""

    def __init__(self, in_channels: int, latent_channels: int, block_out_channels: Sequence[int], layers_per_block: int):
        super().__init__()
        ch = list(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        cout = ch[0]
        for i, c in enumerate(ch):
            cin, cout = cout, c
            self.down_blocks.append(DownEncoderBlock2D(cin, cout, layers_per_block, add_downsample=i != len(ch) - 1))
        self.mid_block = UNetMidBlock2D(ch[-1])
        self.conv_norm_out = nn.GroupNorm(32, ch[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[-1], 2 * latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for blk in self.down_blocks:
            x = blk(x)
        x = self.mid_block(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class VaeEncoderOracle(nn.Module):
    """`vae.encode(x).latent_dist` of AutoencoderKLTemporalDecoder: encoder -> quant_conv -> (mean, logvar clamped to [-30, 20])."""

    def __init__(self, in_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 scaling_factor=0.18215):
        super().__init__()
        self.encoder = Encoder(in_channels, latent_channels, block_out_channels, layers_per_block)
        self.quant_conv = nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1)
        self.scaling_factor = scaling_factor

    def moments(self, x):
        mean, logvar = torch.chunk(self.quant_conv(self.encoder(x)), 2, dim=1)
        return mean, torch.clamp(logvar, -30.0, 20.0)

    def sample(self, x, generator: Optional[torch.Generator] = None):
        """DiagonalGaussianDistribution.sample(): mean + std * randn (the noise is drawn in the moments' shape, dtype, device)."""
        mean, logvar = self.moments(x)
        eps = torch.randn(mean.shape, generator=generator, dtype=mean.dtype, device=mean.device)
        return mean + torch.exp(0.5 * logvar) * eps


def tensor_to_vae_latent(t: torch.Tensor, vae: VaeEncoderOracle, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """train_svd.py:283-291: [b, f, c, h, w] pixels in [-1, 1] -> [b, f, 4, h/8, w/8] latents x scaling_factor."""
    b, f = t.shape[:2]
    z = vae.sample(t.reshape(b * f, *t.shape[2:]), generator)
    return z.reshape(b, f, *z.shape[1:]) * vae.scaling_factor


# --------------------------------------------------------------------------------------------------------------------------------
# temporal decoder (diffusers.models.autoencoders.autoencoder_kl_temporal_decoder.TemporalDecoder)
# --------------------------------------------------------------------------------------------------------------------------------
class TemporalResnetBlock(nn.Module):
    """diffusers.models.resnet.TemporalResnetBlock(temb_channels=None): GroupNorm over (C/32, T, H, W), Conv3d (3,1,1), SiLU,
    identity shortcut (in == out everywhere in the decoder)."""

    def __init__(self, c: int, eps: float):
        super().__init__()
        self.norm1 = nn.GroupNorm(32, c, eps=eps)
        self.conv1 = nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))
        self.norm2 = nn.GroupNorm(32, c, eps=eps)
        self.conv2 = nn.Conv3d(c, c, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, x):                                  # [b, c, t, h, w]
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return x + h


class LearnedBlender(nn.Module):
    """AlphaBlender(alpha=0.0, merge_strategy="learned", switch_spatial_to_temporal_mix=True): alpha = 1 - sigmoid(mix_factor)."""

    def __init__(self):
        super().__init__()
        self.mix_factor = nn.Parameter(torch.tensor([0.0]))

    def forward(self, x_spatial, x_temporal):
        alpha = 1.0 - torch.sigmoid(self.mix_factor)
        return alpha * x_spatial + (1.0 - alpha) * x_temporal


class SpatioTemporalResBlockDec(nn.Module):
    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.spatial_res_block = ResnetBlock2D(cin, cout, eps=1e-6)
        self.temporal_res_block = TemporalResnetBlock(cout, eps=1e-5)
        self.time_mixer = LearnedBlender()

    def forward(self, x, num_frames: int):                 # [b*t, c, h, w]
        x = self.spatial_res_block(x)
        bt, c, h, w = x.shape
        x5 = x.reshape(bt // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        out = self.time_mixer(x5, self.temporal_res_block(x5))
        return out.permute(0, 2, 1, 3, 4).reshape(bt, c, h, w)


class MidBlockTemporalDecoder(nn.Module):
    def __init__(self, c: int, layers: int):
        super().__init__()
        self.attentions = nn.ModuleList([Attention(c)])
        self.resnets = nn.ModuleList([SpatioTemporalResBlockDec(c, c) for _ in range(layers)])

    def forward(self, x, num_frames: int):
        x = self.resnets[0](x, num_frames)
        for resnet, attn in zip(self.resnets[1:], self.attentions):
            x = resnet(attn(x), num_frames)
        return x


class Upsample2D(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class UpBlockTemporalDecoder(nn.Module):
    def __init__(self, cin: int, cout: int, layers: int, add_upsample: bool):
        super().__init__()
        self.resnets = nn.ModuleList([SpatioTemporalResBlockDec(cin if i == 0 else cout, cout) for i in range(layers)])
        if add_upsample:
            self.upsamplers = nn.ModuleList([Upsample2D(cout)])

    def forward(self, x, num_frames: int):
        for r in self.resnets:
            x = r(x, num_frames)
        if hasattr(self, "upsamplers"):
            x = self.upsamplers[0](x)
        return x


class TemporalDecoder(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, block_out_channels: Sequence[int], layers_per_block: int):
        super().__init__()
        ch = list(block_out_channels)
        self.conv_in = nn.Conv2d(in_channels, ch[-1], 3, padding=1)
        self.mid_block = MidBlockTemporalDecoder(ch[-1], layers_per_block)
        self.up_blocks = nn.ModuleList()
        rev = ch[::-1]
        cout = rev[0]
        for i, c in enumerate(rev):
            cin, cout = cout, c
            self.up_blocks.append(UpBlockTemporalDecoder(cin, cout, layers_per_block + 1, add_upsample=i != len(ch) - 1))
        self.conv_norm_out = nn.GroupNorm(32, ch[0], eps=1e-6)
        self.conv_out = nn.Conv2d(ch[0], out_channels, 3, padding=1)
        self.time_conv_out = nn.Conv3d(out_channels, out_channels, (3, 1, 1), padding=(1, 0, 0))

    def forward(self, z, num_frames: int):                 # [b*t, latent, h, w] -> [b*t, out, 8h, 8w]
        x = self.conv_in(z)
        x = self.mid_block(x, num_frames)
        for blk in self.up_blocks:
            x = blk(x, num_frames)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        bt, c, h, w = x.shape
        x5 = x.reshape(bt // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        x5 = self.time_conv_out(x5)
        return x5.permute(0, 2, 1, 3, 4).reshape(bt, c, h, w)


class VaeOracle(VaeEncoderOracle):
    """The whole `AutoencoderKLTemporalDecoder` under diffusers' key names (`encoder.*`, `decoder.*`, `quant_conv.*`): `decode(z,
    num_frames)` as the class does it (image_only_indicator is all-zero and unused by the "learned" blender)."""

    def __init__(self, in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 scaling_factor=0.18215):
        super().__init__(in_channels, latent_channels, block_out_channels, layers_per_block, scaling_factor)
        self.decoder = TemporalDecoder(latent_channels, out_channels, block_out_channels, layers_per_block)

    def decode(self, z, num_frames: int):
        return self.decoder(z, num_frames)


def decode_latents(latents: torch.Tensor, vae: VaeOracle, num_frames: int, decode_chunk_size: int = 14) -> torch.Tensor:
    """StableVideoDiffusionPipeline.decode_latents: [b, f, 4, h, w] latents -> [b, 3, f, 8h, 8w] float frames; the decoder sees
    chunks of `decode_chunk_size` frames, each chunk as ONE clip of that many frames (that is what the pipeline does)."""
    b = latents.shape[0]
    latents = latents.flatten(0, 1) / vae.scaling_factor
    frames = []
    for i in range(0, latents.shape[0], decode_chunk_size):
        chunk = latents[i:i + decode_chunk_size]
        frames.append(vae.decode(chunk, chunk.shape[0]))
    frames = torch.cat(frames, dim=0)
    return frames.reshape(b, num_frames, *frames.shape[1:]).permute(0, 2, 1, 3, 4).float()
