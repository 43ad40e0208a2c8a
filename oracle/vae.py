"""CPU oracle for the step that precedes the UNet in the reference loop: `tensor_to_vae_latent`
(/root/reference/train_svd.py:283-291, called at :948 and :959-960) = the ENCODER half of diffusers'
`AutoencoderKLTemporalDecoder` + `DiagonalGaussianDistribution.sample()` x `scaling_factor`.

TEST INFRASTRUCTURE ONLY (see oracle/unet.py header).  GROUNDWORK for SURVEY.md section 8(f) rank 1: there is no product
counterpart in svd_xtend_amd/ yet.  PARITY UNPINNED: diffusers is not installed here; the encoder is restated from the
published module (diffusers.models.autoencoders.vae.Encoder with DownEncoderBlock2D x4, UNetMidBlock2D with one single-head
attention, `quant_conv`), self-pinned only structurally (tests/test_oracle_vae.py): the diffusers state-dict key set and
the parameter count of the SD / SVD VAE encoder (34,163,592 + 72 for quant_conv).

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
