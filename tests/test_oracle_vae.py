"""Structural pins of oracle/vae.py (groundwork for SURVEY.md 8(f) rank 1, the VAE-encode step in front of the UNet): no diffusers
here, so the restated encoder is checked against what is known of the published module -- its parameter count, its state-dict
keys, and each non-standard piece against an independent formulation."""
import torch
import torch.nn.functional as F

from oracle.vae import SVD_VAE_CONFIG, TINY_VAE_CONFIG, Attention, Downsample2D, VaeEncoderOracle, VaeOracle, tensor_to_vae_latent


def test_svd_vae_encoder_parameter_count_and_keys():
    with torch.device("meta"):
        m = VaeEncoderOracle(**SVD_VAE_CONFIG)
    assert sum(p.numel() for p in m.encoder.parameters()) == 34_163_592          # encoder of the SD / SVD AutoencoderKL
    assert sum(p.numel() for p in m.quant_conv.parameters()) == 72
    keys = set(m.state_dict())
    for k in ("encoder.conv_in.weight", "encoder.down_blocks.0.resnets.1.norm2.bias", "encoder.down_blocks.1.resnets.0.conv_shortcut.weight",
              "encoder.down_blocks.2.downsamplers.0.conv.weight", "encoder.mid_block.attentions.0.group_norm.weight",
              "encoder.mid_block.attentions.0.to_out.0.bias", "encoder.mid_block.resnets.1.conv2.weight", "encoder.conv_norm_out.weight",
              "encoder.conv_out.bias", "quant_conv.weight"):
        assert k in keys, k
    assert not any("downsamplers" in k for k in keys if k.startswith("encoder.down_blocks.3."))   # the last block keeps its resolution
    assert m.state_dict()["encoder.conv_out.weight"].shape == (8, 512, 3, 3)                      # double_z: mean and logvar


def test_downsample_is_bottom_right_padded_stride_2():
    torch.manual_seed(0)
    d = Downsample2D(8)
    x = torch.randn(2, 8, 9, 12)
    y = d(x)
    assert y.shape == (2, 8, 4, 6)                       # floor((H + 1 - 3) / 2) + 1
    # the same numbers from the unfold definition: out[y, x] = sum_{dy,dx} w[dy,dx] . in[2y + dy, 2x + dx], zero outside
    xp = torch.zeros(2, 8, 10, 13)
    xp[:, :, :9, :12] = x
    cols = F.unfold(xp, 3, stride=2)                     # [2, 8*9, 4*6]
    ref = (d.conv.weight.view(8, -1) @ cols).view(2, 8, 4, 6) + d.conv.bias.view(1, 8, 1, 1)
    assert torch.allclose(y, ref, atol=1e-5)
    sym = F.conv2d(x, d.conv.weight, d.conv.bias, stride=2, padding=1)
    assert sym.shape != y.shape or not torch.allclose(sym, y, atol=1e-3)      # NOT the UNet's symmetric pad-1 downsample


def test_mid_block_attention_is_single_head_sdpa_with_residual():
    torch.manual_seed(1)
    a = Attention(64)
    x = torch.randn(2, 64, 5, 7)
    y = a(x)
    n = a.group_norm(x.view(2, 64, 35)).transpose(1, 2)
    o = F.scaled_dot_product_attention(a.to_q(n)[:, None], a.to_k(n)[:, None], a.to_v(n)[:, None])[:, 0]      # one head of dim 64
    ref = a.to_out[0](o).transpose(1, 2).reshape(2, 64, 5, 7) + x
    assert torch.allclose(y, ref, atol=1e-5)


def test_tensor_to_vae_latent_shapes_scaling_and_sampling():
    torch.manual_seed(2)
    vae = VaeEncoderOracle(**TINY_VAE_CONFIG)
    t = torch.rand(2, 3, 3, 64, 48) * 2 - 1
    z1 = tensor_to_vae_latent(t, vae, torch.Generator().manual_seed(5))
    z2 = tensor_to_vae_latent(t, vae, torch.Generator().manual_seed(5))
    assert z1.shape == (2, 3, 4, 8, 6) and torch.equal(z1, z2)
    mean, logvar = vae.moments(t.reshape(6, 3, 64, 48))
    assert float(logvar.max()) <= 20.0 and float(logvar.min()) >= -30.0
    eps = torch.randn(mean.shape, generator=torch.Generator().manual_seed(5))
    ref = (mean + torch.exp(0.5 * logvar) * eps).reshape(2, 3, 4, 8, 6) * 0.18215
    assert torch.allclose(z1, ref, atol=1e-6)
    # frames are encoded independently (the temporal part of this VAE is its decoder only)
    z_first = tensor_to_vae_latent(t[:, :1], vae, torch.Generator().manual_seed(5))
    m_all = vae.moments(t.reshape(6, 3, 64, 48))[0].reshape(2, 3, 4, 8, 6)
    m_first = vae.moments(t[:, 0])[0]
    assert torch.allclose(m_all[:, 0], m_first, atol=1e-5) and z_first.shape == (2, 1, 4, 8, 6)
