"""SURVEY.md 8(c) check (5): the day `diffusers` is importable (it is not in this image, and cannot be installed offline), the
oracle's blocks below the UNet top level stop being "parity unpinned": diffusers' own UNetSpatioTemporalConditionModel, the class
the reference trains (/root/reference/train_svd.py:49, :651-656), loads the oracle's state dict strictly and must reproduce its
output and gradients in fp32.  Skipped while the package is absent."""
import pytest
import torch


def test_oracle_matches_diffusers_unet():
    diffusers = pytest.importorskip("diffusers")
    from oracle.step import edm_inputs, edm_loss, make_synthetic_batch
    from oracle.unet import TINY_CONFIG, UNetSpatioTemporalConditionOracle, scaled_init_
    cfg = dict(TINY_CONFIG)
    orc = UNetSpatioTemporalConditionOracle(**cfg)
    scaled_init_(orc, 21)
    ref = diffusers.UNetSpatioTemporalConditionModel(**cfg)
    missing, unexpected = ref.load_state_dict(orc.state_dict(), strict=True)
    assert not missing and not unexpected
    batch = make_synthetic_batch(2, 3, 16, 24, 22, cross_dim=cfg["cross_attention_dim"])     # B = 2: the HW-major time_context quirk
    unet_in, ts, ehs, ids, noisy, sig = edm_inputs(batch)
    outs = []
    for m in (orc, ref):
        for n, p in m.named_parameters():
            p.requires_grad_("temporal_transformer_block" in n)                             # train_svd.py:761-766
            p.grad = None
        pred = m(unet_in, ts, ehs, added_time_ids=ids).sample
        edm_loss(pred, noisy, batch["latents"], sig).backward()
        outs.append((pred.detach(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    (p0, g0), (p1, g1) = outs
    print("diffusers", diffusers.__version__, "max |dpred|", float((p0 - p1).abs().max()))
    assert torch.allclose(p0, p1, atol=1e-5, rtol=1e-5), float((p0 - p1).abs().max())
    assert g0.keys() == g1.keys()
    for n in g0:
        assert torch.allclose(g0[n], g1[n], atol=1e-5, rtol=1e-4), (n, float((g0[n] - g1[n]).abs().max()))


def test_oracle_vae_encoder_matches_diffusers():
    diffusers = pytest.importorskip("diffusers")
    from oracle.vae import VaeEncoderOracle
    orc = VaeEncoderOracle(block_out_channels=(32, 64, 64, 64))
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in orc.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.ndim > 1 else 0.02) + (1.0 if p.ndim == 1 and p.numel() > 8 else 0.0))
    ref = diffusers.AutoencoderKLTemporalDecoder(block_out_channels=(32, 64, 64, 64), layers_per_block=2, latent_channels=4)
    sd = {k: v for k, v in ref.state_dict().items() if k.startswith(("encoder.", "quant_conv."))}
    assert sd.keys() == orc.state_dict().keys()
    ref.load_state_dict({**ref.state_dict(), **orc.state_dict()}, strict=True)
    x = torch.rand(2, 3, 32, 48, generator=g) * 2 - 1
    with torch.no_grad():
        d = ref.encode(x).latent_dist
        mean, logvar = orc.moments(x)
    assert torch.allclose(mean, d.mean, atol=1e-5, rtol=1e-5), float((mean - d.mean).abs().max())
    assert torch.allclose(logvar, d.logvar, atol=1e-5, rtol=1e-5), float((logvar - d.logvar).abs().max())
