"""the VAE-encode step before the UNet (`tensor_to_vae_latent`, /root/reference/train_svd.py:283-291, :948,
:957-960) on the HIP path (svd_xtend_amd/vae.py) against the oracle's restatement of diffusers' encoder (oracle/vae.py); and the
temporal decoder behind the validation sampler (8(f) rank 4, train_svd.py:1106-1137) against the oracle's TemporalDecoder.
CPU tests run the host orchestration on the fp32 emulation of the C-ABI; `-m gpu` tests run the real kernels."""
import pytest
import torch

from oracle.vae import SVD_VAE_CONFIG, VaeOracle, decode_latents
from svd_xtend_amd.vae import AutoencoderKLTemporalDecoder, tensor_to_vae_latent

gpu = pytest.mark.gpu
SMALL = dict(in_channels=3, latent_channels=4, block_out_channels=(64, 128, 128, 128), layers_per_block=1, scaling_factor=0.18215)


def make_pair(cfg, seed, dev="cpu"):
    orc = VaeOracle(**cfg)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in orc.named_parameters():
            if n.endswith("mix_factor"):
                p.copy_(torch.randn(p.shape, generator=g))            # sigmoid(mix) away from the 0.5 of a fresh blender
            elif p.ndim == 1:
                p.copy_((1.0 if "norm" in n and n.endswith("weight") else 0.0) + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / p[0].numel()) ** 0.5)
    vae = AutoencoderKLTemporalDecoder(**cfg)
    missing = vae.load_state_dict(orc.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return orc, vae.to(dev)


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def test_state_dict_keys_are_diffusers_keys():
    orc = VaeOracle(**SMALL)
    vae = AutoencoderKLTemporalDecoder(**SMALL)
    assert sorted(vae.state_dict().keys()) == sorted(orc.state_dict().keys())
    assert {k: tuple(v.shape) for k, v in vae.state_dict().items()} == {k: tuple(v.shape) for k, v in orc.state_dict().items()}
    full = AutoencoderKLTemporalDecoder(**SVD_VAE_CONFIG)
    assert sum(p.numel() for n, p in full.named_parameters() if not n.startswith("decoder.")) == 34_163_592 + 72
    assert sum(p.numel() for p in full.parameters()) == 97_742_847            # the SVD `vae/` checkpoint


@pytest.mark.parametrize("shape", [(2, 32, 48), (1, 40, 24)])       # 40x24 -> 5x3 = 15 tokens: padded attention reduction
def test_encoder_matches_oracle_on_the_emulated_kernels(emu_backend, shape):
    n, H, W = shape
    orc, vae = make_pair(SMALL, 3)
    vae.prepare(torch.float32)
    x = torch.rand(n, 3, H, W, generator=torch.Generator().manual_seed(4)) * 2 - 1
    with torch.no_grad():
        mean, logvar = orc.moments(x)
    d = vae.encode(x).latent_dist
    assert d.mean.shape == (n, 4, H // 8, W // 8)
    assert rel(d.mean, mean) <= 2e-5 and rel(d.logvar, logvar) <= 2e-5, (rel(d.mean, mean), rel(d.logvar, logvar))
    # sample() = mean + std * randn on the moments' device generator; tensor_to_vae_latent scales and restores [b, f, ...]
    g1, g2 = torch.Generator().manual_seed(7), torch.Generator().manual_seed(7)
    z = tensor_to_vae_latent(x[None], vae, g1)
    eps = torch.randn(mean.shape, generator=g2)
    want = (mean + torch.exp(0.5 * logvar) * eps) * 0.18215
    assert z.shape == (1, n, 4, H // 8, W // 8) and rel(z[0], want) <= 2e-5


def test_frames_are_chunked_below_the_buffer_limit(emu_backend):
    orc, vae = make_pair(SMALL, 5)
    vae.prepare(torch.float32)
    assert AutoencoderKLTemporalDecoder(**SVD_VAE_CONFIG).max_frames(320, 512) >= 15            # c2: one chunk
    assert 1 <= AutoencoderKLTemporalDecoder(**SVD_VAE_CONFIG).max_frames(576, 1024) < 26       # c4: several
    x = torch.rand(3, 3, 16, 16, generator=torch.Generator().manual_seed(6)) * 2 - 1
    whole = vae.encode(x).latent_dist.mean.clone()
    vae.max_frames = lambda H, W: 2
    assert torch.allclose(vae.encode(x).latent_dist.mean, whole, atol=1e-6)


def test_from_pretrained_reads_a_diffusers_vae_folder(tmp_path, emu_backend):
    import json

    from safetensors.torch import save_file
    orc, vae = make_pair(SMALL, 8)
    folder = tmp_path / "vae"
    folder.mkdir()
    sd = {k: v.half().contiguous() for k, v in orc.state_dict().items()}
    save_file(sd, str(folder / "diffusion_pytorch_model.fp16.safetensors"))
    (folder / "config.json").write_text(json.dumps({"_class_name": "AutoencoderKLTemporalDecoder", "force_upcast": True, **SMALL}))
    v2 = AutoencoderKLTemporalDecoder.from_pretrained(str(tmp_path), subfolder="vae", variant="fp16")
    assert v2.config.scaling_factor == 0.18215 and v2.config.force_upcast is True
    for k, v in v2.state_dict().items():
        assert torch.equal(v, sd[k].float()), k


@pytest.mark.parametrize("shape", [(1, 3, 4, 6), (2, 2, 5, 3)])     # (clips, frames, latent h, w)
def test_decoder_matches_oracle_on_the_emulated_kernels(emu_backend, shape):
    B, T, h, w = shape
    orc, vae = make_pair(SMALL, 11)
    vae.prepare(torch.float32)
    z = torch.randn(B * T, 4, h, w, generator=torch.Generator().manual_seed(12))
    with torch.no_grad():
        want = orc.decode(z, T)
    got = vae.decode(z, num_frames=T).sample
    assert got.shape == (B * T, 3, 8 * h, 8 * w)
    assert rel(got, want) <= 3e-5, rel(got, want)
    # frames of different clips do not mix; frames of one clip do (the temporal convolutions)
    if B == 2:
        z2 = z.clone()
        z2[T:] = torch.randn(T, 4, h, w, generator=torch.Generator().manual_seed(13))
        assert torch.allclose(vae.decode(z2, num_frames=T).sample[:T], got[:T], atol=1e-6)


def test_decode_latents_chunks_like_the_pipeline(emu_backend):
    """StableVideoDiffusionPipeline.decode_latents: 1 / scaling_factor, chunks of decode_chunk_size frames each decoded as one clip."""
    from svd_xtend_amd.pipeline import decode_latents as product_decode_latents
    orc, vae = make_pair(SMALL, 14)
    vae.prepare(torch.float32)
    lat = torch.randn(1, 5, 4, 3, 4, generator=torch.Generator().manual_seed(15)) * 0.18215
    with torch.no_grad():
        want = decode_latents(lat, orc, 5, decode_chunk_size=2)
    got = product_decode_latents(lat, vae, 5, decode_chunk_size=2)
    assert got.shape == (1, 3, 5, 24, 32) and rel(got, want) <= 3e-5
    with pytest.raises(ValueError):
        vae.decode(torch.zeros(3, 4, 3, 4), num_frames=2)


@gpu
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_encoder_matches_oracle_small(dt):
    dev = torch.device("cuda")
    orc, vae = make_pair(SMALL, 3, dev)
    vae.prepare(dt)
    for (n, H, W) in [(2, 32, 48), (1, 40, 24), (3, 64, 64)]:
        x = torch.rand(n, 3, H, W, generator=torch.Generator().manual_seed(4)) * 2 - 1
        with torch.no_grad():
            mean, logvar = orc.moments(x)
        d = vae.encode(x.to(dev)).latent_dist
        tol = 1e-2 if dt == torch.float16 else 6e-2
        assert rel(d.mean.cpu(), mean) <= tol and rel(d.logvar.cpu(), logvar) <= tol, (n, H, W, rel(d.mean.cpu(), mean), rel(d.logvar.cpu(), logvar))


@gpu
def test_encoder_matches_oracle_at_the_svd_widths():
    """The real encoder (128 / 256 / 512 / 512 channels, 34.2 M parameters) on two 512x320 frames -- the c2 frame size, so every
    GEMM has the benched N / K and the attention its 2560 tokens -- fp16 against the fp32 oracle."""
    dev = torch.device("cuda")
    orc, vae = make_pair(SVD_VAE_CONFIG, 9, dev)
    vae.prepare(torch.float16)
    x = torch.rand(2, 3, 320, 512, generator=torch.Generator().manual_seed(10)) * 2 - 1
    with torch.no_grad():
        mean, logvar = orc.moments(x)
    d = vae.encode(x.to(dev)).latent_dist
    r = (rel(d.mean.cpu(), mean), rel(d.logvar.cpu(), logvar))
    print("vae 512x320 fp16 rel-L2 (mean, logvar):", r)
    assert r[0] <= 1e-2 and r[1] <= 1e-2, r


@gpu
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_decoder_matches_oracle_small(dt):
    dev = torch.device("cuda")
    orc, vae = make_pair(SMALL, 11, dev)
    vae.prepare(dt)
    for (B, T, h, w) in [(1, 3, 4, 6), (2, 2, 5, 3), (1, 8, 8, 8)]:
        z = torch.randn(B * T, 4, h, w, generator=torch.Generator().manual_seed(12))
        with torch.no_grad():
            want = orc.decode(z, T)
        got = vae.decode(z.to(dev), num_frames=T).sample.cpu()
        tol = 1e-2 if dt == torch.float16 else 6e-2
        assert rel(got, want) <= tol, (B, T, h, w, rel(got, want))


@gpu
def test_decoder_matches_oracle_at_the_svd_widths():
    """The real decoder (512 / 512 / 256 / 128 channels, 63.6 M parameters) on a 3-frame clip of 16x24 latents (128x192 pixels):
    every GEMM has the real N / K, the temporal layers their three taps; fp16 against the fp32 oracle."""
    dev = torch.device("cuda")
    orc, vae = make_pair(SVD_VAE_CONFIG, 16, dev)
    vae.prepare(torch.float16)
    z = torch.randn(3, 4, 16, 24, generator=torch.Generator().manual_seed(17))
    with torch.no_grad():
        want = orc.decode(z, 3)
    got = vae.decode(z.to(dev), num_frames=3).sample.cpu()
    print("vae decoder 128x192 x3 fp16 rel-L2:", rel(got, want))
    assert rel(got, want) <= 1e-2


@gpu
def test_decode_chunk_of_8_at_the_reference_validation_resolution():
    """/root/reference/train_svd.py:1130-1138 validates at 1024 x 576 with decode_chunk_size = 8: eight frames of 72x128 latents put
    2.4 GB activations in front of the last up block, beyond the 2 GiB reach of the buffer-addressed GEMM kernel (max_decode_frames =
    7).  The chunk must decode through the 64-bit-pointer kernel -- and that kernel must agree with the fast one on this decoder's op
    mix, checked at a size both can run (the fp16 results of the two kernels differ only by accumulation order)."""
    dev = torch.device("cuda")
    _, vae = make_pair(SVD_VAE_CONFIG, 23, dev)
    vae.prepare(torch.float16)
    assert vae.max_decode_frames(72, 128) < 8
    z = torch.randn(8, 4, 72, 128, generator=torch.Generator().manual_seed(5)).to(dev)
    out = vae.decode(z, num_frames=8).sample
    assert out.shape == (8, 3, 576, 1024) and bool(torch.isfinite(out).all())
    zs = z[:3, :, :16, :24].contiguous()
    fast = vae.decode(zs, num_frames=3).sample
    vae.rt.gemm_variant = 1
    try:
        slow = vae.decode(zs, num_frames=3).sample
    finally:
        vae.rt.gemm_variant = 4
    assert rel(slow.cpu(), fast.cpu()) <= 2e-3, rel(slow.cpu(), fast.cpu())
