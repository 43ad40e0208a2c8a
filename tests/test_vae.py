"""the VAE-encode step before the UNet (`tensor_to_vae_latent`, /root/reference/train_svd.py:283-291, :948,
:957-960) on the HIP path (svd_xtend_amd/vae.py) against the oracle's restatement of diffusers' encoder (oracle/vae.py).
CPU tests run the host orchestration on the fp32 emulation of the C-ABI; `-m gpu` tests run the real kernels."""
import pytest
import torch

from oracle.vae import SVD_VAE_CONFIG, VaeEncoderOracle
from svd_xtend_amd.vae import AutoencoderKLTemporalDecoder, tensor_to_vae_latent

gpu = pytest.mark.gpu
SMALL = dict(in_channels=3, latent_channels=4, block_out_channels=(64, 128, 128, 128), layers_per_block=1, scaling_factor=0.18215)


def make_pair(cfg, seed, dev="cpu"):
    orc = VaeEncoderOracle(**cfg)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in orc.named_parameters():
            if p.ndim == 1:
                p.copy_((1.0 if "norm" in n and n.endswith("weight") else 0.0) + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(torch.randn(p.shape, generator=g) * (1.0 / p[0].numel()) ** 0.5)
    vae = AutoencoderKLTemporalDecoder(**cfg)
    missing = vae.load_state_dict(orc.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return orc, vae.to(dev)


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def test_state_dict_keys_are_diffusers_encoder_keys():
    orc = VaeEncoderOracle(**SMALL)
    vae = AutoencoderKLTemporalDecoder(**SMALL)
    assert list(vae.state_dict().keys()) == list(orc.state_dict().keys())
    assert sum(p.numel() for p in AutoencoderKLTemporalDecoder(**SVD_VAE_CONFIG).parameters()) == 34_163_592 + 72


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
    sd["decoder.conv_in.weight"] = torch.zeros(4, 4, 3, 3, dtype=torch.float16)            # the decoder half is skipped
    save_file(sd, str(folder / "diffusion_pytorch_model.fp16.safetensors"))
    (folder / "config.json").write_text(json.dumps({"_class_name": "AutoencoderKLTemporalDecoder", "force_upcast": True, **SMALL}))
    v2 = AutoencoderKLTemporalDecoder.from_pretrained(str(tmp_path), subfolder="vae", variant="fp16")
    assert v2.config.scaling_factor == 0.18215 and v2.config.force_upcast is True
    for k, v in v2.state_dict().items():
        assert torch.equal(v, sd[k].float()), k


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
