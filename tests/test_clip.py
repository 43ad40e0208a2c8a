"""the image-conditioning path of `encode_image` (/root/reference/train_svd.py:857-876) on the HIP path
(svd_xtend_amd/clip.py).  Oracles: `transformers.CLIPVisionModelWithProjection` -- the very class the reference instantiates
(train_svd.py:646-648), installed here -- for the tower, and oracle/clip_image.py (pinned bit for bit to the reference's own resize
functions, tests/golden/resize_antialias.safetensors) for the resize + normalisation.  CPU tests drive the host orchestration over
the fp32 emulation of the C-ABI; `-m gpu` tests run the kernels."""
import pytest
import torch

from oracle.clip_image import clip_pixel_values as oracle_pixels
from svd_xtend_amd import clip as C

gpu = pytest.mark.gpu
SMALL = dict(hidden_size=320, intermediate_size=640, projection_dim=64, num_hidden_layers=2, num_attention_heads=4, image_size=56,
             patch_size=14, hidden_act="gelu")              # head dimension 80, as ViT-H/14


def make_pair(cfg, seed, dev="cpu"):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.manual_seed(seed)
    ref = CLIPVisionModelWithProjection(CLIPVisionConfig(**cfg)).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():                  # transformers' init is tiny (std 0.02): give every block a visible effect
            if p.ndim >= 2:
                p.copy_(torch.randn(p.shape) * (1.5 / p[0].numel()) ** 0.5)
            elif "norm" in n and n.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape))
            else:
                p.copy_(0.1 * torch.randn(p.shape))
    mine = C.CLIPVisionModelWithProjection(**cfg)
    sd = {k: v for k, v in ref.state_dict().items() if not k.endswith("position_ids")}
    assert list(mine.state_dict().keys()) == list(sd.keys())
    mine.load_state_dict(sd, strict=True)
    return ref, mine.to(dev)


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def test_state_dict_keys_match_transformers_at_the_svd_config():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    with torch.device("meta"):
        ref = CLIPVisionModelWithProjection(CLIPVisionConfig(hidden_size=1280, intermediate_size=5120, projection_dim=1024,
                                                             num_hidden_layers=32, num_attention_heads=16, image_size=224, patch_size=14,
                                                             hidden_act="gelu"))
        mine = C.CLIPVisionModelWithProjection()
    want = {k: tuple(v.shape) for k, v in ref.state_dict().items() if not k.endswith("position_ids")}
    got = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    assert got == want
    assert sum(v.numel() for v in mine.state_dict().values()) == 632_076_800          # ViT-H/14 vision tower + projection


@pytest.mark.parametrize("act", ["gelu", "quick_gelu"])
def test_tower_matches_transformers_on_the_emulated_kernels(emu_backend, act):
    ref, mine = make_pair(dict(SMALL, hidden_act=act), 3)
    mine.prepare(torch.float32)
    x = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        want = ref(pixel_values=x)
    got = mine(x)
    assert got.image_embeds.shape == (2, 64)
    assert rel(got.last_hidden_state, want.last_hidden_state) <= 2e-5, rel(got.last_hidden_state, want.last_hidden_state)
    assert rel(got.image_embeds, want.image_embeds) <= 2e-5, rel(got.image_embeds, want.image_embeds)


@pytest.mark.parametrize("shape,size", [((1, 3, 320, 512), (224, 224)), ((2, 3, 37, 53), (28, 28)), ((1, 3, 64, 40), (56, 56))])
def test_resize_and_normalisation_match_the_pinned_oracle(emu_backend, shape, size):
    from svd_xtend_amd import kernels as K
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(5)) * 2 - 1
    got = C.clip_pixel_values(x, size, K.backend())
    want = oracle_pixels(x, size)
    assert got.shape == want.shape and float((got - want).abs().max()) <= 2e-5, float((got - want).abs().max())


def test_encode_image_end_to_end_on_the_emulated_kernels(emu_backend):
    from oracle.clip_image import encode_image as oracle_encode
    ref, mine = make_pair(SMALL, 6)
    mine.prepare(torch.float32)
    frames = torch.rand(2, 3, 96, 160, generator=torch.Generator().manual_seed(7)) * 2 - 1
    with torch.no_grad():
        want = ref(pixel_values=oracle_pixels(frames, (56, 56))).image_embeds
    got = C.encode_image(frames, mine)
    assert rel(got, want) <= 5e-5, rel(got, want)
    assert oracle_encode is not None


def test_from_pretrained_reads_a_transformers_folder(tmp_path, emu_backend):
    import json

    from safetensors.torch import save_file
    ref, mine = make_pair(SMALL, 8)
    folder = tmp_path / "image_encoder"
    folder.mkdir()
    sd = {k: v.half().contiguous() for k, v in ref.state_dict().items()}
    save_file(sd, str(folder / "model.fp16.safetensors"))
    (folder / "config.json").write_text(json.dumps({"architectures": ["CLIPVisionModelWithProjection"], "model_type": "clip_vision_model", **SMALL}))
    m2 = C.CLIPVisionModelWithProjection.from_pretrained(str(tmp_path), subfolder="image_encoder", variant="fp16")
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k].float()), k


@gpu
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_encode_image_matches_oracle_small(dt):
    dev = torch.device("cuda")
    ref, mine = make_pair(SMALL, 6, dev)
    mine.prepare(dt)
    frames = torch.rand(2, 3, 96, 160, generator=torch.Generator().manual_seed(7)) * 2 - 1
    px = oracle_pixels(frames, (56, 56))
    got_px = C.clip_pixel_values(frames.to(dev), (56, 56))
    assert float((got_px.cpu() - px).abs().max()) <= 2e-5
    with torch.no_grad():
        want = ref(pixel_values=px).image_embeds
    got = C.encode_image(frames.to(dev), mine).cpu()
    tol = 1e-2 if dt == torch.float16 else 6e-2
    assert rel(got, want) <= tol, rel(got, want)


@gpu
def test_encode_image_matches_transformers_at_vit_h():
    """The SVD image encoder's own configuration (ViT-H/14: 32 layers, 1280 wide, 16 heads of 80, 257 tokens; 632 M parameters,
    random weights) on a 512x320 frame, fp16 against transformers' fp32 forward."""
    dev = torch.device("cuda")
    cfg = dict(hidden_size=1280, intermediate_size=5120, projection_dim=1024, num_hidden_layers=32, num_attention_heads=16,
               image_size=224, patch_size=14, hidden_act="gelu")
    ref, mine = make_pair(cfg, 9, dev)
    mine.prepare(torch.float16)
    frames = torch.rand(1, 3, 320, 512, generator=torch.Generator().manual_seed(10)) * 2 - 1
    px = oracle_pixels(frames, (224, 224))
    with torch.no_grad():
        want = ref(pixel_values=px).image_embeds
    got = C.encode_image(frames.to(dev), mine).cpu()
    r = rel(got, want)
    print("ViT-H/14 fp16 image_embeds rel-L2:", r)
    assert r <= 2e-2, r
