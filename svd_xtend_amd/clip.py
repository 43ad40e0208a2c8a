"""MI355X-native image-conditioning path in front of the UNet: `encode_image` of /root/reference/train_svd.py:857-876 --
`_resize_with_antialiasing` (:140-248) to 224 x 224, back to [0, 1], CLIP normalisation (the `feature_extractor` call of :864-871
with do_resize / do_center_crop / do_rescale off), `CLIPVisionModelWithProjection(...).image_embeds` (:642-648, :875).
SURVEY.md 8(f) rank 2.  The tower is frozen (:660) and runs once per clip on its first frame (:975-976).

`CLIPVisionModelWithProjection` here has the transformers class's config fields, state-dict keys (`vision_model.embeddings.*`,
`vision_model.encoder.layers.N.*`, `visual_projection.weight`; the misspelt `pre_layrnorm` included) and call contract
(`model(pixel_values).image_embeds`), and loads an `image_encoder/` folder of an SVD checkpoint.  The arithmetic is libsvdx launches:

  resize            svdx_blur_axis (x pass, y pass; reflect padding) + svdx_bicubic_affine (align_corners bicubic, then the
                    (v + 1) / 2 un-normalisation and CLIP's mean / std as one per-channel affine map)
  patch embedding   svdx_patch_rows (14 x 14 patches as GEMM rows, K = 588 padded to 640) + GEMM whose residual operand is the
                    position embedding; the class token row is a constant
  encoder layer     svdx_ln_fwd, fused q/k/v GEMM (+bias), svdx_attn_small_fwd (heads of 80 channels over 257 tokens), out-projection
                    GEMM (+bias +residual), svdx_ln_fwd, fc1 GEMM, svdx_act_rows (gelu / quick_gelu), fc2 GEMM (+bias +residual)
  head              post_layernorm of the class token, visual_projection GEMM
Forward only."""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Optional, Sequence

import torch
import torch.nn as nn

from .ops import LayerNormOp, LinearOp, Runtime, gemm_act, rup
from .unet import FrozenConfig

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)        # CLIPImageProcessor defaults (train_svd.py:640-641, :864-871)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
_ACTS = {"gelu": 0, "quick_gelu": 1}


class _Embeddings(nn.Module):
    def __init__(self, hidden, patch, n_pos, channels):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(hidden))
        self.patch_embedding = nn.Conv2d(channels, hidden, patch, stride=patch, bias=False)
        self.position_embedding = nn.Embedding(n_pos, hidden)


class _Attn(nn.Module):
    def __init__(self, hidden):
        super().__init__()
        self.k_proj, self.v_proj, self.q_proj = nn.Linear(hidden, hidden), nn.Linear(hidden, hidden), nn.Linear(hidden, hidden)
        self.out_proj = nn.Linear(hidden, hidden)


class _MLP(nn.Module):
    def __init__(self, hidden, inter):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(hidden, inter), nn.Linear(inter, hidden)


class _Layer(nn.Module):
    def __init__(self, hidden, inter, eps):
        super().__init__()
        self.self_attn = _Attn(hidden)
        self.layer_norm1 = nn.LayerNorm(hidden, eps=eps)
        self.mlp = _MLP(hidden, inter)
        self.layer_norm2 = nn.LayerNorm(hidden, eps=eps)

    def build(self, rt: Runtime) -> None:
        a = self.self_attn
        self.ln1, self.ln2 = LayerNormOp(self.layer_norm1), LayerNormOp(self.layer_norm2)
        self.qkv = LinearOp([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], [a.q_proj.bias, a.k_proj.bias, a.v_proj.bias])
        self.o = LinearOp([a.out_proj.weight], [a.out_proj.bias])
        self.f1 = LinearOp([self.mlp.fc1.weight], [self.mlp.fc1.bias])
        self.f2 = LinearOp([self.mlp.fc2.weight], [self.mlp.fc2.bias])
        for op in (self.qkv, self.o, self.f1, self.f2):
            op.pack(rt, need_dx=False)


class _Encoder(nn.Module):
    def __init__(self, n, hidden, inter, eps):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(hidden, inter, eps) for _ in range(n)])


class _VisionTransformer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        n_pos = (cfg.image_size // cfg.patch_size) ** 2 + 1
        self.embeddings = _Embeddings(cfg.hidden_size, cfg.patch_size, n_pos, cfg.num_channels)
        self.pre_layrnorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)         # sic: the transformers attribute name
        self.encoder = _Encoder(cfg.num_hidden_layers, cfg.hidden_size, cfg.intermediate_size, cfg.layer_norm_eps)
        self.post_layernorm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class CLIPVisionModelWithProjection(nn.Module):
    """Config fields of transformers.CLIPVisionConfig; defaults = the SVD `image_encoder` (OpenCLIP ViT-H/14)."""

    def __init__(self, hidden_size: int = 1280, intermediate_size: int = 5120, projection_dim: int = 1024, num_hidden_layers: int = 32,
                 num_attention_heads: int = 16, num_channels: int = 3, image_size: int = 224, patch_size: int = 14,
                 hidden_act: str = "gelu", layer_norm_eps: float = 1e-5, **other):
        super().__init__()
        if hidden_act not in _ACTS:
            raise ValueError(f"hidden_act {hidden_act!r} unsupported (have {sorted(_ACTS)})")
        d = hidden_size // num_attention_heads
        if hidden_size % 64 or intermediate_size % 64 or d * num_attention_heads != hidden_size or d % 8 or d > 128:
            raise ValueError("hidden / intermediate sizes must be multiples of 64, the head dimension a multiple of 8 up to 128")
        self.config = FrozenConfig(hidden_size=hidden_size, intermediate_size=intermediate_size, projection_dim=projection_dim,
                                   num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads, num_channels=num_channels,
                                   image_size=image_size, patch_size=patch_size, hidden_act=hidden_act, layer_norm_eps=layer_norm_eps, **other)
        self.vision_model = _VisionTransformer(self.config)
        self.visual_projection = nn.Linear(hidden_size, projection_dim, bias=False)
        self.rt: Optional[Runtime] = None
        self._requested_dtype = None

    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = None, variant: Optional[str] = None, torch_dtype=None, **unused):
        """`<path>/<subfolder>/config.json` + `model[.<variant>].safetensors` (train_svd.py:646-648)."""
        from safetensors.torch import load_file
        folder = os.path.join(path, subfolder) if subfolder else path
        with open(os.path.join(folder, "config.json")) as f:
            raw = json.load(f)
        cfg = {k: v for k, v in {**raw.get("vision_config", {}), **raw}.items() if not k.startswith("_") and k not in ("vision_config", "text_config")}
        model = cls(**cfg)
        names = [f"model.{variant}.safetensors"] if variant else []
        wpath = next((os.path.join(folder, n) for n in names + ["model.safetensors"] if os.path.exists(os.path.join(folder, n))), None)
        if wpath is None:
            raise FileNotFoundError(f"no model[.{variant}].safetensors under {folder}")
        sd = {k: v.float() for k, v in load_file(wpath).items() if k != "vision_model.embeddings.position_ids"}
        model.load_state_dict(sd, strict=True)
        if torch_dtype is not None and torch_dtype != torch.float32:
            model._requested_dtype = torch_dtype
        return model

    def to(self, *args, **kwargs):
        device, dtype, non_blocking, _ = torch._C._nn._parse_to(*args, **kwargs)
        if dtype is not None and dtype.is_floating_point and dtype != torch.float32:
            self._requested_dtype = dtype
            return super().to(device=device, non_blocking=non_blocking) if device is not None else self
        return super().to(*args, **kwargs)

    def prepare(self, dtype: Optional[torch.dtype] = None) -> "CLIPVisionModelWithProjection":
        dtype = dtype or self._requested_dtype or torch.float16
        dev = next(self.parameters()).device
        self.requires_grad_(False)
        self.rt = rt = Runtime(dtype, dev)
        cfg, vm = self.config, self.vision_model
        H = cfg.hidden_size
        kk = cfg.num_channels * cfg.patch_size ** 2
        self.k_patch = rup(kk, 64)
        wp = torch.zeros(H, self.k_patch, dtype=torch.float32, device=dev)
        wp[:, :kk] = vm.embeddings.patch_embedding.weight.data.reshape(H, kk)        # k = (c*ps + dy)*ps + dx: svdx_patch_rows order
        self.w_patch = rt.empty(H, self.k_patch)
        rt.k.cast_from_f32(wp.reshape(-1), self.w_patch, wp.numel())
        pos = vm.embeddings.position_embedding.weight.data.float()
        self.pos16 = pos.to(dtype).contiguous()                                          # residual operand of the patch GEMM
        self.cls_row = (vm.embeddings.class_embedding.data.float() + pos[0]).to(dtype).contiguous()
        self.ln_pre, self.ln_post = LayerNormOp(vm.pre_layrnorm), LayerNormOp(vm.post_layernorm)
        for layer in vm.encoder.layers:
            layer.build(rt)
        self.proj = LinearOp([self.visual_projection.weight])
        self.proj.pack(rt, need_dx=False)
        return self

    @torch.no_grad()
    def forward(self, pixel_values: torch.Tensor, return_dict: bool = True, **unused):
        """pixel_values [b, 3, image_size, image_size] (CLIP-normalised) -> `.image_embeds` [b, projection_dim] (float)."""
        if self.rt is None:
            self.prepare()
        rt, cfg, vm = self.rt, self.config, self.vision_model
        k = rt.k
        b, c, hh, ww = pixel_values.shape
        if (c, hh, ww) != (cfg.num_channels, cfg.image_size, cfg.image_size):
            raise ValueError(f"expected [b, {cfg.num_channels}, {cfg.image_size}, {cfg.image_size}], got {tuple(pixel_values.shape)}")
        H, ps, heads = cfg.hidden_size, cfg.patch_size, cfg.num_attention_heads
        g = cfg.image_size // ps
        npatch, S, d = g * g, g * g + 1, H // heads
        x = pixel_values.to(device=rt.dev, dtype=torch.float32).contiguous()
        a = rt.empty(b * npatch, self.k_patch)
        k.patch_rows(x, a, b, c, hh, ww, ps, ps, ps, 0, g, g, self.k_patch)
        tok = rt.empty(b * S, H)
        for i in range(b):                                                              # tokens of image i: [class | patches] + positions
            gemm_act(rt, a[i * npatch:], self.w_patch, tok[i * S + 1:], npatch, H, self.k_patch, self.k_patch, self.k_patch, H,
                     res=self.pos16[1:], ldres=H)
            tok[i * S].copy_(self.cls_row)
        M = b * S
        xcur, _ = self.ln_pre.fwd(rt, tok, M)
        act = _ACTS[cfg.hidden_act]
        F = cfg.intermediate_size
        for layer in vm.encoder.layers:
            n1, _ = layer.ln1.fwd(rt, xcur, M)
            qkv = layer.qkv.fwd(rt, n1, M)
            att = rt.empty(M, H)
            k.attn_small_fwd(qkv, att, b, S, heads, d, d, 3 * H, H, d ** -0.5)
            xcur = layer.o.fwd(rt, att, M, res=xcur)
            n2, _ = layer.ln2.fwd(rt, xcur, M)
            h1 = layer.f1.fwd(rt, n2, M)
            k.act_rows(h1, h1, M * F, act)                                              # element-wise, in place
            xcur = layer.f2.fwd(rt, h1, M, res=xcur)
        pooled = xcur.view(b, S, H)[:, 0].contiguous()                                    # class token rows
        pn, _ = self.ln_post.fwd(rt, pooled, b)
        emb = self.proj.fwd(rt, pn, b).float()
        if not return_dict:
            return (emb,)
        return SimpleNamespace(image_embeds=emb, last_hidden_state=xcur.view(b, S, H))


_TAPS = {}


_AFFINE = {}


def _gaussian_taps(factor: float, dev) -> torch.Tensor:
    """train_svd.py:141-161 + :218-232: sigma = max((factor - 1) / 2, 0.001); window = max(int(4 sigma), 3) made odd."""
    key = (round(float(factor), 9), str(dev))
    if key not in _TAPS:
        sigma = max((factor - 1.0) / 2.0, 0.001)
        ks = int(max(2.0 * 2 * sigma, 3))
        ks += 1 - ks % 2
        xs = torch.arange(ks, dtype=torch.float32) - ks // 2
        gss = torch.exp(-xs.pow(2.0) / (2 * torch.tensor(sigma, dtype=torch.float32).pow(2.0)))
        _TAPS[key] = (gss / gss.sum()).to(dev)
    return _TAPS[key]


def clip_pixel_values(frames: torch.Tensor, size: Sequence[int] = (224, 224), k=None) -> torch.Tensor:
    """frames [b, 3, h, w] in [-1, 1] -> CLIP input [b, 3, size] (float): blur + bicubic resize (train_svd.py:859), (v + 1) / 2
    (:861), mean / std normalisation (:864-871)."""
    from . import kernels as K
    k = k or K.backend()
    x = frames.to(torch.float32).contiguous()
    b, c, h, w = x.shape
    dev = x.device
    t1, t2 = torch.empty_like(x), torch.empty_like(x)
    k.blur_axis(x, t1, b * c, h, w, _gaussian_taps(w / size[1], dev), 0)
    k.blur_axis(t1, t2, b * c, h, w, _gaussian_taps(h / size[0], dev), 1)
    key = (c, str(dev))
    if key not in _AFFINE:                      # built once per device: a host->device copy per call would also forbid hipGraph capture
        mean = torch.tensor(CLIP_MEAN[:c], dtype=torch.float32, device=dev)
        std = torch.tensor(CLIP_STD[:c], dtype=torch.float32, device=dev)
        _AFFINE[key] = ((0.5 / std).contiguous(), ((0.5 - mean) / std).contiguous())
    scale, shift = _AFFINE[key]
    out = torch.empty(b, c, size[0], size[1], dtype=torch.float32, device=dev)
    k.bicubic_affine(t2, out, b, c, h, w, size[0], size[1], scale, shift)
    return out


def encode_image(pixel_values: torch.Tensor, image_encoder: CLIPVisionModelWithProjection) -> torch.Tensor:
    """train_svd.py:857-876: first frames [b, 3, h, w] in [-1, 1] -> image_embeds [b, projection_dim]."""
    if image_encoder.rt is None:
        image_encoder.prepare()
    s = image_encoder.config.image_size
    return image_encoder(clip_pixel_values(pixel_values.to(image_encoder.rt.dev), (s, s), image_encoder.rt.k)).image_embeds
