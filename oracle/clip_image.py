"""CPU oracle for the image-conditioning path in front of the UNet: `encode_image` of /root/reference/train_svd.py:857-876 =
`_resize_with_antialiasing` (:140-166, with its helpers :169-248) -> un-normalise -> CLIP normalisation -> CLIP vision tower
-> `image_embeds`.

TEST INFRASTRUCTURE ONLY (see oracle/unet.py header); the product counterpart is svd_xtend_amd/clip.py.
PINNED: the resize is checked against outputs of the reference's own functions (tests/golden/resize_antialias.safetensors, made by
tests/golden/make_golden_resize.py, which executes the reference's function definitions from /root/reference/train_svd.py in this
container); the vision tower is `transformers.CLIPVisionModelWithProjection` itself (installed here; the reference calls the same
class, train_svd.py:642-644), so it needs no restatement.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)        # CLIPImageProcessor defaults (the `feature_extractor` of train_svd.py:864-871)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def blur_taps(factor: float) -> Tuple[int, float]:
    """Kernel size and sigma of the anti-aliasing blur along one axis for a down-scaling `factor` = in / out
    (train_svd.py:141-161): sigma = max((factor - 1) / 2, 0.001), taps = max(int(4 sigma), 3) made odd."""
    sigma = max((factor - 1.0) / 2.0, 0.001)
    taps = int(max(2.0 * 2 * sigma, 3))
    return taps + (1 - taps % 2), sigma


def gaussian_taps(taps: int, sigma: float, dtype=torch.float32) -> torch.Tensor:
    """Normalised Gaussian window centred on taps // 2 (train_svd.py:218-232; taps is odd here)."""
    x = torch.arange(taps, dtype=dtype) - taps // 2
    g = torch.exp(-x.pow(2.0) / (2 * torch.tensor(sigma, dtype=dtype).pow(2.0)))
    return g / g.sum()


def blur_axis(x: torch.Tensor, w: torch.Tensor, dim: int) -> torch.Tensor:
    """Depth-wise 1-D correlation along `dim` (-1 or -2) with reflect padding of (taps - 1) / 2 on both sides
    (train_svd.py:169-215: `_compute_padding` + `F.pad(mode="reflect")` + grouped conv2d)."""
    b, c, h, wd = x.shape
    half = (w.numel() - 1) // 2
    pad = (half, half, 0, 0) if dim == -1 else (0, 0, half, half)
    k = w.to(x).view(1, 1, 1, -1) if dim == -1 else w.to(x).view(1, 1, -1, 1)
    y = F.conv2d(F.pad(x, pad, mode="reflect").reshape(b * c, 1, h + (0 if dim == -1 else 2 * half), wd + (2 * half if dim == -1 else 0)), k)
    return y.view(b, c, h, wd)


def resize_with_antialiasing(x: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    """train_svd.py:140-166: separable Gaussian blur (x pass, then y pass), then bicubic interpolation with align_corners=True."""
    h, w = x.shape[-2:]
    ky, sy = blur_taps(h / size[0])
    kx, sx = blur_taps(w / size[1])
    x = blur_axis(x, gaussian_taps(kx, sx, x.dtype), -1)
    x = blur_axis(x, gaussian_taps(ky, sy, x.dtype), -2)
    return F.interpolate(x, size=tuple(size), mode="bicubic", align_corners=True)


def clip_pixel_values(frames: torch.Tensor, size=(224, 224)) -> torch.Tensor:
    """train_svd.py:859-871: frames in [-1, 1], [b, 3, h, w] -> CLIP input (resize, back to [0, 1], normalise; no crop, no rescale)."""
    x = (resize_with_antialiasing(frames, size) + 1.0) / 2.0
    mean = torch.tensor(CLIP_MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std


def encode_image(frames: torch.Tensor, image_encoder) -> torch.Tensor:
    """train_svd.py:857-876 with `image_encoder` a transformers CLIPVisionModelWithProjection: [b, 3, h, w] -> image_embeds [b, D]."""
    return image_encoder(clip_pixel_values(frames)).image_embeds
