"""CPU oracle for the validation sampler the reference runs every `validation_steps` (/root/reference/train_svd.py:1093-1150;
infer_svd.ipynb cell 3): `StableVideoDiffusionPipeline.__call__` with an `EulerDiscreteScheduler` configured as SVD's
`scheduler/scheduler_config.json` (v_prediction, Karras sigmas 700 -> 0.002, continuous timesteps 0.25 ln sigma, "leading" spacing).

TEST INFRASTRUCTURE ONLY (see oracle/unet.py header).  PARITY UNPINNED: the pipeline and the scheduler live in diffusers, which is
neither vendored in /root/reference nor installed here; both are restated from the published diffusers 0.26 modules
(`pipelines/stable_video_diffusion/pipeline_stable_video_diffusion.py`, `schedulers/scheduling_euler_discrete.py`).  The pieces
that ARE pinned elsewhere are reused, not restated: the anti-aliased resize + CLIP normalisation (oracle/clip_image.py, pinned to the
reference's own functions), the UNet top level (pinned to the reference's class), `_get_add_time_ids` semantics.

Everything is float32 on the CPU; `unet`, `vae`, `image_encoder` are the oracle modules (oracle/unet.py, oracle/vae.py, transformers'
CLIPVisionModelWithProjection)."""
from __future__ import annotations

import math
from typing import Optional

import torch

from .clip_image import clip_pixel_values
from .vae import decode_latents

SVD_SCHEDULER_CONFIG = dict(sigma_min=0.002, sigma_max=700.0, rho=7.0, prediction_type="v_prediction", timestep_type="continuous",
                            timestep_spacing="leading", use_karras_sigmas=True)


def karras_sigmas(num_inference_steps: int, sigma_min: float = 0.002, sigma_max: float = 700.0, rho: float = 7.0) -> torch.Tensor:
    """EulerDiscreteScheduler._convert_to_karras with the config's sigma_min / sigma_max, plus the trailing 0 that set_timesteps
    appends: [n + 1] float32."""
    ramp = torch.linspace(0, 1, num_inference_steps, dtype=torch.float64)
    lo, hi = sigma_min ** (1 / rho), sigma_max ** (1 / rho)
    s = (hi + ramp * (lo - hi)) ** rho
    return torch.cat([s.to(torch.float32), torch.zeros(1)])


def euler_step_v(sample: torch.Tensor, model_output: torch.Tensor, sigma: float, sigma_next: float) -> torch.Tensor:
    """EulerDiscreteScheduler.step, prediction_type = v_prediction, s_churn = 0 (so sigma_hat = sigma, no noise is added)."""
    pred_x0 = model_output * (-sigma / (sigma ** 2 + 1) ** 0.5) + sample / (sigma ** 2 + 1)
    return sample + (sample - pred_x0) / sigma * (sigma_next - sigma)


@torch.no_grad()
def svd_sample(image: torch.Tensor, unet, vae, image_encoder, num_frames: int, num_inference_steps: int = 25,
               min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0, fps: int = 7, motion_bucket_id: int = 127,
               noise_aug_strength: float = 0.02, decode_chunk_size: Optional[int] = None,
               generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None, output_latents: bool = False):
    """image [1, 3, H, W] in [0, 1] (what VaeImageProcessor hands on after pil_to_numpy / numpy_to_pt, already at the target size).
    Returns frames [1, 3, num_frames, H, W] in [-1, 1] (decode_latents' output, before tensor2vid's denormalisation)."""
    b, _, H, W = image.shape
    cfg = max_guidance_scale > 1.0
    x = image * 2.0 - 1.0
    # 3. CLIP embedding of the conditioning image (pipeline._encode_image == train_svd.py:857-876)
    side = image_encoder.config.image_size                                                  # 224 for SVD's ViT-H/14 (the pipeline hard-codes 224)
    emb = image_encoder(clip_pixel_values(x, (side, side))).image_embeds.unsqueeze(1)       # [b, 1, D]
    if cfg:
        emb = torch.cat([torch.zeros_like(emb), emb])
    # 4. noise-augmented conditioning latent: latent_dist.mode(), NOT scaled by scaling_factor (pipeline._encode_vae_image)
    noise = torch.randn(x.shape, generator=generator, dtype=x.dtype)
    cond = vae.moments(x + noise_aug_strength * noise)[0]
    if cfg:
        cond = torch.cat([torch.zeros_like(cond), cond])
    cond = cond.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)
    # 5. micro-conditioning: fps - 1, motion bucket, noise strength
    ids = torch.tensor([[float(fps - 1), float(motion_bucket_id), float(noise_aug_strength)]]).repeat(b, 1)
    if cfg:
        ids = torch.cat([ids, ids])
    # 6. schedule and start noise
    sig = karras_sigmas(num_inference_steps)
    init_noise_sigma = float((sig.max() ** 2 + 1) ** 0.5)                                   # timestep_spacing "leading"
    if latents is None:
        latents = torch.randn(b, num_frames, 4, H // 8, W // 8, generator=generator)
    latents = latents * init_noise_sigma
    gs = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames).view(1, num_frames, 1, 1, 1)
    # 8. Euler loop
    for i in range(num_inference_steps):
        s, s_next = float(sig[i]), float(sig[i + 1])
        t = torch.tensor(0.25 * math.log(s))
        inp = torch.cat([latents] * 2) if cfg else latents
        inp = inp / (s ** 2 + 1) ** 0.5
        inp = torch.cat([inp, cond], dim=2)
        out = unet(inp, t, emb, ids, return_dict=False)[0]
        if cfg:
            u, c = out.chunk(2)
            out = u + gs * (c - u)
        latents = euler_step_v(latents, out, s, s_next)
    if output_latents:
        return latents
    return decode_latents(latents, vae, num_frames, decode_chunk_size or num_frames)
