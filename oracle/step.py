"""CPU oracle: the training-step math of /root/reference/train_svd.py:941-1049 on latents.

TEST INFRASTRUCTURE ONLY (see oracle/unet.py header).  PINNED for the arithmetic that lives in the reference file itself:
`rand_log_normal`, the EDM noising, the conditioning dropout + channel concat and the weighted-MSE loss are checked against
values computed by the reference's own statements (tests/golden/make_golden_step_math.py and make_golden_resize.py lift them out of
/root/reference/train_svd.py and execute them in this container; tests/test_oracle_step_math.py).  The UNet call in the middle is
oracle/unet.py (parity unpinned, see its header).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch


def rand_log_normal(shape, loc=0.0, scale=1.0, device="cpu", dtype=torch.float32, generator=None):
    """train_svd.py:64-67 (k-diffusion).  CPU RNG by default, like the reference."""
    u = torch.rand(shape, dtype=dtype, device=device, generator=generator) * (1 - 2e-7) + 1e-7
    return torch.distributions.Normal(loc, scale).icdf(u).exp()


def get_add_time_ids(fps, motion_bucket_id, noise_aug_strength, dtype, batch_size):
    """train_svd.py:878-898 (without the config check, which lives on the product model)."""
    add_time_ids = torch.tensor([[fps, motion_bucket_id, float(noise_aug_strength)]], dtype=dtype)
    return add_time_ids.repeat(batch_size, 1)


def make_synthetic_batch(batch_size: int, num_frames: int, h: int, w: int, seed: int,
                         cross_dim: int = 1024) -> Dict[str, torch.Tensor]:
    """Synthetic stand-ins for the VAE latents / CLIP embed of train_svd.py:948-976, on the CPU RNG.

    latents ~ N(0,1)*0.7 (SVD latents after scaling_factor have O(1) std), conditional latents
    ~ N(0,1) / 0.18215-free scale, ehs ~ N(0,1) [B,1,cross_dim] (train_svd.py:1000-1001 unsqueeze)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    latents = 0.7 * torch.randn(batch_size, num_frames, 4, h, w, generator=g)
    noise = torch.randn(batch_size, num_frames, 4, h, w, generator=g)
    cond_latents = torch.randn(batch_size, 4, h, w, generator=g)
    ehs = torch.randn(batch_size, 1, cross_dim, generator=g)
    cond_sigmas = rand_log_normal([batch_size], loc=-3.0, scale=0.5, generator=g)     # :954
    sigmas = rand_log_normal([batch_size], loc=0.7, scale=1.6, generator=g)           # :964
    return dict(latents=latents, noise=noise, cond_latents=cond_latents, ehs=ehs,
                cond_sigmas=cond_sigmas, sigmas=sigmas)


def edm_inputs(batch: Dict[str, torch.Tensor]):
    """train_svd.py:964-972, 981-988, 1014-1017: builds the UNet inputs from a synthetic batch."""
    latents, noise = batch["latents"], batch["noise"]
    bsz, T = latents.shape[:2]
    sigmas = batch["sigmas"].to(latents)[:, None, None, None, None]
    noisy_latents = latents + noise * sigmas                                            # :968
    timesteps = torch.tensor([0.25 * s.log() for s in batch["sigmas"]], dtype=torch.float32)   # :969-970
    inp_noisy = noisy_latents / ((sigmas ** 2 + 1) ** 0.5)                              # :972
    added_time_ids = get_add_time_ids(7, 127, batch["cond_sigmas"][0], torch.float32, bsz)     # :981-988
    cond = batch["cond_latents"].unsqueeze(1).repeat(1, T, 1, 1, 1)                     # :1014-1015
    unet_in = torch.cat([inp_noisy, cond], dim=2)                                       # :1016-1017
    return unet_in, timesteps, batch["ehs"], added_time_ids, noisy_latents, sigmas


def conditioning_dropout(random_p: torch.Tensor, encoder_hidden_states: torch.Tensor, conditional_latents: torch.Tensor, prob: float):
    """train_svd.py:992-1011 (classifier-free-guidance dropout): with p = random_p per sample, the image embedding [B, D] is zeroed
    (and gains its sequence axis, [B, 1, D]) where p < 2 prob, the conditioning latents where prob <= p < 3 prob."""
    bsz = random_p.shape[0]
    prompt_mask = (random_p < 2 * prob).reshape(bsz, 1, 1)
    ehs = torch.where(prompt_mask, torch.zeros_like(encoder_hidden_states).unsqueeze(1), encoder_hidden_states.unsqueeze(1))
    dt = conditional_latents.dtype
    image_mask = 1 - ((random_p >= prob).to(dt) * (random_p < 3 * prob).to(dt))
    return ehs, image_mask.reshape(bsz, 1, 1, 1) * conditional_latents


def edm_loss(model_pred: torch.Tensor, noisy_latents: torch.Tensor, target: torch.Tensor,
             sigmas: torch.Tensor) -> torch.Tensor:
    """train_svd.py:1025-1036."""
    c_out = -sigmas / ((sigmas ** 2 + 1) ** 0.5)
    c_skip = 1 / (sigmas ** 2 + 1)
    denoised = model_pred * c_out + c_skip * noisy_latents
    weighing = (1 + sigmas ** 2) * (sigmas ** -2.0)
    loss = torch.mean((weighing.float() * (denoised.float() - target.float()) ** 2).reshape(target.shape[0], -1),
                      dim=1)
    return loss.mean()


def train_step(unet, batch: Dict[str, torch.Tensor], optimizer: Optional[torch.optim.Optimizer] = None):
    """One fp32 step: UNet fwd (:1021) -> EDM loss (:1025-1036) -> backward (:1044) -> AdamW (:1047-1049).
    Returns (loss, model_pred)."""
    unet_in, timesteps, ehs, added_time_ids, noisy_latents, sigmas = edm_inputs(batch)
    model_pred = unet(unet_in, timesteps, ehs, added_time_ids=added_time_ids).sample
    loss = edm_loss(model_pred, noisy_latents, batch["latents"], sigmas)
    loss.backward()
    if optimizer is not None:
        optimizer.step()
        optimizer.zero_grad()
    return loss.detach(), model_pred.detach()


def make_optimizer(unet, lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8):
    """train_svd.py:761-773: AdamW over params whose name contains 'temporal_transformer_block'."""
    params = []
    for name, p in unet.named_parameters():
        if "temporal_transformer_block" in name:
            p.requires_grad = True
            params.append(p)
        else:
            p.requires_grad = False
    return torch.optim.AdamW(params, lr=lr, betas=betas, weight_decay=weight_decay, eps=eps)
