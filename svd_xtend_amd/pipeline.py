"""The validation sampler of the reference loop on the MI355X-native modules: `StableVideoDiffusionPipeline(...)(image, height, width,
num_frames, decode_chunk_size=8, motion_bucket_id=127, fps=7, noise_aug_strength=0.02).frames[0]`
(/root/reference/train_svd.py:1106-1137, infer_svd.ipynb cell 3).  The reference builds diffusers'
pipeline around its own UNet, CLIP tower and VAE; this class has that surface (`from_pretrained(path, unet=, image_encoder=, vae=)`,
`.to`, `set_progress_bar_config`, `__call__` -> `.frames`) and runs

  CLIP embedding of the image            svd_xtend_amd.clip (anti-aliased resize kernels + ViT-H tower)
  noise-augmented conditioning latent    svd_xtend_amd.vae encoder, `latent_dist.mode()` (NOT scaled by scaling_factor, as diffusers)
  25 Euler steps, v-prediction, Karras   UNet forward on the HIP path (classifier-free guidance = a batch of 2, guidance rising
  sigmas 700 -> 0.002                    linearly over the frames)
  temporal VAE decoder                   svd_xtend_amd.vae decoder, `decode_chunk_size` frames per call

`EulerDiscreteScheduler` restates the subset of diffusers' scheduler that SVD's `scheduler_config.json` selects (Karras sigmas from
sigma_min / sigma_max, `timestep_type="continuous"` -> t = 0.25 ln sigma, `timestep_spacing="leading"` -> init_noise_sigma =
sqrt(sigma_max^2 + 1), `prediction_type="v_prediction"`, s_churn = 0).  The per-step latent arithmetic is fp32 torch on the device --
a few elementwise passes over [B, T, 4, h, w] (143 k values at 512x320) between UNet forwards of ~20 ms; not a kernel target.

Known deviation from diffusers: its pipeline casts the VAE to fp32 for the single conditioning-frame encode and for decoding when
`vae.config.force_upcast` is set (SVD's VAE sets it) and back afterwards.  The VAE here has no fp32 storage path: encoder and decoder
run with 16-bit activations / weights, fp32 MFMA accumulation and fp32 GroupNorm statistics.  Against the fp32 CPU oracle at the SVD
widths that costs rel-L2 1.2e-3 / 1.4e-3 on the latent mean / logvar and 1.7e-3 on decoded frames (tests/test_vae.py), below the 16-bit
rounding of the UNet input the latent is concatenated into."""
from __future__ import annotations

import json
import math
import os
from types import SimpleNamespace
from typing import List, Optional, Union

import numpy as np
import torch

from .clip import CLIPVisionModelWithProjection, encode_image
from .unet import FrozenConfig, UNetSpatioTemporalConditionModel
from .vae import AutoencoderKLTemporalDecoder


class EulerDiscreteScheduler:
    def __init__(self, sigma_min: float = 0.002, sigma_max: float = 700.0, prediction_type: str = "v_prediction",
                 timestep_type: str = "continuous", timestep_spacing: str = "leading", use_karras_sigmas: bool = True, **other):
        if prediction_type != "v_prediction" or timestep_type != "continuous" or not use_karras_sigmas:
            raise NotImplementedError("only SVD's scheduler configuration (v_prediction, continuous timesteps, Karras sigmas) is built")
        self.config = FrozenConfig(sigma_min=sigma_min, sigma_max=sigma_max, prediction_type=prediction_type, timestep_type=timestep_type,
                                   timestep_spacing=timestep_spacing, use_karras_sigmas=use_karras_sigmas, **other)
        self.sigmas = self.timesteps = None
        self._step_index = None

    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = "scheduler", **unused):
        f = os.path.join(path, subfolder or "", "scheduler_config.json")
        cfg = {}
        if os.path.exists(f):
            with open(f) as fh:
                cfg = {k: v for k, v in json.load(fh).items() if not k.startswith("_")}
        return cls(**cfg)

    def set_timesteps(self, num_inference_steps: int, device=None) -> None:
        rho = 7.0
        ramp = torch.linspace(0, 1, num_inference_steps, dtype=torch.float64)
        lo, hi = self.config.sigma_min ** (1 / rho), self.config.sigma_max ** (1 / rho)
        s = ((hi + ramp * (lo - hi)) ** rho).to(torch.float32)
        self.timesteps = (0.25 * s.log()).to(device)
        self.sigmas = torch.cat([s, torch.zeros(1)])            # host copy: the loop reads sigma as a Python float
        self._step_index = 0

    @property
    def init_noise_sigma(self) -> float:
        m = float(self.sigmas.max())
        return m if self.config.timestep_spacing in ("linspace", "trailing") else (m * m + 1.0) ** 0.5

    def scale_model_input(self, sample: torch.Tensor, timestep=None) -> torch.Tensor:
        s = float(self.sigmas[self._step_index])
        return sample / (s * s + 1.0) ** 0.5

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, return_dict: bool = True):
        s, s_next = float(self.sigmas[self._step_index]), float(self.sigmas[self._step_index + 1])
        sample = sample.to(torch.float32)
        pred_x0 = model_output * (-s / (s * s + 1.0) ** 0.5) + sample / (s * s + 1.0)
        prev = sample + (sample - pred_x0) / s * (s_next - s)
        self._step_index += 1
        if not return_dict:
            return (prev,)
        return SimpleNamespace(prev_sample=prev, pred_original_sample=pred_x0)


def decode_latents(latents: torch.Tensor, vae: AutoencoderKLTemporalDecoder, num_frames: int, decode_chunk_size: int = 14) -> torch.Tensor:
    """StableVideoDiffusionPipeline.decode_latents: [b, f, 4, h, w] -> [b, 3, f, 8h, 8w] floats in about [-1, 1]; every chunk of
    `decode_chunk_size` frames is decoded as one clip of that many frames."""
    b = latents.shape[0]
    flat = latents.flatten(0, 1).to(torch.float32) / vae.config.scaling_factor
    frames = []
    for i in range(0, flat.shape[0], decode_chunk_size):
        chunk = flat[i:i + decode_chunk_size]
        frames.append(vae.decode(chunk, num_frames=chunk.shape[0]).sample)
    frames = torch.cat(frames, dim=0)
    return frames.reshape(b, num_frames, *frames.shape[1:]).permute(0, 2, 1, 3, 4).float()


def _to_unit_tensor(image, height: int, width: int) -> torch.Tensor:
    """PIL image / uint8 HWC array / float tensor [3, H, W] or [n, 3, H, W] in [0, 1] -> float [n, 3, height, width] in [0, 1]
    (VaeImageProcessor.preprocess before its normalisation; PIL inputs are resized with Lanczos as diffusers does)."""
    if isinstance(image, torch.Tensor):
        t = image.to(torch.float32)
        t = t[None] if t.ndim == 3 else t
    else:
        if hasattr(image, "resize") and hasattr(image, "size") and not isinstance(image, np.ndarray):      # PIL
            if tuple(image.size) != (width, height):
                from PIL import Image
                image = image.resize((width, height), Image.LANCZOS)
            image = np.asarray(image.convert("RGB"))
        a = np.asarray(image)
        a = a[None] if a.ndim == 3 else a
        t = torch.from_numpy(a.astype(np.float32) / (255.0 if a.dtype == np.uint8 else 1.0)).permute(0, 3, 1, 2)
    if tuple(t.shape[-2:]) != (height, width):
        raise ValueError(f"image tensor is {tuple(t.shape[-2:])}, expected ({height}, {width}): resize it first")
    return t.contiguous()


class StableVideoDiffusionPipelineOutput(SimpleNamespace):
    pass


class StableVideoDiffusionPipeline:
    def __init__(self, vae: AutoencoderKLTemporalDecoder, image_encoder: CLIPVisionModelWithProjection,
                 unet: UNetSpatioTemporalConditionModel, scheduler: Optional[EulerDiscreteScheduler] = None, feature_extractor=None):
        self.vae, self.image_encoder, self.unet = vae, image_encoder, unet
        self.scheduler = scheduler or EulerDiscreteScheduler()
        self.vae_scale_factor = 2 ** (len(vae.config.block_out_channels) - 1)
        self.device = None

    @classmethod
    def from_pretrained(cls, path, unet=None, image_encoder=None, vae=None, scheduler=None, revision=None, variant=None,
                        torch_dtype=None, **unused):
        """train_svd.py:1106-1114: the training loop hands its live UNet / CLIP tower / VAE over; whatever is missing is loaded from
        `<path>/{unet,image_encoder,vae,scheduler}`."""
        if unet is None:
            unet = UNetSpatioTemporalConditionModel.from_pretrained(path, subfolder="unet", variant=variant)
        if image_encoder is None:
            image_encoder = CLIPVisionModelWithProjection.from_pretrained(path, subfolder="image_encoder", variant=variant)
        if vae is None:
            vae = AutoencoderKLTemporalDecoder.from_pretrained(path, subfolder="vae", variant=variant)
        if scheduler is None:
            scheduler = EulerDiscreteScheduler.from_pretrained(path) if path and os.path.isdir(str(path)) else EulerDiscreteScheduler()
        pipe = cls(vae, image_encoder, unet, scheduler)
        pipe._dtype = torch_dtype
        return pipe

    def to(self, device=None, dtype=None):
        if device is not None:
            self.device = torch.device(device)
            self.unet.to(self.device)
            self.vae.to(self.device)
            self.image_encoder.to(self.device)
        return self

    def set_progress_bar_config(self, **kw) -> None:
        pass

    def _prepare(self):
        dt = getattr(self, "_dtype", None)
        dt = dt if dt in (torch.float16, torch.bfloat16) else None
        for m in (self.unet, self.vae, self.image_encoder):
            if m.rt is None:
                m.prepare(dt) if dt is not None else m.prepare()
        return self.unet.rt.dev

    @torch.no_grad()
    def __call__(self, image, height: int = 576, width: int = 1024, num_frames: Optional[int] = None, num_inference_steps: int = 25,
                 min_guidance_scale: float = 1.0, max_guidance_scale: float = 3.0, fps: int = 7, motion_bucket_id: int = 127,
                 noise_aug_strength: float = 0.02, decode_chunk_size: Optional[int] = None, num_videos_per_prompt: int = 1,
                 generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None,
                 output_type: str = "pil", return_dict: bool = True):
        if height % 8 or width % 8:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        dev = self._prepare()
        num_frames = num_frames if num_frames is not None else self.unet.config.num_frames
        decode_chunk_size = decode_chunk_size if decode_chunk_size is not None else num_frames
        img = _to_unit_tensor(image, height, width)
        b = img.shape[0] * num_videos_per_prompt
        cfg = max_guidance_scale > 1.0
        x = (img * 2.0 - 1.0).to(dev)
        # CLIP embedding (pipeline._encode_image); the unconditional half is zeros
        emb = encode_image(x, self.image_encoder).to(torch.float32).unsqueeze(1).repeat_interleave(num_videos_per_prompt, 0)
        if cfg:
            emb = torch.cat([torch.zeros_like(emb), emb])
        # noise-augmented conditioning latent, the mode of the posterior, not scaled
        gdev = generator.device if generator is not None else dev
        noise = torch.randn(x.shape, generator=generator, device=gdev, dtype=torch.float32).to(dev)
        cond = self.vae.encode(x + noise_aug_strength * noise).latent_dist.mode().repeat_interleave(num_videos_per_prompt, 0)
        if cfg:
            cond = torch.cat([torch.zeros_like(cond), cond])
        cond = cond.unsqueeze(1).repeat(1, num_frames, 1, 1, 1)
        ids = torch.tensor([[float(fps - 1), float(motion_bucket_id), float(noise_aug_strength)]], device=dev).repeat(b, 1)
        if cfg:
            ids = torch.cat([ids, ids])
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        zc = self.unet.config.in_channels // 2
        shape = (b, num_frames, zc, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32)
        latents = latents.to(dev) * self.scheduler.init_noise_sigma
        gs = torch.linspace(min_guidance_scale, max_guidance_scale, num_frames, device=dev).view(1, num_frames, 1, 1, 1)
        for i in range(num_inference_steps):
            t = self.scheduler.timesteps[i]
            inp = torch.cat([latents] * 2) if cfg else latents
            inp = torch.cat([self.scheduler.scale_model_input(inp, t), cond], dim=2)
            out = self.unet(inp, t, encoder_hidden_states=emb, added_time_ids=ids, return_dict=False)[0]
            if cfg:
                u, c = out.chunk(2)
                out = u + gs * (c - u)
            latents = self.scheduler.step(out, t, latents).prev_sample
        if output_type == "latent":
            frames = latents
        else:
            frames = tensor2vid(decode_latents(latents, self.vae, num_frames, decode_chunk_size), output_type)
        if not return_dict:
            return frames
        return StableVideoDiffusionPipelineOutput(frames=frames)


def tensor2vid(video: torch.Tensor, output_type: str = "np") -> Union[torch.Tensor, np.ndarray, List[list]]:
    """diffusers' tensor2vid + VaeImageProcessor.postprocess: [b, 3, f, H, W] in [-1, 1] -> per clip `f` frames in [0, 1]
    ("pt": [b, f, 3, H, W] tensor; "np": [b, f, H, W, 3] array; "pil": list of lists of PIL images)."""
    v = (video.permute(0, 2, 1, 3, 4) / 2 + 0.5).clamp(0, 1)
    if output_type == "pt":
        return v
    a = v.cpu().permute(0, 1, 3, 4, 2).float().numpy()
    if output_type == "np":
        return a
    if output_type == "pil":
        from PIL import Image
        return [[Image.fromarray((f * 255).round().astype("uint8")) for f in clip] for clip in a]
    raise ValueError(f"output_type {output_type!r}: expected 'pil', 'np', 'pt' or 'latent'")
