"""The loop body of /root/reference/train_svd.py:931-1058 around the MI355X-native train step.

`TrainLoop` is what one iteration of the reference's `for step, batch in enumerate(train_dataloader)` does, re-scheduled for one HIP
stream and one hipGraph per optimizer step:

    pixel clip (host, pinned) --H2D--> VAE encode of the clip and of its noise-augmented first frame (:948-960)
                                       CLIP image embed of the first frame (:975-976)
                                       EDM noising, added_time_ids, conditioning dropout, channel concat (:951-1017)
                                   --> the captured input tensors of `GraphedStep` --> replay (forward, loss, backward, [all-reduce], AdamW)
                                   --> EMA (:1053-1054), loss read on the host (:1039-1041)

Schedule (north_star): the conditioners are frozen and do not depend on the update, so the NEXT clip's conditioners are queued
between the backward sweep and the optimizer of the CURRENT step (`GraphedStep.__call__(side_work=...)`): on several ranks they run
beside the one gradient all-reduce, on one rank they simply fill the stream while the host is free.  The only host synchronisation
per iteration is the loss read, and it happens after the next clip's work has been queued, so the device never waits for Python.
Everything random (noise, sigmas, dropout mask) is drawn on the device from one seeded generator; nothing is copied back.

`examples/train_svd_amd.py` is the runnable script around this class (arguments named as in the reference, checkpoints,
validation sampler); `bench.py` times it as `real_loop` beside the UNet-only headline."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from .clip import encode_image
from .train import GraphedStep, Trainer, conditioning_dropout, edm_prepare
from .vae import tensor_to_vae_latent

BATCH_KEYS = ("unet_in", "timesteps", "ehs", "added_time_ids", "noisy_latents", "target", "sigmas")


def rand_log_normal(shape, loc: float, scale: float, generator: torch.Generator, device) -> torch.Tensor:
    """train_svd.py:63-66 on the device: exp(Normal(loc, scale).icdf(u)), u ~ U(1e-7, 1 - 1e-7)."""
    u = torch.rand(shape, generator=generator, device=device, dtype=torch.float32) * (1 - 2e-7) + 1e-7
    return (loc + scale * (2.0 ** 0.5) * torch.erfinv(2.0 * u - 1.0)).exp()


class TrainLoop:
    def __init__(self, trainer: Trainer, vae, image_encoder, conditioning_dropout_prob: Optional[float] = None, seed: int = 0,
                 use_graph: bool = True, ema=None, fps: int = 7, motion_bucket_id: int = 127):
        self.tr, self.vae, self.enc = trainer, vae, image_encoder
        self.p_drop = conditioning_dropout_prob
        self.use_graph = use_graph and trainer.dev.type == "cuda"
        self.ema = ema
        self.fps, self.bucket = fps, motion_bucket_id
        self.dev = trainer.dev
        self.gen = torch.Generator(device=self.dev).manual_seed(seed)
        self.batches: Optional[List[Dict[str, torch.Tensor]]] = None      # the tensors the captured step reads
        self.staged: Optional[List[Dict[str, torch.Tensor]]] = None       # the next step's batch, produced while this one runs
        self.graphed: Optional[GraphedStep] = None
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.global_step = 0

    # ---- conditioners + EDM data prep of one micro-batch, all on the device (train_svd.py:942-1017) ----------------------------
    @torch.no_grad()
    def prepare_batch(self, pixel_values: torch.Tensor) -> Dict[str, torch.Tensor]:
        """pixel_values [B, T, 3, H, W] in [-1, 1] (host or device) -> the step's inputs on the device."""
        dev, g = self.dev, self.gen
        pix = pixel_values.to(dev, non_blocking=True).to(torch.float32)
        bsz, T = pix.shape[:2]
        cond_sigmas = rand_log_normal([bsz], -3.0, 0.5, g, dev)                                   # :954
        noise_aug_strength = cond_sigmas[0]                                                       # :955 (the reference's batch-1 TODO)
        cpix = pix[:, 0:1]
        cpix = torch.randn(cpix.shape, generator=g, device=dev) * cond_sigmas[:, None, None, None, None] + cpix      # :957-958
        # one encoder pass over the clip's T frames and the noise-augmented first frame (:948 and :959 are two calls there)
        z = tensor_to_vae_latent(torch.cat([pix, cpix], dim=1), self.vae, generator=g)
        latents = z[:, :T]
        conditional_latents = z[:, T] / self.vae.config.scaling_factor                            # :959-960
        noise = torch.randn(latents.shape, generator=g, device=dev, dtype=latents.dtype)          # :951
        sigmas = rand_log_normal([bsz], 0.7, 1.6, g, dev)                                         # :964
        ehs = encode_image(pix[:, 0], self.enc).to(torch.float32)                                 # :975-976
        ids = torch.stack([torch.full_like(noise_aug_strength, float(self.fps)), torch.full_like(noise_aug_strength, float(self.bucket)),
                           noise_aug_strength]).unsqueeze(0).repeat(bsz, 1)                       # :981-988 (fps passed as 7 there)
        if self.p_drop is not None:                                                               # :992-1011
            random_p = torch.rand(bsz, generator=g, device=dev)
            ehs, conditional_latents = conditioning_dropout(random_p, ehs, conditional_latents, self.p_drop)
        else:
            ehs = ehs.unsqueeze(1)
        unet_in, timesteps, noisy = edm_prepare(latents, noise, conditional_latents, sigmas)      # :966-972, :1014-1017
        return dict(unet_in=unet_in.contiguous(), timesteps=timesteps, ehs=ehs.contiguous(), added_time_ids=ids.contiguous(),
                    noisy_latents=noisy.contiguous(), target=latents.contiguous(), sigmas=sigmas)

    def _prepare_all(self, clips: Sequence[torch.Tensor]) -> List[Dict[str, torch.Tensor]]:
        if len(clips) != self.tr.grad_accum:
            raise ValueError(f"expected {self.tr.grad_accum} clip(s) per optimizer step, got {len(clips)}")
        return [self.prepare_batch(c) for c in clips]

    @staticmethod
    def _as_list(clips) -> List[torch.Tensor]:
        return [clips] if isinstance(clips, torch.Tensor) else list(clips)

    # ---- the pipeline ----------------------------------------------------------------------------------------------------------
    def start(self, clips) -> None:
        """First clip(s) of the run: conditioners, then capture of the step on the tensors they produced."""
        self.batches = self._prepare_all(self._as_list(clips))
        self.staged = None
        if self.use_graph:
            # GraphedStep's warm-up pass is one real optimizer step on this batch; the loop counts it as step 1 (see step())
            self.graphed = GraphedStep(self.tr, self.batches)
            self._after_step()
            self._warm = True
        else:
            self._warm = False

    def _after_step(self) -> None:
        self._loss.copy_(self.tr.last_loss())            # the next zero_grad clears the loss slot: keep this step's value
        if self.ema is not None:
            self.ema.step(self.tr.model.parameters())
        self.global_step += 1

    def step(self, next_clips=None) -> float:
        """One optimizer step on the current clip(s); `next_clips` (the following iteration's pixel clips, or None at the end of the
        run) go through the conditioners between this step's backward sweep and its optimizer.  Returns this step's mean loss."""
        if self.batches is None:
            raise RuntimeError("TrainLoop.start(first_clips) first")
        nxt = self._as_list(next_clips) if next_clips is not None else None

        def side():
            self.staged = self._prepare_all(nxt) if nxt is not None else None

        if self._warm:                                   # the capture's warm-up pass already was this step
            self._warm = False
            side()
        else:
            if self.graphed is not None:
                self.graphed(side_work=side)
            else:
                self.tr.step(self.batches if self.tr.grad_accum > 1 else self.batches[0], side_work=side)
            self._after_step()
        if self.staged is not None:                      # hand the next batch to the captured tensors (in stream order: after the step)
            for dst, src in zip(self.batches, self.staged):
                for k in BATCH_KEYS:
                    dst[k].copy_(src[k])
            self.staged = None
        return float(self._loss)                         # the iteration's one host synchronisation (train_svd.py:1039-1041)
