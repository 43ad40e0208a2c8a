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

from typing import Dict, List, Optional

import torch

from .clip import encode_image
from .train import GraphedStep, Trainer, conditioning_dropout, edm_prepare

BATCH_KEYS = ("unet_in", "timesteps", "ehs", "added_time_ids", "noisy_latents", "target", "sigmas")


def rand_log_normal_reference(shape, loc: float = 0.0, scale: float = 1.0) -> torch.Tensor:
    """The reference's sigma draw, arithmetic and generator included (train_svd.py:63-67, called without a device at :954 and :964):
    uniforms from the PROCESS-GLOBAL CPU generator, the normal's inverse CDF on the host.  `torch.manual_seed(s)` therefore reproduces
    the reference's sigma sequence (tests/test_train_loop.py pins it to the values the reference's own function produced)."""
    u = torch.rand(shape, dtype=torch.float32, device="cpu") * (1 - 2e-7) + 1e-7
    return torch.distributions.Normal(loc, scale).icdf(u).exp()


class TrainLoop:
    def __init__(self, trainer: Trainer, vae, image_encoder, conditioning_dropout_prob: Optional[float] = None, seed: int = 0,
                 use_graph: bool = True, ema=None, fps: int = 7, motion_bucket_id: int = 127, graph_conditioners: bool = True,
                 reference_rng: bool = False, overlap_clip: bool = True, overlap_optimizer: bool = True):
        """reference_rng: draw cond_sigmas (:954) and sigmas (:964) the reference's way -- on the host, from the process-global generator, in
        that order -- instead of on the device from this loop's generator: a run seeded with `torch.manual_seed` then walks the reference's
        sigma sequence.  (The Gaussian noise tensors stay device draws: the reference's come from the CUDA generator, which no other device
        reproduces.)  Costs two 4-byte host-to-device copies per micro-batch."""
        self.reference_rng = reference_rng
        self.tr, self.vae, self.enc = trainer, vae, image_encoder
        # overlap_clip: the CLIP embed of the first frame (~300 launches of 5-25 us on <= 150 workgroups each: latency-bound, 3.5 ms) runs on a
        # second stream BESIDE the VAE encode (chip-filling convolutions, 15.5 ms) instead of after it -- fork / join by events, also inside the
        # captured conditioner graph.  Same arithmetic, same results; the two towers share nothing but the pixel clip.
        self.overlap_clip = overlap_clip and trainer.dev.type == "cuda"
        self._clip_stream = torch.cuda.Stream(device=trainer.dev) if self.overlap_clip else None
        # overlap_optimizer: on one rank the captured optimizer (AdamW: HBM streaming, no matrix pipe) runs on a second stream beside the next
        # clip's conditioners instead of after them (GraphedStep.opt_beside_side_work)
        self.overlap_optimizer = overlap_optimizer
        self.p_drop = conditioning_dropout_prob
        self.use_graph = use_graph and trainer.dev.type == "cuda"
        self.ema = ema
        self.fps, self.bucket = fps, motion_bucket_id
        self.dev = trainer.dev
        self.gen = torch.Generator(device=self.dev).manual_seed(seed)
        self.batches: Optional[List[Dict[str, torch.Tensor]]] = None      # the tensors the (captured) step reads
        self.graphed: Optional[GraphedStep] = None
        self.graph_conditioners = graph_conditioners and self.use_graph
        self.cond_graph = None
        self._loss = torch.zeros(1, dtype=torch.float32, device=self.dev)
        self.global_step = 0

    # ---- conditioners + EDM data prep of one micro-batch, all on the device (train_svd.py:942-1017) ----------------------------
    def _draw_shapes(self, bsz: int, T: int, H: int, W: int) -> Dict[str, tuple]:
        down = 2 ** (len(self.vae.config.block_out_channels) - 1)
        zc, h, w = self.vae.config.latent_channels, H // down, W // down
        shapes = dict(u_cond=(bsz,), n_cpix=(bsz, 1, 3, H, W), eps=(bsz, T + 1, zc, h, w), noise=(bsz, T, zc, h, w), u_sig=(bsz,), p_drop=(bsz,))
        if self.reference_rng:
            shapes.update(cond_sigmas=(bsz,), sigmas=(bsz,))
        return shapes

    def _draw(self, bsz: int, T: int, H: int, W: int, into: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
        """Every random number of one micro-batch, in a fixed order from the loop's generator (eager launches: the arithmetic that consumes
        them may then be replayed from a hipGraph without the generator being part of the capture)."""
        shapes = self._draw_shapes(bsz, T, H, W)
        out = into if into is not None else {k: torch.empty(v, dtype=torch.float32, device=self.dev) for k, v in shapes.items()}
        for k in ("u_cond", "u_sig", "p_drop"):
            torch.rand(shapes[k], generator=self.gen, out=out[k])
        for k in ("n_cpix", "eps", "noise"):
            torch.randn(shapes[k], generator=self.gen, out=out[k])
        if self.reference_rng:                                      # the reference's order: cond_sigmas (:954), then sigmas (:964)
            out["cond_sigmas"].copy_(rand_log_normal_reference(shapes["cond_sigmas"], loc=-3.0, scale=0.5), non_blocking=True)
            out["sigmas"].copy_(rand_log_normal_reference(shapes["sigmas"], loc=0.7, scale=1.6), non_blocking=True)
        return out

    @torch.no_grad()
    def _compute(self, pix: torch.Tensor, d: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """pix [B, T, 3, H, W] float in [-1, 1] on the device + the draws -> the step's inputs.  No host<->device traffic, no generator: capturable."""
        bsz, T = pix.shape[:2]
        log_normal = lambda u, loc, scale: (loc + scale * (2.0 ** 0.5) * torch.erfinv(2.0 * (u * (1 - 2e-7) + 1e-7) - 1.0)).exp()   # noqa: E731  (:63-66)
        cond_sigmas = d["cond_sigmas"] if "cond_sigmas" in d else log_normal(d["u_cond"], -3.0, 0.5)   # :954
        noise_aug_strength = cond_sigmas[0]                                                       # :955 (the reference's batch-1 TODO)
        cpix = d["n_cpix"] * cond_sigmas[:, None, None, None, None] + pix[:, 0:1]                 # :957-958
        # one encoder pass over the clip's T frames and the noise-augmented first frame (:948 and :959 are two calls there)
        frames = torch.cat([pix, cpix], dim=1)
        ehs = None
        if self.overlap_clip:
            cur = torch.cuda.current_stream()
            self._clip_stream.wait_stream(cur)                                                    # fork: the pixel clip is ready
            with torch.cuda.stream(self._clip_stream):
                ehs = encode_image(pix[:, 0], self.enc).to(torch.float32)                         # :975-976
            ehs.record_stream(cur)
        dist_ = self.vae.encode(frames.reshape(bsz * (T + 1), *frames.shape[2:])).latent_dist    # tensor_to_vae_latent, :283-291
        z = (dist_.mean + dist_.std * d["eps"].reshape(dist_.mean.shape)).reshape(bsz, T + 1, *dist_.mean.shape[1:]) * self.vae.config.scaling_factor
        latents = z[:, :T]
        conditional_latents = z[:, T] / self.vae.config.scaling_factor                            # :959-960
        sigmas = d["sigmas"] if "sigmas" in d else log_normal(d["u_sig"], 0.7, 1.6)              # :964
        if ehs is None:
            ehs = encode_image(pix[:, 0], self.enc).to(torch.float32)                             # :975-976
        else:
            torch.cuda.current_stream().wait_stream(self._clip_stream)                            # join
        ids = torch.stack([torch.full_like(noise_aug_strength, float(self.fps)), torch.full_like(noise_aug_strength, float(self.bucket)),
                           noise_aug_strength]).unsqueeze(0).repeat(bsz, 1)                       # :981-988 (fps passed as 7 there)
        if self.p_drop is not None:                                                               # :992-1011
            ehs, conditional_latents = conditioning_dropout(d["p_drop"], ehs, conditional_latents, self.p_drop)
        else:
            ehs = ehs.unsqueeze(1)
        unet_in, timesteps, noisy = edm_prepare(latents, d["noise"], conditional_latents, sigmas)  # :966-972, :1014-1017
        return dict(unet_in=unet_in.contiguous(), timesteps=timesteps, ehs=ehs.contiguous(), added_time_ids=ids.contiguous(),
                    noisy_latents=noisy.contiguous(), target=latents.contiguous(), sigmas=sigmas)

    @torch.no_grad()
    def prepare_batch(self, pixel_values: torch.Tensor) -> Dict[str, torch.Tensor]:
        """pixel_values [B, T, 3, H, W] in [-1, 1] (host or device) -> the step's inputs on the device (eager launches)."""
        pix = pixel_values.to(self.dev, non_blocking=True).to(torch.float32)
        return self._compute(pix, self._draw(pix.shape[0], pix.shape[1], pix.shape[3], pix.shape[4]))

    def _capture_conditioners(self, like: torch.Tensor) -> None:
        """The conditioners of one micro-batch as ONE hipGraph over static inputs (pixels, draws) and outputs: ~600 eager launches of
        5-10 us of host time each become one replay (the eager form is host-bound once the UNet step itself comes from a graph)."""
        self._pix = torch.empty(like.shape, dtype=torch.float32, device=self.dev)
        # static draw buffers: filled per clip by _draw(into=...); zeros for the warm-up and the capture (the generator is not touched here,
        # so the captured and the eager loop consume the same random sequence)
        self._draws = {k: torch.zeros(v, dtype=torch.float32, device=self.dev)
                       for k, v in self._draw_shapes(like.shape[0], like.shape[1], like.shape[3], like.shape[4]).items()}
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            self._pix.copy_(like.to(self.dev, non_blocking=True))
            self._compute(self._pix, self._draws)              # warm-up on the capture stream
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                self._cond_out = self._compute(self._pix, self._draws)
        torch.cuda.current_stream().wait_stream(s)
        self.cond_graph = g

    def _produce(self, j: int, clip: torch.Tensor) -> None:
        """Micro-batch j of the NEXT step: conditioners on `clip`, result written into the tensors the captured step reads (in stream
        order this runs after the last backward sweep that read them)."""
        if self.cond_graph is not None:
            self._pix.copy_(clip.to(self.dev, non_blocking=True))
            self._draw(clip.shape[0], clip.shape[1], clip.shape[3], clip.shape[4], into=self._draws)
            self.cond_graph.replay()
            src = self._cond_out
        else:
            src = self.prepare_batch(clip)
        for k in BATCH_KEYS:
            self.batches[j][k].copy_(src[k])

    # ---- the pipeline ----------------------------------------------------------------------------------------------------------
    @staticmethod
    def _as_list(clips) -> List[torch.Tensor]:
        return [clips] if isinstance(clips, torch.Tensor) else list(clips)

    def start(self, clips) -> None:
        """First clip(s) of the run: conditioners (eager), then capture of the step on the tensors they produced and of the conditioners."""
        clips = self._as_list(clips)
        if len(clips) != self.tr.grad_accum:
            raise ValueError(f"expected {self.tr.grad_accum} clip(s) per optimizer step, got {len(clips)}")
        self.batches = [self.prepare_batch(c) for c in clips]
        if self.use_graph:
            # GraphedStep's warm-up pass is one real optimizer step on this batch; the loop counts it as step 1 (see step())
            self.graphed = GraphedStep(self.tr, self.batches)
            self.graphed.opt_beside_side_work = self.overlap_optimizer
            self._after_step()
            self._warm = True
            if self.graph_conditioners:
                self._capture_conditioners(clips[0])
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
        if nxt is not None and len(nxt) != self.tr.grad_accum:
            raise ValueError(f"expected {self.tr.grad_accum} clip(s) per optimizer step, got {len(nxt)}")

        def side():
            if nxt is not None:
                for j, c in enumerate(nxt):
                    self._produce(j, c)

        if self._warm:                                   # the capture's warm-up pass already was this step
            self._warm = False
            side()
        else:
            if self.graphed is not None:
                self.graphed(side_work=side)
            else:
                self.tr.step(self.batches if self.tr.grad_accum > 1 else self.batches[0], side_work=side)
            self._after_step()
        return float(self._loss)                         # the iteration's one host synchronisation (train_svd.py:1039-1041)
