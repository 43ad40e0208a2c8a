"""bench.py -- train-step samples/sec of the MI355X-native SVD UNet (BASELINE.json metric).

A "step" = one pass of the hot path over one synthetic clip per rank: zero grads -> UNet forward -> EDM loss
-> hand-written backward -> [gradient all-reduce over RCCL when N > 1] -> inf-check + AdamW + re-pack of the
trainable weights (the 397.6 M `temporal_transformer_block` parameters of train_svd.py:761-766).  Workload at
N=1: BASELINE.json configs[1] -- SVD UNet (1,524,623,082 params), 14 frames, 512x320 (latent 64x40), fp16,
batch 1, random-init weights, synthetic latents already resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (see the driver contract).  Extra objects:
  roofline     -- the dominant kernel family (MFMA GEMM / implicit convolution / weight gradient): algorithmic FLOPs of all its launches in
                  one step / their summed duration.  Clock at one rank under hipGraph: device timestamps written by one-lane `svdx_stamp`
                  kernels around every launch inside a second capture of the step (replay conditions; `clock` says so); otherwise HIP
                  events on the launch stream around the launches of one eager step (also reported as `kernel_ms_per_step_events`).
                  `traffic` = fabric bytes per launch from profiles/pmc_traffic.json (rocprofv3 counter passes of this command);
                  `temporal_self_attention` = the north-star op for all 16 temporal blocks, forward and backward
  cpu_baseline -- the CPU oracle train step timed on the host cores on a bounded sample (c1': 8 frames 256x192), plus one oracle step at
                  the benched shape when the host allows; both batches also go through the HIP path (parity_full_model / parity_c2)
  config.gpu_clock -- shader clock and socket power of THIS device sampled during the timed region (a slow box is recognisable)
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
STEP_TFLOP_C2 = 24.80          # 2x fwd (10.707 T) + dW of the trainable set, 14x512x320, B=1


class ClockSampler:
    """Shader clock / socket power of the benched GPU while the timed region runs (box-to-box spread is +-5 %: a slow box shows as a
    lower sustained sclk).  Reads the amdgpu sysfs nodes of the device (`pp_dpm_sclk`: the level marked `*`; hwmon `power1_average` /
    `power1_input`, microwatts) from a background thread every 0.5 s; when sysfs has no such node, ONE `rocm-smi --showclocks
    --showpower` call in the middle of the region."""

    def __init__(self, index: int):
        import glob
        import threading
        self.samples, self.power, self.source = [], [], None
        self._stop = threading.Event()
        # the box holds other GPUs (other tenants): find THIS device's node by its PCI address, not by card number
        self.node = None
        try:
            pr = torch.cuda.get_device_properties(index)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            cand = f"/sys/bus/pci/devices/{addr}/pp_dpm_sclk"
            self.node = cand if os.path.exists(cand) else None
            self.pci = addr
            # the DRM card / render node of THIS device (an outside sampler that reads /sys/class/drm/card0 may be looking at another
            # tenant's GPU: the record names the card it should read)
            drm = glob.glob(f"/sys/bus/pci/devices/{addr}/drm/card*") + glob.glob(f"/sys/bus/pci/devices/{addr}/drm/renderD*")
            self.drm = sorted(os.path.basename(d) for d in drm) or None
        except Exception:  # noqa: BLE001
            self.pci = None
        self.drm = getattr(self, "drm", None)
        self.pnode = None
        if self.node:
            hw = glob.glob(os.path.join(os.path.dirname(self.node), "hwmon", "hwmon*", "power1_average")) + \
                 glob.glob(os.path.join(os.path.dirname(self.node), "hwmon", "hwmon*", "power1_input"))
            self.pnode = hw[0] if hw else None
        self._t = threading.Thread(target=self._run, daemon=True)

    def _read_sysfs(self):
        try:
            for line in open(self.node):
                if line.rstrip().endswith("*"):
                    self.samples.append(float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip()))
                    self.source = "sysfs pp_dpm_sclk"
            if self.pnode:
                self.power.append(float(open(self.pnode).read()) / 1e6)
        except Exception:  # noqa: BLE001
            pass

    def _read_smi(self):
        import re
        import subprocess
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
            m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            if m:
                self.samples.append(float(m.group(1)))
                self.source = "rocm-smi --showclocks"
            m = re.search(r"(?:Socket|Average) Graphics Package Power \(W\): ([0-9.]+)", out)
            if m:
                self.power.append(float(m.group(1)))
        except Exception:  # noqa: BLE001
            pass

    def _run(self):
        if self.node:
            while not self._stop.wait(0.5):
                self._read_sysfs()
        if not self.samples and not self._stop.wait(1.0):
            self._read_smi()

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._t.join(timeout=25)

    def summary(self):
        if not self.samples:
            return None
        v = sorted(self.samples)
        out = {"sclk_mhz_median": v[len(v) // 2], "sclk_mhz_min": v[0], "sclk_mhz_max": v[-1], "samples": len(v), "source": self.source,
               "pci": self.pci, "drm_nodes": self.drm, "max_sclk_mhz": 2400}
        if self.power:
            w = sorted(self.power)
            out["power_w_median"] = w[len(w) // 2]
        return out


def init_weights_(model: torch.nn.Module, seed: int) -> None:
    """Random init with O(1) activations (same law as oracle.unet.scaled_init_, drawn on device)."""
    g = torch.Generator(device=next(model.parameters()).device).manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("mix_factor"):
                p.fill_(0.5)
            elif p.ndim == 1:
                if "norm" in name and name.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g, device=p.device))
                elif "norm" in name:
                    p.copy_(0.1 * torch.randn(p.shape, generator=g, device=p.device))
                else:
                    p.copy_(0.02 * torch.randn(p.shape, generator=g, device=p.device))
            else:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g, device=p.device) * math.sqrt(1.0 / fan_in))


def make_batch(B, T, h, w, cross_dim, seed, dev):
    from svd_xtend_amd.train import edm_prepare
    g = torch.Generator(device="cpu").manual_seed(seed)
    latents = 0.7 * torch.randn(B, T, 4, h, w, generator=g)
    noise = torch.randn(B, T, 4, h, w, generator=g)
    cond = torch.randn(B, 4, h, w, generator=g)
    ehs = torch.randn(B, 1, cross_dim, generator=g)
    u = torch.rand(B, generator=g) * (1 - 2e-7) + 1e-7
    cond_sigma = torch.distributions.Normal(-3.0, 0.5).icdf(u).exp()          # train_svd.py:954
    u = torch.rand(B, generator=g) * (1 - 2e-7) + 1e-7
    sigmas = torch.distributions.Normal(0.7, 1.6).icdf(u).exp()               # train_svd.py:964
    unet_in, ts, noisy = edm_prepare(latents, noise, cond, sigmas)
    ids = torch.tensor([[7.0, 127.0, float(cond_sigma[0])]]).repeat(B, 1)     # train_svd.py:981-988
    return dict(unet_in=unet_in.to(dev), timesteps=ts.to(dev), ehs=ehs.to(dev), added_time_ids=ids.to(dev),
                noisy_latents=noisy.to(dev), target=latents.to(dev), sigmas=sigmas.to(dev))


def cpu_baseline(dev=None, dtype=torch.float16, c2=True):
    """The oracle (kind 'port': pure-PyTorch restatement; diffusers is not installed) on the host cores.
    Sample: c1' = one 8-frame 256x192 clip, fp32, full SVD UNet, fwd + loss + bwd + AdamW: 1 warm-up step, then the median of 3.
    When the host has the cores and the memory, ONE step of the benched shape itself (c2, 14 x 512x320) follows on the same model.
    Each oracle batch also goes through the HIP path on the oracle's weights (`parity_full_model`, `parity_c2`): the checker role of
    the oracle at the FULL 1.52 B-parameter topology, beside the parity tests of tests/."""
    import statistics

    from oracle.step import edm_inputs, make_optimizer, make_synthetic_batch, train_step
    from oracle.unet import SVD_CONFIG, UNetSpatioTemporalConditionOracle, no_default_init, scaled_init_
    torch.manual_seed(0)
    # Threads: the oracle is memory-bound torch fp32 work and PyTorch's default (one thread per physical core: 128 on the GPU box's 2 x 64-core
    # host) is far from its best there -- one c1' step takes 4.6 s on 16 threads, 7 s on 32, 12 s on 64 and 20.9 s on 128; one step of the
    # 64x40-level block 30 s on 32 threads, 43 s on 64, 81 s on 128 (profiles/r6b_oracle_threads.txt, r6c_oracle_threads_c1.txt).  The baseline
    # is timed at the fastest setting measured, and `cores` states the threads actually used.
    host_threads = torch.get_num_threads()
    cores = min(host_threads, 16)
    cores_c2 = min(host_threads, 32)
    torch.set_num_threads(cores)
    t0 = time.time()
    with no_default_init():                         # scaled_init_ fills every parameter: skip 1.52 B kaiming draws on one core
        orc = UNetSpatioTemporalConditionOracle(**SVD_CONFIG)
    scaled_init_(orc, 0)                            # O(1) activations through all 4 levels: the prediction matters in the loss
    opt = make_optimizer(orc, lr=0.0)               # lr 0: the timed steps do the full AdamW arithmetic but leave the weights where
    batch = make_synthetic_batch(1, 8, 24, 32, 1)   # the HIP model copied them, so every step is the same step
    t_build = time.time() - t0
    prod = None
    perr = None
    if dev is not None:
        try:
            from svd_xtend_amd.train import Trainer
            from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
            with torch.device(dev):
                pm = UNetSpatioTemporalConditionModel(**SVD_CONFIG)
            pm.load_state_dict(orc.state_dict(), strict=True)
            prod = Trainer(pm, dtype=dtype, lr=1e-5)
        except Exception as e:  # noqa: BLE001
            perr = {"error": repr(e)[:200]}

    def hip_loss(b, loss_ref, pred_ref):
        if prod is None:
            return perr
        try:
            unet_in, ts, ehs, ids, noisy, _ = edm_inputs(b)
            with torch.no_grad():
                pred = prod.model(unet_in.to(dev), ts.to(dev), ehs.to(dev), ids.to(dev)).sample.float().cpu()
            prod.zero_grad()
            prod.forward_backward(unet_in.to(dev), ts.to(dev), ehs.to(dev), ids.to(dev), noisy.to(dev), b["latents"].to(dev),
                                  b["sigmas"].to(dev))
            lg = float(prod.last_loss())
            # two bars, each next to the number it applies to: north_star's metric is the LOSS (noise-prediction MSE, <= 1e-3 relative
            # in fp16); the prediction tensor itself -- 16-bit storage through ~100 chained layers -- is held to the bar of
            # tests/e2e_checks.assert_parity (a quarter above the worst case ever measured)
            fp16 = dtype == torch.float16
            loss_rel = abs(lg - float(loss_ref)) / abs(float(loss_ref))
            pred_rel = float((pred - pred_ref).norm() / pred_ref.norm())
            loss_tol, pred_bar = (1e-3, 3e-3) if fp16 else (8e-3, 2e-2)
            return {"loss_oracle_fp32": float(loss_ref), "loss_hip": lg, "loss_rel_err": loss_rel, "loss_rel_tolerance": loss_tol,
                    "pred_rel_l2": pred_rel, "pred_rel_l2_bar": pred_bar, "within_tolerance": loss_rel <= loss_tol and pred_rel <= pred_bar,
                    "dtype": str(dtype).split(".")[-1]}
        except Exception as e:  # noqa: BLE001
            return {"error": repr(e)[:200]}

    times = []
    for i in range(4):                              # step 0 = warm-up (allocator, thread pools), then 3 timed
        t1 = time.time()
        loss, pred = train_step(orc, batch, opt)
        times.append(time.time() - t1)
    dt = statistics.median(times[1:])
    parity = hip_loss(batch, loss, pred)
    out = {"value": 1.0 / dt, "unit": "samples/s", "cores": cores, "kind": "port",
           "sample": "c1': the full SVD UNet on one 8-frame 256x192 clip (latent 24x32), fp32, fwd+loss+bwd+AdamW; 1 warm-up step "
                     f"({times[0]:.1f}s) + median of 3 ({dt:.1f}s; +{t_build:.1f}s model build); 4.1 TFLOP/step",
           "seconds": dt, "seconds_all": times, "loss": float(loss), "parity_full_model": parity}
    # one step at the benched shape (c2) when the host can afford it: ~25 TFLOP of fp32 work and ~60 GB of saved activations
    try:
        import psutil
        avail = psutil.virtual_memory().available / 2 ** 30
    except Exception:  # noqa: BLE001
        avail = 0.0
    if c2 and host_threads >= 32 and avail >= 110.0 and dt <= 45.0:
        b2 = make_synthetic_batch(1, 14, 40, 64, 2)
        torch.set_num_threads(cores_c2)
        t1 = time.time()
        loss2, pred2 = train_step(orc, b2, opt)
        t2 = time.time() - t1
        out["c2"] = {"value": 1.0 / t2, "unit": "samples/s", "seconds": t2, "loss": float(loss2), "cores": cores_c2,
                     "sample": "c2: ONE un-warmed step of the benched shape (14 frames 512x320, latent 40x64), same model, fp32; 24.8 TFLOP",
                     "parity_c2": hip_loss(b2, loss2, pred2)}
    else:
        out["c2"] = {"skipped": f"needs >= 32 cores, >= 110 GB free RAM and a c1' step <= 45 s (have {host_threads} cores, {avail:.0f} GB, {dt:.1f} s)"}
    torch.set_num_threads(host_threads)
    out["host_threads_default"] = host_threads
    del prod
    if dev is not None:
        torch.cuda.empty_cache()
    return out


def real_loop_leg(args, trainer, dev, dt, world, rank, T):
    """The step as a training loop runs it (svd_xtend_amd.loop.TrainLoop = the loop body of train_svd.py:931-1058): every iteration a NEW
    pixel clip arrives from pinned host memory, goes through the frozen conditioners (VAE encode of T + 1 frames, CLIP ViT-H embed of the first
    frame, EDM noising / dropout on the device) -- queued between the backward sweep and the optimizer of the running step, i.e. beside the
    gradient all-reduce on several ranks (north_star's schedule) -- its batch is copied into the captured tensors, the step is replayed and
    the loss is read on the host.  The UNet-only headline above replays ONE resident batch; this is the end-to-end figure next to it."""
    from svd_xtend_amd.clip import CLIPVisionModelWithProjection
    from svd_xtend_amd.loop import TrainLoop
    from svd_xtend_amd.vae import AutoencoderKLTemporalDecoder
    with torch.device(dev):
        if args.tiny:
            vae = AutoencoderKLTemporalDecoder(block_out_channels=(64, 128, 128, 128), layers_per_block=1)
            enc = CLIPVisionModelWithProjection(hidden_size=320, intermediate_size=640, projection_dim=trainer.model.config.cross_attention_dim,
                                                num_hidden_layers=2, num_attention_heads=4, image_size=56, patch_size=14)
        else:
            vae, enc = AutoencoderKLTemporalDecoder(), CLIPVisionModelWithProjection()
    init_weights_(vae, seed=4321)
    init_weights_(enc, seed=4322)
    for m in (vae, enc):
        m.requires_grad_(False)
        m.prepare(dt)
    g = torch.Generator().manual_seed(777 + rank)
    host_clips = [(torch.rand(1, T, 3, args.height, args.width, generator=g) * 2 - 1).pin_memory() for _ in range(4)]
    ga = args.grad_accum
    pick = lambda i: [host_clips[(i * ga + j) % len(host_clips)] for j in range(ga)]           # noqa: E731
    loop = TrainLoop(trainer, vae, enc, conditioning_dropout_prob=0.1, seed=99 + rank, use_graph=not args.no_graph, overlap_clip=not args.serial_conditioners,
                     overlap_optimizer=not args.serial_optimizer)
    loop.start(pick(0))
    it = 1
    for _ in range(3):
        loop.step(pick(it))
        it += 1
    K = max(5, min(args.steps, 30))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = []
    for _ in range(K):
        losses.append(loop.step(pick(it)))
        it += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        te = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        el = float(te)
    # the conditioners on their own (same eager launches, nothing beside them)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(5):
        loop.prepare_batch(host_clips[i % len(host_clips)])
    e1.record()
    torch.cuda.synchronize()
    # ... and split: the VAE encode of T + 1 frames and the CLIP embed each on their own, with the matmul FLOPs their launches carry (2 M N K of
    # every svdx_gemm* call of one pass, read from the binding's launch log) against the dense MFMA peak
    def timed(fn, n=5):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n

    def gemm_flops(fn):
        k = trainer.rt.k
        k.launch_log = log = []
        try:
            fn()
        finally:
            k.launch_log = None
        return sum(2.0 * a[3] * a[4] * a[5] for name, a, _ in log if name.startswith("svdx_gemm") and name not in ("svdx_gemm_finalize", "svdx_gemm_finalize_gn")), len(log)

    from svd_xtend_amd.clip import encode_image
    pix = host_clips[0].to(dev).to(torch.float32)
    frames = torch.cat([pix, pix[:, 0:1]], dim=1).reshape(T + 1, *pix.shape[2:])
    with torch.no_grad():
        vae_fn, clip_fn = (lambda: vae.encode(frames).latent_dist), (lambda: encode_image(pix[:, 0], enc))
        vae_ms, clip_ms = timed(vae_fn), timed(clip_fn)
        (vae_fl, vae_n), (clip_fl, clip_n) = gemm_flops(vae_fn), gemm_flops(clip_fn)
    cond = {"vae_encode": {"ms": vae_ms, "launches": vae_n, "gemm_tflop": vae_fl / 1e12, "frac_of_mfma_peak": vae_fl / 1e12 / (vae_ms * 1e-3) / MFMA_PEAK_TFLOPS,
                           "what": f"AutoencoderKLTemporalDecoder.encode of {T + 1} frames {args.width}x{args.height}"},
            "clip_embed": {"ms": clip_ms, "launches": clip_n, "gemm_tflop": clip_fl / 1e12, "frac_of_mfma_peak": clip_fl / 1e12 / (clip_ms * 1e-3) / MFMA_PEAK_TFLOPS,
                           "what": "resize to 224 + CLIP ViT-H/14 tower + projection of the first frame"}}
    del pix, frames
    out = {"ms_per_step": el / K * 1e3, "value": world * ga * K / el, "unit": "samples/s", "steps": K,
           "conditioners_ms_alone": e0.elapsed_time(e1) / 5, "conditioners": cond,
           "what": "TrainLoop.step: new pixel clip from pinned host memory -> VAE encode (T + 1 frames) with the CLIP image embed beside it on a second stream (serial with --serial-conditioners) + EDM noising on the "
                   "device, queued between this step's backward sweep and its optimizer (beside the gradient all-reduce when N > 1) -> batch copied "
                   "into the captured tensors -> hipGraph replay -> loss read on the host (one host sync per step)",
           "loss_first": losses[0], "loss_last": losses[-1], "losses_finite": all(l == l and abs(l) != float("inf") for l in losses),
           "host_syncs_per_step": 1, "exec": "hipgraph" if loop.graphed is not None else "eager"}
    del loop, vae, enc
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # ~11 s of GPU work: long enough for an outside power / utilisation sampler to see it
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--height", type=int, default=320)     # pixels; README.md:40-53 trains 512x320
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--gemm-variant", type=int, default=int(os.environ.get("SVDX_GEMM_VARIANT", "4")))
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--lora-rank", type=int, default=0,
                    help="reference config 5 (train_svd_lora.py): LoRA adapters of this rank on every attention projection are the "
                         "trainable set (the reference runs it in bf16: pass --dtype bf16); 0 = config 2 (default)")
    ap.add_argument("--lora-param-dtype", default=None, choices=[None, "reference"],
                    help="with --lora-rank and --dtype bf16: 'reference' keeps adapters and AdamW state as bf16 numbers, torch.optim.AdamW's op "
                         "sequence (train_svd_lora.py:666-674); default: fp32 masters")
    ap.add_argument("--tune", action="store_true",
                    help="in-situ GEMM tile/split-K tuning sweeps before timing (faster while the GPU is cool, ~1%% slower "
                         "than the built-in formula once the step is power-limited: off by default)")
    ap.add_argument("--tune-rounds", type=int, default=1, help="--tune: sweeps over the candidate list of the largest problem")
    ap.add_argument("--gemm-table", action="store_true", help="dump per-shape GEMM timings of one step")
    ap.add_argument("--tiny", action="store_true", help="debug: tiny topology instead of the SVD config")
    ap.add_argument("--grad-accum", type=int, default=1,
                    help="micro-batches per optimizer step (reference config 4 runs gradient_accumulation_steps = 2); gradients are "
                         "reduced over ranks on the last one only")
    ap.add_argument("--overlap", default="auto", choices=["auto", "all", "buckets", "single", "direct", "vae"],
                    help="N > 1, how the gradient sum over ranks is scheduled.  `auto` (default): an untimed probe of a few steps per "
                         "candidate -- `single`, `buckets` and, when its result agrees with RCCL's on this node, `direct` -- picks the fastest, "
                         "which is then timed; every candidate's probe is reported in config.schedules.  `all`: the same with longer probes.  "
                         "`single` = ONE RCCL all-reduce of the flat buffer after the sweep; `buckets` = one RCCL all-reduce per transformer block "
                         "started during the backward sweep; `direct` = svdx_allreduce_grads (reduce-scatter + all-gather over peer-mapped "
                         "buffers, all seven xGMI links at once) after the sweep; `vae` = `single` with the VAE encode of the NEXT micro-batch "
                         "(train_svd.py:948) replayed beside it and AdamW after the wait.  north_star's schedule with the real conditioners "
                         "(VAE + CLIP + EDM prep of a new clip every step) is the `real_loop` object of every run")
    ap.add_argument("--no-overlap", action="store_true", help="alias of --overlap single")
    ap.add_argument("--no-cpu-c2", action="store_true", help="skip the single CPU-oracle step at the benched shape")
    ap.add_argument("--with-vae", action="store_true",
                    help="also time the step with the VAE encode of the next micro-batch (train_svd.py:948, 957-960) on a second "
                         "stream; reported as a second field, the headline metric stays UNet-only")
    ap.add_argument("--rt", action="append", default=[], metavar="NAME=VALUE", help="set an ops.Runtime switch (developer A/B between two processes), e.g. --rt big_m_rules=0")
    ap.add_argument("--serial-optimizer", action="store_true", help="real loop: the optimizer after the next clip's conditioners instead of beside them (A/B of TrainLoop(overlap_optimizer))")
    ap.add_argument("--serial-conditioners", action="store_true", help="real loop: CLIP embed after the VAE encode on one stream (A/B of TrainLoop(overlap_clip))")
    ap.add_argument("--no-real-loop", action="store_true",
                    help="skip the `real_loop` leg (the step inside svd_xtend_amd.loop.TrainLoop: a new pixel clip through VAE + CLIP + EDM prep every "
                         "iteration, loss read on the host)")
    ap.add_argument("--launch-log", default=None,
                    help="developer aid: write the ordered list of libsvdx calls of ONE eager step (entry name + scalar arguments) to this "
                         "JSON file; tools/step_trace.py joins it with a rocprofv3 kernel trace of the graph-replayed steps")
    args = ap.parse_args()
    if args.no_overlap:
        args.overlap = "single"
    if args.overlap == "vae":
        args.with_vae = True
    if args.gpus > 1 and "NCCL_DEBUG" not in os.environ:
        # what RCCL decided (rings / trees, channels, algorithm and protocol per size) goes to a per-process file, never to stdout
        os.environ.update(NCCL_DEBUG="INFO", NCCL_DEBUG_SUBSYS="INIT,GRAPH,TUNING", NCCL_DEBUG_FILE="/tmp/svdx_rccl_%h_%p.log")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher the driver contract describes -- one rank per GPU under
        # torch.distributed.run on this node (127.0.0.1 rendezvous), same arguments
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # developer knobs (single-GPU rehearsal of the multi-rank path: all ranks on one device over gloo; RCCL refuses that)
    backend = os.environ.get("SVDX_DIST_BACKEND", "nccl")
    if os.environ.get("SVDX_BENCH_DEVICE") == "cpu":
        # host-logic rehearsal without a GPU (tests/bench_cpu_harness.py: the kernel emulation of the test suite is installed as the backend
        # first; without that, kernels.backend() raises as everywhere -- this is not a CPU mode of the product)
        dev = torch.device("cpu")
    else:
        if os.environ.get("SVDX_BENCH_DEVICE") is not None:
            local_rank = int(os.environ["SVDX_BENCH_DEVICE"])
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from svd_xtend_amd.train import GraphedStep, Trainer
    from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    cfg = {}
    if args.tiny:
        cfg = dict(block_out_channels=(64, 128, 128, 128), addition_time_embed_dim=32,
                   projection_class_embeddings_input_dim=96, cross_attention_dim=64, num_attention_heads=(1, 2, 2, 2))
    with torch.device(dev):
        model = UNetSpatioTemporalConditionModel(**cfg)
    init_weights_(model, seed=1234)            # identical on every rank (DDP's init broadcast)
    if args.lora_rank:
        from svd_xtend_amd.lora import LoraConfig
        for p in model.parameters():           # train_svd_lora.py:655-671
            p.requires_grad_(False)
        with torch.device(dev):
            model.add_adapter(LoraConfig(r=args.lora_rank, lora_alpha=args.lora_rank, init_lora_weights="gaussian"))
    trainer = Trainer(model, dtype=dt, lr=1e-5, grad_accum=args.grad_accum, lora_param_dtype=args.lora_param_dtype)
    for kv in args.rt:                                   # developer A/B of a Runtime switch between two processes (configurations whose two graphs
        name, _, val = kv.partition("=")                 # do not fit one process: tools/ab_inproc.py is the same-process form)
        if not hasattr(trainer.rt, name):
            raise SystemExit(f"--rt {kv}: ops.Runtime has no attribute {name!r}")
        setattr(trainer.rt, name, type(getattr(trainer.rt, name))(int(val)) if val.lstrip("-").isdigit() else val)
    trainer.rt.gemm_variant = args.gemm_variant
    schedules = {}

    def set_schedule(name):
        """single / vae: one RCCL collective; buckets: per-block RCCL collectives under the sweep; direct: svdx_allreduce_grads"""
        trainer.overlap = name == "buckets"
        trainer.use_direct_allreduce(name == "direct")

    if world > 1 and args.overlap not in ("auto", "all"):
        set_schedule(args.overlap)
    n_params = sum(p.numel() for p in model.parameters())
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    B, T, h, w = 1, args.frames, args.height // 8, args.width // 8
    cross = model.config.cross_attention_dim
    # rank-distinct data, one clip per micro-batch
    batches = [make_batch(B, T, h, w, cross, seed=123 + rank + 1000 * i, dev=dev) for i in range(args.grad_accum)]
    batch = batches[0]

    def fwd_bwd():
        trainer.zero_grad()
        for b in batches:
            trainer.forward_backward(**b)

    def opt_step():
        trainer.optimizer_step()

    def step_eager(side_work=None):
        fwd_bwd()
        trainer.finish_grads(side_work)
        opt_step()

    # ---- one-off GEMM tuning (untimed set-up, like graph capture), warmup (eager), then capture -------------
    tune_sweeps = 0
    if args.tune and args.gemm_variant == 4:
        tune_sweeps = trainer.tune_gemms(batch, rounds=args.tune_rounds, max_steps=400)
    for _ in range(max(1, args.warmup)):
        step_eager()
    torch.cuda.synchronize()
    if args.launch_log and rank == 0:
        trainer.rt.k.launch_log = []
        step_eager()
        torch.cuda.synchronize()
        os.makedirs(os.path.dirname(os.path.abspath(args.launch_log)), exist_ok=True)
        with open(args.launch_log, "w") as f:
            json.dump(trainer.rt.k.launch_log, f)
        trainer.rt.k.launch_log = None
    # ---- N > 1: which schedule of the gradient sum is fastest HERE (untimed probe; every rank takes the same decision) -------------
    direct_check = None
    if world > 1 and args.overlap in ("auto", "all"):
        cands = ["single", "buckets"]
        # collective: IPC handles of every rank's gradient buffer, then Trainer's own agreement probe (two rounds of fresh random data
        # against the library's sum, MIN-voted over the ranks: a rank-local failure turns into "everyone stays on RCCL", never a hang)
        if trainer.use_direct_allreduce(True):
            cands.append("direct")
        direct_check = dict(getattr(trainer, "direct_check", None) or {})
        direct_check["agrees_with_rccl"] = bool(direct_check.get("agrees_with_library", False))
        trainer.use_direct_allreduce(False)
        n_probe = 24 if args.overlap == "all" else 8
        for name in cands:
            set_schedule(name)
            try:
                sg = step_eager if args.no_graph else GraphedStep(trainer, batches)
                for _ in range(2):
                    sg()
                trainer.exposed_events = []
                dist.barrier()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(n_probe):
                    sg()
                torch.cuda.synchronize()
                dist.barrier()
                te = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
                dist.all_reduce(te, op=dist.ReduceOp.MAX)
                ex = trainer.exposed_events
                schedules[name] = {"ms_per_step": float(te) / n_probe * 1e3, "steps": n_probe,
                                   "allreduce_ms_exposed": (sum(a.elapsed_time(b) for a, b in ex) / len(ex)) if ex else None}
            except Exception as e:  # noqa: BLE001
                schedules[name] = {"error": repr(e)[:200]}
            trainer.exposed_events = None
            sg = None
            torch.cuda.empty_cache()
        ok_names = [n_ for n_ in cands if "ms_per_step" in schedules.get(n_, {})]
        best = min(ok_names, key=lambda n_: schedules[n_]["ms_per_step"]) if ok_names else "single"
        set_schedule(best)
        args.overlap = best
    exec_mode = "eager"
    step = step_eager
    if not args.no_graph:
        try:
            # chain of graph segments cut at the transformer blocks: each block's gradient slice starts its all-reduce
            # (eager RCCL call between two replays) while the rest of the backward sweep runs -- svd_xtend_amd.train.GraphedStep
            step_graph = GraphedStep(trainer, batches)
            step_graph()
            torch.cuda.synchronize()
            step = step_graph
            exec_mode = "hipgraph"
        except Exception as e:  # noqa: BLE001 - fall back to eager launches, say so in the JSON line
            print(f"[bench] graph capture failed ({e!r}); running eager", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            step = step_eager

    # ---- timed region ------------------------------------------------------------------------------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    sampler = ClockSampler(local_rank) if rank == 0 and os.environ.get("SVDX_NO_CLOCK_SAMPLER") != "1" else None
    trainer.exposed_events = [] if world > 1 else None     # two event records per step: what of the collective nothing hid
    t0 = time.perf_counter()
    if sampler is not None:
        sampler.__enter__()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ex = trainer.exposed_events
    trainer.exposed_events = None
    exposed_ms = (sum(a.elapsed_time(b) for a, b in ex) / len(ex)) if ex else None
    if sampler is not None:
        sampler.__exit__()
    if world > 1:
        te = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te)
    loss = float(trainer.last_loss())
    state = trainer.opt_state.cpu().tolist()
    ms = elapsed / args.steps * 1e3
    value = world * B * args.grad_accum * args.steps / elapsed

    # ---- the same step with the VAE encode of the NEXT micro-batch running beside it (train_svd.py:948, 957-960) ----------------
    vae_info = None
    if args.with_vae:
        from svd_xtend_amd.vae import AutoencoderKLTemporalDecoder
        with torch.device(dev):
            vae = AutoencoderKLTemporalDecoder() if not args.tiny else AutoencoderKLTemporalDecoder(block_out_channels=(64, 128, 128, 128), layers_per_block=1)
        init_weights_(vae, seed=4321)
        vae.prepare(dt)
        frames = B * (T + 1)                            # the clip's T frames + its noised conditioning frame
        pix = torch.rand(frames, 3, args.height, args.width, device=dev) * 2 - 1
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                z = vae.encode(pix).latent_dist.sample() * vae.config.scaling_factor
            torch.cuda.synchronize()
            gv = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gv, stream=side):
                z = vae.encode(pix).latent_dist.sample() * vae.config.scaling_factor
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(side):
            e0.record()
            for _ in range(5):
                gv.replay()
            e1.record()
        torch.cuda.synchronize()
        vae_alone = e0.elapsed_time(e1) / 5
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        trainer.exposed_events = [] if world > 1 else None
        for _ in range(args.steps):
            if args.overlap == "vae":
                step(side_work=gv.replay)                # ONE collective in flight, the next clip's encode queued beside it, AdamW after
            else:
                with torch.cuda.stream(side):
                    gv.replay()                          # frozen, no dependency on the update: free to run under the all-reduce / AdamW
                step()
        torch.cuda.synchronize()
        exposed = trainer.exposed_events
        trainer.exposed_events = None
        if world > 1:
            dist.barrier()
        el2 = time.perf_counter() - t1
        if world > 1:
            te = torch.tensor([el2], device=dev, dtype=torch.float64)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el2 = float(te)
        vae_info = {"ms_per_step": el2 / args.steps * 1e3, "value": world * B * args.grad_accum * args.steps / el2, "unit": "samples/s",
                    "vae_ms_alone": vae_alone, "frames_encoded_per_step": frames,
                    "what": "UNet step + AutoencoderKLTemporalDecoder.encode of the next clip (T + 1 frames, pixels resident in HBM) on a second "
                            "HIP stream, both replayed from hipGraphs; the headline `value` above is the UNet step alone",
                    "latent_mean_abs": float(z.abs().mean()),
                    "schedule": ("one all-reduce after the backward sweep, the encode replayed beside it on the compute stream, AdamW after the wait"
                                 if args.overlap == "vae" else "encode on a second stream beside the whole step"),
                    "allreduce_ms_exposed": (sum(a.elapsed_time(b) for a, b in exposed) / len(exposed)) if exposed else None}
        del vae, gv, pix

    # ---- the step as a training loop runs it ------------------------------------------------------------------------------------
    real = None
    if not args.no_real_loop:
        try:
            real = real_loop_leg(args, trainer, dev, dt, world, rank, T)
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            real = {"error": repr(e)[:300]}
        torch.cuda.synchronize()

    # ---- what RCCL saw: ranks (an all-reduce of ones) and the cost of the gradient exchange on its own -------------------------
    ranks_seen, allreduce_ms, direct_ms, rccl = 1, None, None, None
    if world > 1:
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        gbuf = torch.zeros_like(trainer.g_flat)
        for _ in range(2):
            dist.all_reduce(gbuf)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            dist.all_reduce(gbuf)
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - t1) / 5 * 1e3
        del gbuf
        # the same exchange as svdx_allreduce_grads (direct reduce-scatter + all-gather over peer-mapped buffers), when this node maps peers
        direct_ms = None
        try:
            had = trainer.direct is not None
            trainer.use_direct_allreduce(True)
            keep = trainer.g_flat.clone()
            for _ in range(2):
                trainer.direct.all_reduce()
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            for _ in range(5):
                trainer.direct.all_reduce()
            torch.cuda.synchronize()
            direct_ms = (time.perf_counter() - t1) / 5 * 1e3
            trainer.g_flat.copy_(keep)
            del keep
            if not had:
                trainer.use_direct_allreduce(False)
        except Exception as e:  # noqa: BLE001
            direct_ms = {"error": repr(e)[:160]}
        rccl = {"version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None}
        try:
            import glob
            import re
            lines = []
            for f in sorted(glob.glob(f"/tmp/svdx_rccl_*_{os.getpid()}.log")):
                lines += open(f, errors="replace").read().splitlines()
            pick = [ln for ln in lines if re.search(r"Algo|algorithm|proto|Connected all|[Cc]hannels|Ring \d|Tree", ln)]
            seen, uniq = set(), []
            for ln in pick:
                key = re.sub(r"\[\d+\]|0x[0-9a-f]+|\d+:\d+:\d+", "", ln)[-160:]
                if key not in seen:
                    seen.add(key)
                    uniq.append(ln[-220:])
            rccl["debug_lines"] = uniq[:24]
        except Exception as e:  # noqa: BLE001
            rccl["debug_lines"] = [repr(e)[:120]]

    # ---- roofline of the dominant kernel (one instrumented eager step; events on the launch stream) ---------
    roof = None
    # Every rank runs the instrumented step (same collectives on all ranks -- a rank-0-only step would leave its all-reduces without
    # peers); only rank 0 installs the timing hooks and reports.
    if not args.no_roofline:
        k = trainer.rt.k
        orig, orig_tn = k.gemm, k.gemm_tn
        recs, recs_bytes = [], []
        # Two clocks.  `stamps` (one rank, graph mode): every launch of the family is bracketed by svdx_stamp -- a one-lane kernel that writes
        # the device wall clock -- INSIDE a second capture of the step, so the durations are those of the replayed step (the kernel
        # boundary a stamp adds is calibrated from back-to-back stamps and subtracted).  `events` (eager fallback, and the multi-rank
        # path whose collectives cannot be captured): HIP events around each eager launch, which also time ~5 us of marker handling per
        # launch -- round 3's method; its figure is reported beside the stamps' as `kernel_ms_per_step_events`.
        use_stamps = world == 1 and exec_mode == "hipgraph" and hasattr(k, "stamp") and k.wall_clock_khz() > 0
        slots = torch.zeros(16384, dtype=torch.int64, device=dev) if use_stamps else None
        nslot = [0]

        class Span:
            """duration of what was enqueued between begin() and end(); ms() after the measurement"""
            calib_ms, host = 0.0, None

            def __init__(self):
                self.h = None
                if use_stamps:
                    if torch.cuda.is_current_stream_capturing() and nslot[0] + 2 <= slots.numel():
                        self.h = nslot[0]
                        nslot[0] += 2
                        k.stamp(slots, self.h)
                else:
                    self.h = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    self.h[0].record()

            def end(self):
                if self.h is None:
                    return self
                if use_stamps:
                    k.stamp(slots, self.h + 1)
                else:
                    self.h[1].record()
                return self

            @property
            def live(self):
                return self.h is not None

            def ms(self):
                if use_stamps:
                    return max(0.0, float(Span.host[self.h + 1] - Span.host[self.h]) / k.wall_clock_khz() - Span.calib_ms)
                return self.h[0].elapsed_time(self.h[1])

        def timed_gemm(A, Bm, C, M, N, Kd, *a, **kw):
            sp = Span()
            orig(A, Bm, C, M, N, Kd, *a, **kw)
            if not sp.end().live:
                return
            g = kw.get("gather")
            recs.append((sp, 2.0 * M * N * Kd, ("nt", M, N, Kd, g.mode if g is not None else 0, kw.get("split_k", 1))))
            # algorithmic bytes of the launch: every operand once (the activations of a convolution once, not once per tap), the result
            # once (float slabs: 4 B x slices), the residual it adds, and what a fused GEGLU epilogue moves besides the plain result
            # (forward: h [M, F] beside pre [M, 2F]; backward: reads pre [M, 2F] and writes d(pre) [M, 2F] where the plain GEMM writes [M, F])
            taps = {0: 1, 1: 9, 2: 9, 3: 3, 4: 9}.get(g.mode if g is not None else 0, 1)
            osz = 4 * kw.get("split_k", 1) if kw.get("out_mode", 0) != 0 else 2       # float slabs vs 16-bit activations
            epi = kw.get("epilogue", 0)
            recs_bytes.append(2.0 * M * Kd / taps + 2.0 * N * Kd + osz * M * N + (2.0 * M * N if kw.get("res") is not None else 0.0)
                              + (1.0 * M * N if epi == 1 else 6.0 * M * N if epi == 2 else 0.0))

        def timed_gemm_tn(A, Bm, C, R, N, Kd, *a, **kw):
            sp = Span()
            orig_tn(A, Bm, C, R, N, Kd, *a, **kw)
            if not sp.end().live:
                return
            recs.append((sp, 2.0 * R * N * Kd, ("tn", N, Kd, R, 0, kw.get("split_k", 1))))
            recs_bytes.append(2.0 * R * N + 2.0 * R * Kd + 4.0 * N * Kd * max(1, kw.get("split_k", 1)))
        # the band kernel (fused temporal self-attention) carries projections that used to be launches of the family above; it is
        # timed beside it, not inside it
        orig_tsa = getattr(k, "tsa_fwd", None)
        band = []

        def timed_tsa(*a):
            sp = Span()
            orig_tsa(*a)
            if not sp.end().live:
                return
            Bq, Tq, HWq, Cq = a[16], a[17], a[18], a[19]
            band.append((sp, 8.0 * Bq * Tq * HWq * Cq * Cq + 4.0 * Bq * Tq * HWq * Tq * Cq))

        # the temporal self-attention OP (north_star's "temporal-attention kernel"; SURVEY 8d: LN + q/k/v + core + out-projection is the
        # only definition under which an MFMA fraction means anything): every launch between the region marks of
        # TemporalBasicTransformerBlock.fwd / .bwd, per block
        regions, open_sp = [], {}

        def on_region(name, info, begin):
            if begin:
                open_sp[name] = Span()
            else:
                sp = open_sp.pop(name).end()
                if sp.live:
                    regions.append((name, dict(info), sp))

        def install():
            k.gemm, k.gemm_tn = timed_gemm, timed_gemm_tn
            trainer.rt.on_region = on_region
            if orig_tsa is not None:
                k.tsa_fwd = timed_tsa

        def uninstall():
            k.gemm, k.gemm_tn = orig, orig_tn
            trainer.rt.on_region = None
            k.__dict__.pop("tsa_fwd", None)

        events_ms = None
        if use_stamps:
            # the stamped capture needs a memory pool of its own (as large as the timed one: ~25 GB at c2, ~100 GB at config 4's shape):
            # the timed graph has done its work -- give its pool and the allocator's cached eager blocks back first
            import gc
            step = step_graph = None        # noqa: F841
            gc.collect()
            torch.cuda.empty_cache()
            # (1) round 3's eager + events figure, for continuity; (2) the stamped capture
            use_stamps = False
            install()
            try:
                torch.cuda.synchronize()
                torch.cuda._sleep(int(0.040 * 2.1e9))
                fwd_bwd()
                torch.cuda.synchronize()
            finally:
                uninstall()
            trainer.allreduce_grads()
            opt_step()
            torch.cuda.synchronize()
            events_ms = sum(sp.ms() for sp, _, _ in recs)
            recs.clear(), recs_bytes.clear(), band.clear(), regions.clear()
            use_stamps = True
            install()
            try:
                g2 = GraphedStep(trainer, batches)         # its warm-up sweep is eager (no stamps, nothing recorded); the capture is stamped
            finally:
                uninstall()
            ncal = 32
            c0 = nslot[0]
            gc = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(gc, stream=side):
                    for i in range(ncal):
                        k.stamp(slots, c0 + 2 * i)
                        k.stamp(slots, c0 + 2 * i + 1)
            torch.cuda.current_stream().wait_stream(side)
            for _ in range(3):
                g2()
                gc.replay()
            torch.cuda.synchronize()
            Span.host = slots.cpu().tolist()
            cal = sorted(float(Span.host[c0 + 2 * i + 1] - Span.host[c0 + 2 * i]) / k.wall_clock_khz() for i in range(ncal))
            Span.calib_ms = cal[ncal // 2]
            del g2, gc
        elif rank == 0 or world > 1:
            if rank == 0:
                install()
            try:
                # The events must time kernels, not the host: an eager step is ~1900 launches + 1400 event records, and wherever the
                # host falls behind (the 8x5 / 16x10 levels: 20 us kernels) the idle gap would land inside an event pair.  A spin kernel
                # holds the stream for ~40 ms first, so the whole step is enqueued before its first kernel starts.
                torch.cuda.synchronize()
                torch.cuda._sleep(int(0.040 * 2.1e9))
                fwd_bwd()
                torch.cuda.synchronize()
            finally:
                uninstall()
            trainer.allreduce_grads()
            opt_step()
            torch.cuda.synchronize()
        if rank == 0:
            if args.gemm_table:
                agg = {}
                for sp, f, key in recs:
                    e = agg.setdefault(key, [0, 0.0, 0.0])
                    e[0] += 1
                    e[1] += sp.ms()
                    e[2] += f
                rows = sorted(([list(kk) + [v[0], v[1], v[2] / (v[1] * 1e-3) / 1e12] for kk, v in agg.items()]), key=lambda r: -r[7])
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                with open(os.path.join(ROOT, "gpurun_out", "gemm_table.json"), "w") as f:
                    json.dump(rows, f)
                tn = trainer.rt.tuner
                if tn is not None:
                    with open(os.path.join(ROOT, "gpurun_out", "gemm_tuned.json"), "w") as f:
                        json.dump([[repr(kk), repr(tn.table.get(kk)), [[c if not isinstance(c, tuple) else list(c), (st[0] / st[1]) if st[1] else None]
                                                                      for c, st in zip(tn.cands[kk], tn.stats[kk])]] for kk in tn.cands], f)
            t_ms = sum(sp.ms() for sp, _, _ in recs)
            fl = sum(f for _, f, _ in recs)
            ach = fl / (t_ms * 1e-3) / 1e12
            # HBM-side bytes per launch of the same kernel family: rocprofv3 FETCH_SIZE / WRITE_SIZE passes over this exact command
            # (tools/pmc_traffic.py, gfx950 FETCH_SIZE x2 correction), committed under profiles/ -- bench.py cannot run a profiler
            traffic, traffic_source = None, None
            tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if (not args.tiny) and (T, h, w) == (14, 40, 64) and args.dtype == "fp16" and not args.lora_rank and os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    traffic = tj["gemm"]["bytes_per_launch"]
                    traffic_source = f"profiles/pmc_traffic.json@{tj.get('commit', 'unknown')} (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE passes " \
                                     "of this command, tools/pmc_traffic.py; NOT a counter of this run)"
                except Exception:  # noqa: BLE001
                    traffic = None
            tsa_op = None
            if regions:
                lv = {}
                for name, info, sp in regions:
                    M_, C_, T_ = info["M"], info["C"], info["T"]
                    fwd = name.endswith(".fwd")
                    # fwd: q/k/v 6MC^2 + out 2MC^2 + core (QK^T, PV) 4MTC;  bwd: data-grads 8MC^2 + weight-grads 8MC^2 + core (S again, dP, dV, dQ, dK) 10MTC
                    ofl = (8.0 * M_ * C_ * C_ + 4.0 * M_ * T_ * C_) if fwd else (16.0 * M_ * C_ * C_ + 10.0 * M_ * T_ * C_)
                    d = lv.setdefault((M_, C_), {"rows": M_, "channels": C_, "frames": T_, "fwd": [0, 0.0, 0.0], "bwd": [0, 0.0, 0.0]})
                    a = d["fwd" if fwd else "bwd"]
                    a[0] += 1
                    a[1] += sp.ms()
                    a[2] += ofl
                tot = {"fwd": [0.0, 0.0], "bwd": [0.0, 0.0]}
                levels = []
                for key in sorted(lv, reverse=True):
                    d = lv[key]
                    row = {"rows": d["rows"], "channels": d["channels"], "frames": d["frames"]}
                    for which in ("fwd", "bwd"):
                        n, ms_, ofl = d[which]
                        tot[which][0] += ms_
                        tot[which][1] += ofl
                        row[which] = {"blocks": n, "ms": ms_, "tflops": ofl / (ms_ * 1e-3) / 1e12 if ms_ else None,
                                  "frac_of_mfma_peak": ofl / (ms_ * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if ms_ else None}
                    levels.append(row)
                allms, allfl = tot["fwd"][0] + tot["bwd"][0], tot["fwd"][1] + tot["bwd"][1]
                tsa_op = {"what": "temporal self-attention op of every TemporalBasicTransformerBlock (norm1 -> q/k/v -> attention over frames -> "
                                  "out-projection + residual; backward: both data-grads, both weight-grads, the core, norm1): algorithmic FLOPs / "
                                  "summed duration of ALL launches between the region marks (the `clock` of this roofline object)",
                          "flops": "fwd 8MC^2 + 4MTC, bwd 16MC^2 + 10MTC", "levels": levels,
                          "fwd_frac_of_mfma_peak": tot["fwd"][1] / (tot["fwd"][0] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if tot["fwd"][0] else None,
                          "bwd_frac_of_mfma_peak": tot["bwd"][1] / (tot["bwd"][0] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if tot["bwd"][0] else None,
                          "op_frac_of_mfma_peak": allfl / (allms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if allms else None,
                          "ms_per_step": allms, "north_star_target": 0.40}
            roof = {"bound": "mfma", "kernel": "MFMA GEMM family (NT + implicit conv, TN weight-grad)", "achieved": ach, "peak": MFMA_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source, "launches": len(recs),
                    "flops_per_step": fl, "kernel_ms_per_step": t_ms,
                    "band_kernels": ({"what": "fused temporal self-attention launches (their projections are not in the family above)", "launches": len(band), "flops_per_step": sum(f for _, f in band),
                                      "kernel_ms_per_step": sum(sp.ms() for sp, _ in band)} if band else None),
                    "clock": ("svdx_stamp: device wall clock written by one-lane kernels around every launch of the family inside a second "
                              f"capture of the step (replay conditions); calibrated kernel boundary {Span.calib_ms * 1e3:.2f} us subtracted per launch"
                              if use_stamps else "HIP events around eager launches"),
                    "kernel_ms_per_step_events": events_ms,
                    "algorithmic_bytes_per_launch": sum(recs_bytes) / max(len(recs_bytes), 1),
                    "temporal_self_attention": tsa_op}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            cpu = cpu_baseline(dev, dt, c2=not args.no_cpu_c2 and not args.tiny)
        except Exception as e:  # noqa: BLE001
            cpu = {"error": repr(e)[:200]}

    if rank == 0:
        full = (not args.tiny) and (T, h, w) == (14, 40, 64) and not args.lora_rank
        line = {
            "metric": "train-step samples/sec (tiny debug topology)" if args.tiny else
                      "train-step samples/sec (14-frame 512x320 fp16 SVD UNet)" if not args.lora_rank else
                      f"train-step samples/sec (14-frame 512x320 {args.dtype} SVD UNet, LoRA r={args.lora_rank})",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic latents/CLIP embed, random-init weights (no checkpoints offline)",
            "config": {"workload": f"{'TINY debug topology (not the SVD UNet)' if args.tiny else 'SVD UNet'} train step, {T} frames {args.width}x{args.height}, batch 1/GPU"
                                   f"{' x %d micro-batches' % args.grad_accum if args.grad_accum > 1 else ''}, "
                                   f"{n_params} params ({n_train} trainable: "
                                   f"{'LoRA r=%d adapters on to_q/to_k/to_v/to_out.0' % args.lora_rank if args.lora_rank else 'temporal_transformer_block*'}), "
                                   "fwd + EDM loss + bwd + grad all-reduce + AdamW",
                       "global_batch": world * B * args.grad_accum, "grad_accum": args.grad_accum, "parallelism": f"dp{world}",
                       "grad_allreduce": ("none" if world == 1 else "one RCCL all-reduce after backward" if args.overlap == "single" else
                                          "one RCCL all-reduce after backward, under the next clip's VAE encode" if args.overlap == "vae" else
                                          "svdx_allreduce_grads (direct reduce-scatter + all-gather over peer-mapped buffers) after backward" if args.overlap == "direct" else
                                          "per-transformer-block RCCL all-reduces overlapped with the backward sweep"),
                       "schedules": schedules or None, "direct_allreduce_check": direct_check, "direct_allreduce_in_use": trainer.direct is not None,
                       "allreduce_ms_exposed": exposed_ms,
                       "ranks_seen": ranks_seen, "allreduce_ms": allreduce_ms, "allreduce_direct_ms": direct_ms, "rccl": rccl,
                       "allreduce_bytes": trainer.n_total * 4,
                       "exec": exec_mode, "gemm_tuning_sweeps": tune_sweeps, "gpu_clock": sampler.summary() if sampler is not None else None,
                       "gemm_variant": args.gemm_variant, "loss": loss, "loss_scale": state[1], "opt_steps": state[0],
                       "step_tflops_per_gpu": (STEP_TFLOP_C2 * args.grad_accum / (ms * 1e-3) if full else None),
                       "step_frac_of_mfma_peak": (STEP_TFLOP_C2 * args.grad_accum / (ms * 1e-3) / MFMA_PEAK_TFLOPS if full else None)},
            "roofline": roof, "cpu_baseline": cpu, "with_vae": vae_info, "real_loop": real,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
