#!/usr/bin/env python
"""train_svd.py on the MI355X-native step: the reference's training script (/root/reference/train_svd.py) with its loop body
(:931-1058) running on `svd_xtend_amd` -- VAE encode -> CLIP image embed -> EDM noising -> UNet step from a hipGraph -> EMA ->
`checkpoint-N` -> validation sampler.  Argument names and defaults are the reference's (:294-569); what the reference delegates to
accelerate (device placement, mixed precision, DDP, save_state / load_state) is `svd_xtend_amd.train.Trainer`.

    python examples/train_svd_amd.py --max_train_steps 10 --output_dir /tmp/svd_out                        # synthetic clips, random init
    python examples/train_svd_amd.py --pretrained_model_name_or_path <svd folder> --base_folder <frames>   # as the reference
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_svd_amd.py ...

Without `--pretrained_model_name_or_path` the three models are built at the SVD configuration with random weights (there is no
network here); without `--base_folder` the clips are synthetic U(-1, 1) pixels of the requested shape (the range of the
reference's DummyDataset, :125).  Out of scope, as in DESIGN.md section 7: trackers, hub upload, 8-bit Adam, xformers."""
from __future__ import annotations

import argparse
import math
import os
import random
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Stable Video Diffusion fine-tuning on MI355X (svd_xtend_amd)")
    ap.add_argument("--base_folder", default=None, help="folder of per-video frame folders (train_svd.py:299); default: synthetic clips")
    ap.add_argument("--pretrained_model_name_or_path", default=None)
    ap.add_argument("--num_frames", type=int, default=25)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--height", type=int, default=576)
    ap.add_argument("--num_validation_images", type=int, default=1)
    ap.add_argument("--validation_steps", type=int, default=500)
    ap.add_argument("--num_validation_steps", type=int, default=25, help="sampler steps of a validation video (the reference uses the pipeline's 25)")
    ap.add_argument("--validation_image", default=None, help="conditioning image of the validation videos (the reference reads demo.jpg)")
    ap.add_argument("--output_dir", default="./outputs")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--per_gpu_batch_size", type=int, default=1)
    ap.add_argument("--num_train_epochs", type=int, default=100)
    ap.add_argument("--max_train_steps", type=int, default=None)
    ap.add_argument("--gradient_accumulation_steps", type=int, default=1)
    ap.add_argument("--gradient_checkpointing", action="store_true", help="accepted; activations fit HBM (DESIGN.md section 2)")
    ap.add_argument("--learning_rate", type=float, default=1e-4)
    ap.add_argument("--scale_lr", action="store_true")
    ap.add_argument("--lr_scheduler", default="constant")
    ap.add_argument("--lr_warmup_steps", type=int, default=500)
    ap.add_argument("--conditioning_dropout_prob", type=float, default=0.1)
    ap.add_argument("--use_ema", action="store_true")
    ap.add_argument("--num_workers", type=int, default=8)
    ap.add_argument("--adam_beta1", type=float, default=0.9)
    ap.add_argument("--adam_beta2", type=float, default=0.999)
    ap.add_argument("--adam_weight_decay", type=float, default=1e-2)
    ap.add_argument("--adam_epsilon", type=float, default=1e-8)
    ap.add_argument("--mixed_precision", default="fp16", choices=["fp16", "bf16"])
    ap.add_argument("--checkpointing_steps", type=int, default=500)
    ap.add_argument("--checkpoints_total_limit", type=int, default=2)
    ap.add_argument("--resume_from_checkpoint", default=None)
    ap.add_argument("--pretrain_unet", default=None)
    ap.add_argument("--num_samples", type=int, default=100000, help="length of the dataset (train_svd.py:70)")
    ap.add_argument("--tiny", action="store_true", help="smoke configuration: tiny UNet / VAE / CLIP topologies instead of SVD's")
    ap.add_argument("--no_graph", action="store_true", help="eager launches instead of the captured step")
    ap.add_argument("--reference_rng", action="store_true",
                    help="draw cond_sigmas / sigmas the reference's way (host, process-global generator, train_svd.py:954 / :964): with --seed the run "
                         "walks the reference's sigma sequence")
    return ap.parse_args(argv)


class SyntheticClips(torch.utils.data.Dataset):
    """Stand-in for the reference's DummyDataset (train_svd.py:69-137) when no `--base_folder` is given: seeded U(-1, 1) pixels,
    [num_frames, 3, height, width]."""

    def __init__(self, num_samples, width, height, sample_frames, seed):
        self.n, self.shape, self.seed = num_samples, (sample_frames, 3, height, width), seed

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        return {"pixel_values": torch.rand(self.shape, generator=g) * 2 - 1}


class FrameFolders(torch.utils.data.Dataset):
    """The reference's DummyDataset (train_svd.py:69-137): a random run of `sample_frames` consecutive frames of a random folder,
    resized, scaled to [-1, 1]."""

    def __init__(self, base_folder, num_samples, width, height, sample_frames):
        self.base, self.folders = base_folder, sorted(os.listdir(base_folder))
        self.n, self.w, self.h, self.f = num_samples, width, height, sample_frames

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        from PIL import Image
        folder = os.path.join(self.base, random.choice(self.folders))
        frames = sorted(os.listdir(folder))
        if len(frames) < self.f:
            raise ValueError(f"The selected folder '{folder}' contains fewer than `{self.f}` frames.")
        s = random.randint(0, len(frames) - self.f)
        out = torch.empty(self.f, 3, self.h, self.w)
        for i, name in enumerate(frames[s:s + self.f]):
            with Image.open(os.path.join(folder, name)) as img:
                t = torch.from_numpy(np.array(img.convert("RGB").resize((self.w, self.h)))).float()
            out[i] = (t / 127.5 - 1).permute(2, 0, 1)
        return {"pixel_values": out}


TINY = dict(unet=dict(block_out_channels=(64, 128, 128, 128), addition_time_embed_dim=32, projection_class_embeddings_input_dim=96,
                      cross_attention_dim=64, num_attention_heads=(1, 2, 2, 2)),
            vae=dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1),
            clip=dict(hidden_size=320, intermediate_size=640, projection_dim=64, num_hidden_layers=2, num_attention_heads=4,
                      image_size=56, patch_size=14))


def build_models(args, dev, dtype):
    from svd_xtend_amd.clip import CLIPVisionModelWithProjection
    from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
    from svd_xtend_amd.vae import AutoencoderKLTemporalDecoder
    path = args.pretrained_model_name_or_path
    if path:                                                                     # train_svd.py:646-656
        image_encoder = CLIPVisionModelWithProjection.from_pretrained(path, subfolder="image_encoder", variant="fp16")
        vae = AutoencoderKLTemporalDecoder.from_pretrained(path, subfolder="vae", variant="fp16")
        unet = UNetSpatioTemporalConditionModel.from_pretrained(args.pretrain_unet or path, subfolder="unet", variant="fp16")
    else:
        import bench
        cfg = TINY if args.tiny else dict(unet={}, vae={}, clip={})
        torch.manual_seed(args.seed or 0)
        with torch.device(dev):
            image_encoder = CLIPVisionModelWithProjection(**cfg["clip"])
            vae = AutoencoderKLTemporalDecoder(**cfg["vae"])
            unet = UNetSpatioTemporalConditionModel(**cfg["unet"])
        for i, m in enumerate((unet, vae, image_encoder)):
            bench.init_weights_(m, seed=1234 + i)
    for m in (vae, image_encoder):                                               # :659-660
        m.requires_grad_(False)
        m.to(dev)
        m.prepare(dtype)
    return unet.to(dev), vae, image_encoder


def main(argv=None):
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    is_main = rank == 0
    if args.seed is not None:                                                    # set_seed, train_svd.py:620-621
        random.seed(args.seed)
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
    if is_main:
        os.makedirs(args.output_dir, exist_ok=True)
    dtype = torch.float16 if args.mixed_precision == "fp16" else torch.bfloat16

    from svd_xtend_amd import checkpoint
    from svd_xtend_amd.loop import TrainLoop
    from svd_xtend_amd.optimization import get_scheduler
    from svd_xtend_amd.pipeline import StableVideoDiffusionPipeline
    from svd_xtend_amd.train import Trainer
    from svd_xtend_amd.training_utils import EMAModel
    from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel

    unet, vae, image_encoder = build_models(args, dev, dtype)
    if args.gradient_checkpointing:
        unet.enable_gradient_checkpointing()
    if args.scale_lr:                                                            # :738-742
        args.learning_rate = args.learning_rate * args.gradient_accumulation_steps * args.per_gpu_batch_size * world
    if args.per_gpu_batch_size != 1:
        raise NotImplementedError("per_gpu_batch_size: every BASELINE configuration trains one clip per GPU (the reference's own "
                                  "noise_aug_strength is batch-1 only, train_svd.py:955)")
    # trainable set (:758-766), AdamW (:767-773), mixed precision + DDP (:815): the Trainer
    trainer = Trainer(unet, dtype=dtype, lr=args.learning_rate, betas=(args.adam_beta1, args.adam_beta2),
                      weight_decay=args.adam_weight_decay, eps=args.adam_epsilon, grad_accum=args.gradient_accumulation_steps)
    ema_unet = None
    if args.use_ema:                                                             # :677-679
        ema_unet = EMAModel(unet.parameters(), model_cls=UNetSpatioTemporalConditionModel, model_config=unet.config,
                            on_weights_changed=trainer.weights_changed)

    dataset = (FrameFolders(args.base_folder, args.num_samples, args.width, args.height, args.num_frames) if args.base_folder
               else SyntheticClips(args.num_samples, args.width, args.height, args.num_frames, (args.seed or 0) + rank))
    sampler_gen = torch.Generator().manual_seed((args.seed or 0) + rank)
    sampler = torch.utils.data.RandomSampler(dataset, generator=sampler_gen)
    loader = torch.utils.data.DataLoader(dataset, sampler=sampler, batch_size=args.per_gpu_batch_size,
                                         num_workers=args.num_workers if args.base_folder else 0, pin_memory=True)   # :780-786
    num_update_steps_per_epoch = math.ceil(len(loader) / args.gradient_accumulation_steps)                           # :789-793
    if args.max_train_steps is None:
        args.max_train_steps = args.num_train_epochs * num_update_steps_per_epoch
    lr_scheduler = get_scheduler(args.lr_scheduler, optimizer=trainer, num_warmup_steps=args.lr_warmup_steps * world,
                                 num_training_steps=args.max_train_steps * world)                                    # :807-813

    global_step = 0
    if args.resume_from_checkpoint:                                              # :900-924
        path = checkpoint.latest_checkpoint(args.output_dir, args.resume_from_checkpoint)
        if path is None:
            print(f"Checkpoint '{args.resume_from_checkpoint}' does not exist. Starting a new training run.")
        else:
            print(f"Resuming from checkpoint {path}")
            trainer.load_state(os.path.join(args.output_dir, path), ema=ema_unet, scheduler=lr_scheduler)
            global_step = checkpoint.global_step_of(path)

    # A resumed run must not replay the clip order and the sigma / noise / dropout draws of steps 1..N (the reference skips the consumed
    # batches and restores the generators that drive its noise, train_svd.py:900-931): both streams are re-seeded from (seed, rank, global_step)
    sampler_gen.manual_seed((args.seed or 0) + rank + 1000003 * global_step)
    loop = TrainLoop(trainer, vae, image_encoder, conditioning_dropout_prob=args.conditioning_dropout_prob,
                     seed=(args.seed or 0) * 7919 + rank + 104729 * global_step, use_graph=not args.no_graph, ema=ema_unet, reference_rng=args.reference_rng)

    def clips():
        while True:
            it = iter(loader)
            while True:
                group = []
                try:
                    for _ in range(args.gradient_accumulation_steps):
                        group.append(next(it)["pixel_values"])
                except StopIteration:
                    break
                yield group

    def validate(step):                                                          # :1093-1154
        if args.use_ema:
            ema_unet.store(unet.parameters())
            ema_unet.copy_to(unet.parameters())
        pipeline = StableVideoDiffusionPipeline.from_pretrained(args.pretrained_model_name_or_path, unet=unet, image_encoder=image_encoder,
                                                                vae=vae, torch_dtype=dtype).to(dev)
        pipeline.set_progress_bar_config(disable=True)
        val_dir = os.path.join(args.output_dir, "validation_images")
        os.makedirs(val_dir, exist_ok=True)
        if args.validation_image:
            from PIL import Image
            image = Image.open(args.validation_image).convert("RGB").resize((args.width, args.height))
        else:
            image = torch.rand(1, 3, args.height, args.width, generator=torch.Generator().manual_seed(1))
        for i in range(args.num_validation_images):
            frames = pipeline(image, height=args.height, width=args.width, num_frames=args.num_frames, decode_chunk_size=8,
                              motion_bucket_id=127, fps=7, noise_aug_strength=0.02, num_inference_steps=args.num_validation_steps).frames[0]
            out = os.path.join(val_dir, f"step_{step}_val_img_{i}.gif")           # export_to_gif of the reference (:1143)
            frames[0].save(out, save_all=True, append_images=list(frames[1:]), duration=125, loop=0)
        if args.use_ema:
            ema_unet.restore(unet.parameters())
        del pipeline
        torch.cuda.empty_cache()

    if is_main:
        n_tr = sum(p.numel() for p in unet.parameters() if p.requires_grad)
        print(f"***** Running training *****\n  Num examples = {len(dataset)}\n  Instantaneous batch size per device = {args.per_gpu_batch_size}\n"
              f"  Total train batch size = {args.per_gpu_batch_size * world * args.gradient_accumulation_steps}\n"
              f"  Gradient Accumulation steps = {args.gradient_accumulation_steps}\n  Total optimization steps = {args.max_train_steps}\n"
              f"  Trainable parameters = {n_tr:,}", flush=True)
    feed = clips()
    train_loss = float("nan")
    if global_step < args.max_train_steps:                # (the capture's warm-up pass is a real optimizer step: none is due when the run is already complete)
        loop.start(next(feed))                            # conditioners of the first clip + capture of the step
    t0, seen = time.perf_counter(), 0
    while global_step < args.max_train_steps:
        nxt = next(feed) if global_step + 1 < args.max_train_steps else None
        train_loss = loop.step(nxt)                       # mean loss over ranks and micro-batches (the all-reduced loss slot)
        lr_scheduler.step()                               # :1048 (the schedule advances on the device)
        global_step += 1
        seen += 1
        if is_main:
            print(f"step {global_step} train_loss {train_loss:.6f} lr {lr_scheduler.get_last_lr()[0]:.3e} "
                  f"{(time.perf_counter() - t0) / seen * 1e3:.1f} ms/step", flush=True)
            if global_step % args.checkpointing_steps == 0:                      # :1059-1090
                checkpoint.rotate_checkpoints(args.output_dir, args.checkpoints_total_limit)
                save_path = os.path.join(args.output_dir, f"checkpoint-{global_step}")
                trainer.save_state(save_path, ema=ema_unet, scheduler=lr_scheduler)
                print(f"Saved state to {save_path}", flush=True)
            if global_step % args.validation_steps == 0 or global_step == 1:     # :1092-1096
                validate(global_step)
    if world > 1:
        dist.barrier()                                    # accelerator.wait_for_everyone(), :1167
    if is_main:                                           # :1171-1187: the final pipeline folder
        if args.use_ema:
            ema_unet.copy_to(unet.parameters())
        unet.save_pretrained(os.path.join(args.output_dir, "unet"))
    if world > 1:
        dist.destroy_process_group()
    return dict(global_step=global_step, train_loss=train_loss)


if __name__ == "__main__":
    main()
