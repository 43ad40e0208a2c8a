"""What the product's one deliberate numeric deviation from reference config 5 costs or buys (missing item 5).

/root/reference/train_svd_lora.py:669-674: under `--mixed_precision bf16` the UNet is cast to bf16 BEFORE `add_adapter`, and only fp16
runs upcast the adapters (`cast_training_params`): the reference trains bf16 LoRA parameters with torch.optim.AdamW state in bf16.  The
product keeps float32 masters and float32 Adam moments for the adapters (the kernels read a bf16 twin that AdamW re-rounds every step).
This script measures the two against an all-float32 run on the CPU oracle (tiny topology, rank-8 adapters, one fixed batch):
  A  float32 adapters, float32 AdamW                                   (the yardstick)
  B  float32 masters + float32 AdamW, forward through bf16-rounded adapters   (the product's semantics)
  C  bf16 adapters + bf16 AdamW state, forward through them                    (the reference's semantics)
Everything else (base weights, activations) is float32 in all three, so the difference is the master / optimizer-state dtype alone.

    python tools/lora_dtype_deviation.py [--steps 30] [--lr 1e-4]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(mode: str, steps: int, lr: float, seed: int = 4):
    from oracle.lora import add_adapter
    from oracle.step import edm_inputs, edm_loss, make_synthetic_batch
    from oracle.unet import TINY_CONFIG, UNetSpatioTemporalConditionOracle, scaled_init_
    cfg = TINY_CONFIG
    orc = UNetSpatioTemporalConditionOracle(**cfg)
    scaled_init_(orc, seed)
    for p in orc.parameters():
        p.requires_grad_(False)
    torch.manual_seed(seed + 5)
    add_adapter(orc, 8, 8)
    gen = torch.Generator().manual_seed(seed + 17)
    for n, p in orc.named_parameters():
        if ".lora_B." in n:
            p.data.copy_(torch.randn(p.shape, generator=gen) * 0.05)
    lora = [(n, p) for n, p in orc.named_parameters() if ".lora_" in n]
    batch = make_synthetic_batch(1, 3, 16, 16, seed + 1, cross_dim=cfg["cross_attention_dim"])
    unet_in, ts, ehs, ids, noisy, sig = edm_inputs(batch)
    if mode == "C":
        masters = [torch.nn.Parameter(p.detach().to(torch.bfloat16)) for _, p in lora]        # the parameters ARE bf16
    else:
        masters = [torch.nn.Parameter(p.detach().clone()) for _, p in lora]
    opt = torch.optim.AdamW(masters, lr=lr, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    losses = []
    for _ in range(steps):
        with torch.no_grad():
            for (_, p), m in zip(lora, masters):
                p.copy_(m.detach().float() if mode != "B" else m.detach().to(torch.bfloat16).float())     # what the forward multiplies with
        for _, p in lora:
            p.grad = None
            p.requires_grad_(True)
        loss = edm_loss(orc(unet_in, ts, ehs, added_time_ids=ids).sample, noisy, batch["latents"], sig)
        loss.backward()
        for (_, p), m in zip(lora, masters):
            m.grad = p.grad.to(m.dtype)
        opt.step()
        losses.append(float(loss.detach()))
    return losses, torch.cat([m.detach().float().reshape(-1) for m in masters])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--lr", type=float, default=1e-4)
    args = ap.parse_args()
    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    out = {}
    ref_l, ref_p = run("A", args.steps, args.lr)
    p0 = run("A", 0, args.lr)[1]
    moved = float((ref_p - p0).norm())
    for mode, what in (("B", "product: float32 masters + float32 AdamW, bf16 twin in the forward"), ("C", "reference: bf16 adapters + bf16 AdamW state")):
        l, p = run(mode, args.steps, args.lr)
        out[mode] = {"what": what, "update_error_rel": float((p - ref_p).norm()) / moved,
                     "update_cosine": float(((p - p0) @ (ref_p - p0)) / ((p - p0).norm() * (ref_p - p0).norm() + 1e-30)),
                     "loss_last": l[-1], "loss_last_ref": ref_l[-1], "loss_rel_err_last": abs(l[-1] - ref_l[-1]) / abs(ref_l[-1])}
    out["steps"], out["lr"], out["float32_update_norm"] = args.steps, args.lr, moved
    print(json.dumps(out, indent=1))
    return out


if __name__ == "__main__":
    main()
