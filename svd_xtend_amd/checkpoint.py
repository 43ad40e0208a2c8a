"""`checkpoint-N` folders of the reference's training loop: what `accelerator.save_state` / `load_state` write and read with the
hooks of /root/reference/train_svd.py:698-729 installed, the "latest checkpoint" rule of :900-924 and the rotation of :1062-1082.

Layout (accelerate's file names, so that a run can be resumed across the two implementations):
    unet/config.json + diffusion_pytorch_model.safetensors     save_model_hook (:703)
    unet_ema/...                                                save_model_hook with --use_ema (:699-700)
    optimizer.bin        torch.save of torch.optim.AdamW.state_dict(): per-parameter step / exp_avg / exp_avg_sq, in the order of the
                         optimizer's parameter list (= named_parameters order of the trainables, :761-773)
    scaler.pt            GradScaler.state_dict() (fp16 runs)
    scheduler.bin        LambdaLR.state_dict()
    random_states_<rank>.pkl   python / numpy / torch (/ device) RNG states
The optimizer state lives in flat float buffers here (train.Trainer); this module converts to and from the per-parameter form.
"""
from __future__ import annotations

import os
import random
import shutil
from typing import List, Optional

import numpy as np
import torch

OPTIMIZER_NAME, SCHEDULER_NAME, SCALER_NAME, RNG_STATE_NAME = "optimizer.bin", "scheduler.bin", "scaler.pt", "random_states"


# ---- directory rules ---------------------------------------------------------------------------------------------------------
def _checkpoints(output_dir: str) -> List[str]:
    dirs = [d for d in os.listdir(output_dir) if d.startswith("checkpoint")]
    return sorted(dirs, key=lambda x: int(x.split("-")[1]))


def latest_checkpoint(output_dir: str, resume_from_checkpoint: str = "latest") -> Optional[str]:
    """train_svd.py:900-909: the folder NAME to resume from (`checkpoint-<global_step>`), or None when there is none."""
    if resume_from_checkpoint != "latest":
        return os.path.basename(resume_from_checkpoint)
    dirs = _checkpoints(output_dir) if os.path.isdir(output_dir) else []
    return dirs[-1] if dirs else None


def global_step_of(name: str) -> int:
    """train_svd.py:919."""
    return int(name.split("-")[1])


def resume_position(global_step: int, num_update_steps_per_epoch: int, gradient_accumulation_steps: int):
    """train_svd.py:921-925: (first_epoch, resume_step) -- the epoch to restart in and how many of its micro-batches to skip."""
    resume_global_step = global_step * gradient_accumulation_steps
    return (global_step // num_update_steps_per_epoch,
            resume_global_step % (num_update_steps_per_epoch * gradient_accumulation_steps))


def rotate_checkpoints(output_dir: str, total_limit: Optional[int]) -> List[str]:
    """train_svd.py:1062-1082, run BEFORE saving a new checkpoint: keep at most `total_limit - 1` of the existing ones (oldest
    removed first).  Returns the removed folder names."""
    if total_limit is None:
        return []
    have = _checkpoints(output_dir)
    if len(have) < total_limit:
        return []
    gone = have[:len(have) - total_limit + 1]
    for d in gone:
        shutil.rmtree(os.path.join(output_dir, d))
    return gone


# ---- optimizer state <-> torch.optim.AdamW.state_dict() -------------------------------------------------------------------------
def optimizer_state_dict(trainer) -> dict:
    st = trainer.opt_state.detach().cpu()
    step = float(st[0])
    state = {}
    for i, (p, off) in enumerate(zip(trainer.params, trainer.offsets)):
        n = p.numel()
        state[i] = {"step": torch.tensor(step), "exp_avg": trainer.m_flat[off:off + n].view(p.shape).detach().cpu().clone(),
                    "exp_avg_sq": trainer.v_flat[off:off + n].view(p.shape).detach().cpu().clone()}
    from .optimization import lr_lambda
    sched = trainer.schedule
    lr_now = trainer.lr * lr_lambda(int(step) * max(1, int(sched["steps_per_step"])), base_lr=trainer.lr, **sched)
    group = {"lr": lr_now, "betas": tuple(trainer.betas), "eps": trainer.eps, "weight_decay": trainer.wd, "amsgrad": False,
             "foreach": None, "maximize": False, "capturable": False, "differentiable": False, "fused": None,
             "initial_lr": trainer.lr, "params": list(range(len(trainer.params)))}
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(trainer, sd: dict) -> None:
    groups = sd["param_groups"]
    order = [i for g in groups for i in g["params"]]
    if len(order) != len(trainer.params):
        raise ValueError(f"optimizer state holds {len(order)} parameters, the trainer has {len(trainer.params)} trainables")
    g0 = groups[0]
    theirs = [float(g0.get("initial_lr", g0["lr"])), float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]),
              float(g0["weight_decay"])]
    mine = [trainer.lr, trainer.betas[0], trainer.betas[1], trainer.eps, trainer.wd]
    if any(abs(a - b) > 1e-12 * max(1.0, abs(a)) for a, b in zip(mine, theirs)):
        raise ValueError(f"optimizer hyper-parameters differ: checkpoint (lr, beta1, beta2, eps, wd) = {theirs}, trainer = {mine}")
    steps = set()
    trainer.m_flat.zero_()
    trainer.v_flat.zero_()
    for i, p, off in zip(order, trainer.params, trainer.offsets):
        s = sd["state"].get(i)
        if s is None:                       # a parameter torch never stepped (no gradient): zero moments
            continue
        if tuple(s["exp_avg"].shape) != tuple(p.shape):
            raise ValueError(f"optimizer state {i}: shape {tuple(s['exp_avg'].shape)} != parameter {tuple(p.shape)}")
        n = p.numel()
        trainer.m_flat[off:off + n].copy_(s["exp_avg"].reshape(-1).to(trainer.dev, torch.float32))
        trainer.v_flat[off:off + n].copy_(s["exp_avg_sq"].reshape(-1).to(trainer.dev, torch.float32))
        steps.add(float(s["step"]))
    if len(steps) > 1:
        raise ValueError(f"optimizer state has parameters at different steps {sorted(steps)}: one shared step counter is kept here")
    step = steps.pop() if steps else 0.0
    st = trainer.opt_state.detach().cpu()
    st[0] = step
    st[5] = 1.0 - trainer.betas[0] ** step if step > 0 else 1.0
    st[6] = 1.0 - trainer.betas[1] ** step if step > 0 else 1.0
    st[3] = st[7] = 0.0
    trainer.opt_state.copy_(st)


def scaler_state_dict(trainer) -> dict:
    st = trainer.opt_state.detach().cpu()
    return {"scale": float(st[1]), "growth_factor": 2.0, "backoff_factor": 0.5, "growth_interval": trainer.growth_interval,
            "_growth_tracker": int(st[2])}


def load_scaler_state_dict(trainer, sd: dict) -> None:
    st = trainer.opt_state.detach().cpu()
    st[1], st[2] = float(sd["scale"]), float(sd["_growth_tracker"])
    st[4] = 1.0 / float(sd["scale"])
    trainer.growth_interval = int(sd.get("growth_interval", trainer.growth_interval))
    trainer.opt_state.copy_(st)


# ---- save_state / load_state ---------------------------------------------------------------------------------------------------
def _rank(trainer) -> int:
    import torch.distributed as dist
    return dist.get_rank(trainer.pg) if (dist.is_available() and dist.is_initialized()) else 0


def save_state(trainer, output_dir: str, ema=None, scheduler=None) -> None:
    """Every rank writes its RNG states; rank 0 writes the rest (weights and optimizer state are replicated)."""
    os.makedirs(output_dir, exist_ok=True)
    rank = _rank(trainer)
    states = {"random_state": random.getstate(), "numpy_random_seed": np.random.get_state(), "torch_manual_seed": torch.get_rng_state()}
    if trainer.dev.type == "cuda":
        states["torch_cuda_manual_seed"] = torch.cuda.get_rng_state_all()
    torch.save(states, os.path.join(output_dir, f"{RNG_STATE_NAME}_{rank}.pkl"))
    if rank != 0:
        return
    if ema is not None:
        ema.save_pretrained(os.path.join(output_dir, "unet_ema"))
    trainer.model.save_pretrained(os.path.join(output_dir, "unet"))
    torch.save(optimizer_state_dict(trainer), os.path.join(output_dir, OPTIMIZER_NAME))
    if trainer.dynamic:
        torch.save(scaler_state_dict(trainer), os.path.join(output_dir, SCALER_NAME))
    if scheduler is not None:
        torch.save(scheduler.state_dict(), os.path.join(output_dir, SCHEDULER_NAME))


def load_state(trainer, input_dir: str, ema=None, scheduler=None) -> None:
    """Weights (strict, diffusers key names), EMA, optimizer moments + step, loss scale, RNG states; the packed 16-bit weights and
    every table derived from them are rebuilt.  Capture `GraphedStep` AFTER loading."""
    from safetensors.torch import load_file

    from .unet import WEIGHTS_NAME
    if not os.path.isdir(input_dir):
        raise FileNotFoundError(f"checkpoint folder {input_dir} does not exist")
    model = trainer.model
    if ema is not None:                                                             # load_model_hook, :709-714
        from .training_utils import _EMA_KEYS
        _, kw = type(model).load_config(os.path.join(input_dir, "unet_ema"), return_unused_kwargs=True)
        names = [n for n, _ in model.named_parameters()]
        sd = load_file(os.path.join(input_dir, "unet_ema", WEIGHTS_NAME.format(variant="")))
        ema.load_state_dict({**{k: v for k, v in kw.items() if k in _EMA_KEYS}, "shadow_params": [sd[n] for n in names]})
    cfg = type(model).load_config(os.path.join(input_dir, "unet"))                  # :720-725
    model.register_to_config(**cfg)
    sd = load_file(os.path.join(input_dir, "unet", WEIGHTS_NAME.format(variant="")))
    model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    load_optimizer_state_dict(trainer, torch.load(os.path.join(input_dir, OPTIMIZER_NAME), map_location="cpu", weights_only=False))
    spath = os.path.join(input_dir, SCALER_NAME)
    if trainer.dynamic and os.path.exists(spath):
        load_scaler_state_dict(trainer, torch.load(spath, map_location="cpu", weights_only=False))
    if scheduler is not None and os.path.exists(os.path.join(input_dir, SCHEDULER_NAME)):
        scheduler.load_state_dict(torch.load(os.path.join(input_dir, SCHEDULER_NAME), map_location="cpu", weights_only=False))
    rpath = os.path.join(input_dir, f"{RNG_STATE_NAME}_{_rank(trainer)}.pkl")
    if os.path.exists(rpath):
        states = torch.load(rpath, map_location="cpu", weights_only=False)
        random.setstate(states["random_state"])
        np.random.set_state(states["numpy_random_seed"])
        torch.set_rng_state(states["torch_manual_seed"])
        if trainer.dev.type == "cuda" and "torch_cuda_manual_seed" in states:
            torch.cuda.set_rng_state_all(states["torch_cuda_manual_seed"])
    trainer._build_runtime()
    trainer.micro = 0
