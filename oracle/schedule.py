"""CPU oracle: learning-rate schedules and EMA of the weights, as the reference's training loop uses them.

TEST INFRASTRUCTURE ONLY (see oracle/unet.py header).  The algorithms live in a dependency that is absent from
/root/reference: `diffusers.optimization.get_scheduler` (train_svd.py:51, 807-813, stepped at :1048) and
`diffusers.training_utils.EMAModel` (train_svd.py:48, 677-679, 1053-1054, 1101-1104, 1152-1154); diffusers is not installed in
this image.  The schedule multipliers are PINNED in tests/test_resume_ema.py against `transformers.optimization` (installed; the
module diffusers.optimization was derived from, same closed forms).  EMAModel is PARITY UNPINNED: restated from its published
algorithm (the reference does not pin diffusers; its import paths imply >= 0.26).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List

import torch

SCHEDULES = ("constant", "constant_with_warmup", "linear", "cosine", "cosine_with_restarts", "polynomial")


def lr_lambda(name: str, step: int, num_warmup_steps: int = 0, num_training_steps: int = 0, num_cycles=None, power: float = 1.0,
              lr_init: float = 1.0, lr_end: float = 1e-7) -> float:
    """Multiplier LambdaLR applies to the base lr after `step` scheduler steps (diffusers/optimization.py get_*_schedule*)."""
    w, t = num_warmup_steps, num_training_steps
    if name == "constant":
        return 1.0
    if name == "constant_with_warmup":
        return float(step) / float(max(1.0, w)) if step < w else 1.0
    if name == "linear":
        if step < w:
            return float(step) / float(max(1, w))
        return max(0.0, float(t - step) / float(max(1, t - w)))
    if name == "cosine":
        c = 0.5 if num_cycles is None else num_cycles
        if step < w:
            return float(step) / float(max(1, w))
        progress = float(step - w) / float(max(1, t - w))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(c) * 2.0 * progress)))
    if name == "cosine_with_restarts":
        c = 1 if num_cycles is None else num_cycles
        if step < w:
            return float(step) / float(max(1, w))
        progress = float(step - w) / float(max(1, t - w))
        if progress >= 1.0:
            return 0.0
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(c) * progress) % 1.0))))
    if name == "polynomial":
        if step < w:
            return float(step) / float(max(1, w))
        if step > t:
            return lr_end / lr_init
        lr_range = lr_init - lr_end
        pct_remaining = 1 - (step - w) / (t - w)
        return (lr_range * pct_remaining ** power + lr_end) / lr_init
    raise ValueError(f"unknown schedule {name!r}")


def lr_trajectory(name: str, base_lr: float, n_steps: int, num_processes: int = 1, skipped: Iterable[int] = (), **kw) -> List[float]:
    """lr seen by optimizer steps 0..n_steps-1 of the reference loop: the scheduler is built with warmup / total already
    multiplied by num_processes (train_svd.py:810-812) and accelerate's wrapper steps it num_processes times after every
    optimizer step that the GradScaler did not skip (`skipped`: indices of skipped steps)."""
    skipped = set(skipped)
    sched_step, out = 0, []
    for i in range(n_steps):
        out.append(base_lr * lr_lambda(name, sched_step, **kw))
        if i not in skipped:
            sched_step += num_processes
    return out


class EMAModel:
    """diffusers.training_utils.EMAModel on plain tensors (decay schedule, step, copy_to, store / restore, state dict)."""

    def __init__(self, parameters, decay: float = 0.9999, min_decay: float = 0.0, update_after_step: int = 0,
                 use_ema_warmup: bool = False, inv_gamma: float = 1.0, power: float = 2 / 3):
        self.shadow_params = [p.clone().detach() for p in parameters]
        self.temp_stored_params = None
        self.decay, self.min_decay, self.update_after_step = decay, min_decay, update_after_step
        self.use_ema_warmup, self.inv_gamma, self.power = use_ema_warmup, inv_gamma, power
        self.optimization_step = 0
        self.cur_decay_value = None

    def get_decay(self, optimization_step: int) -> float:
        step = max(0, optimization_step - self.update_after_step - 1)
        if step <= 0:
            return 0.0
        if self.use_ema_warmup:
            cur = 1 - (1 + step / self.inv_gamma) ** -self.power
        else:
            cur = (1 + step) / (10 + step)
        return max(min(cur, self.decay), self.min_decay)

    @torch.no_grad()
    def step(self, parameters) -> None:
        self.optimization_step += 1
        decay = self.get_decay(self.optimization_step)
        self.cur_decay_value = decay
        one_minus_decay = 1 - decay
        for s, p in zip(self.shadow_params, list(parameters)):
            if p.requires_grad:
                s.sub_(one_minus_decay * (s - p))
            else:
                s.copy_(p)

    def copy_to(self, parameters) -> None:
        for s, p in zip(self.shadow_params, list(parameters)):
            p.data.copy_(s.to(p.device).data)

    def store(self, parameters) -> None:
        self.temp_stored_params = [p.detach().cpu().clone() for p in parameters]

    def restore(self, parameters) -> None:
        if self.temp_stored_params is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
        for c, p in zip(self.temp_stored_params, list(parameters)):
            p.data.copy_(c.data)
        self.temp_stored_params = None

    def state_dict(self) -> Dict:
        return dict(decay=self.decay, min_decay=self.min_decay, optimization_step=self.optimization_step,
                    update_after_step=self.update_after_step, use_ema_warmup=self.use_ema_warmup, inv_gamma=self.inv_gamma,
                    power=self.power, shadow_params=self.shadow_params)
