"""Learning-rate schedules of the reference's training loop (`from diffusers.optimization import get_scheduler`,
/root/reference/train_svd.py:51, built at :807-813, stepped at :1048, read at :1160).

The schedule is not a host-side LambdaLR: `get_scheduler` writes its parameters into the Trainer's device state and
`svdx_optim_prep` evaluates lambda(step) there from the optimizer's own step counter (include/svdx.h, opt_state[8..15]).  A step
replayed from hipGraphs (train.GraphedStep) therefore follows the schedule with no host work, and a step the loss scaler skipped
does not advance it -- accelerate's wrapper behaves the same way (it does not step the scheduler after a skipped optimizer step).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

from . import kernels as K

_NEEDS_WARMUP = {"constant_with_warmup", "linear", "cosine", "cosine_with_restarts", "polynomial"}
_NEEDS_TOTAL = {"linear", "cosine", "cosine_with_restarts", "polynomial"}
_DEFAULT_CYCLES = {"cosine": 0.5, "cosine_with_restarts": 1}


class LRSchedule:
    """What the reference holds as `lr_scheduler`: `step()`, `get_last_lr()`, `state_dict()` / `load_state_dict()`."""

    def __init__(self, trainer, **params):
        self.trainer = trainer
        trainer.set_schedule(**params)

    def step(self) -> None:
        """The device advances the schedule together with the optimizer step (train_svd.py:1047-1048 become one call)."""
        return None

    @property
    def last_epoch(self) -> int:
        """Scheduler steps taken so far = optimizer steps that were not skipped x steps_per_step (host sync)."""
        return int(float(self.trainer.opt_state[0])) * max(1, int(self.trainer.schedule["steps_per_step"]))

    def get_last_lr(self) -> List[float]:
        """lr of the NEXT optimizer step, as LambdaLR reports it after scheduler.step() (train_svd.py:1160).  Host sync."""
        return [self.trainer.lr * lr_lambda(self.last_epoch, base_lr=self.trainer.lr, **self.trainer.schedule)]

    def state_dict(self) -> Dict:
        n = self.last_epoch
        lr = self.get_last_lr()
        return {"base_lrs": [self.trainer.lr], "last_epoch": n, "verbose": False, "_step_count": n + 1,
                "_get_lr_called_within_step": False, "_last_lr": lr, "lr_lambdas": [None]}

    def load_state_dict(self, sd: Dict) -> None:
        """The position of the schedule is the optimizer's step counter (restored with the optimizer state); a LambdaLR state
        that disagrees with it cannot be honoured and is reported."""
        if "last_epoch" in sd and int(sd["last_epoch"]) != self.last_epoch:
            raise ValueError(f"scheduler state is at step {sd['last_epoch']}, the optimizer state implies {self.last_epoch}: "
                             "load the optimizer state first, and use the same num_processes as the run that saved it")
        if "base_lrs" in sd and abs(float(sd["base_lrs"][0]) - self.trainer.lr) > 1e-12 * max(1.0, self.trainer.lr):
            raise ValueError(f"scheduler state has base lr {sd['base_lrs'][0]}, the trainer {self.trainer.lr}")


def parse_step_rules(step_rules: str):
    """diffusers.optimization.get_piecewise_constant_schedule's rule string "m0:s0,m1:s1,...,m_last" -> (boundaries sorted ascending,
    their multipliers, m_last).  Same parsing: `value:steps` pairs, the last entry a bare multiplier."""
    rule_list = step_rules.split(",")
    rules = {}
    for rule_str in rule_list[:-1]:
        value_str, steps_str = rule_str.split(":")
        rules[int(steps_str)] = float(value_str)
    last = float(rule_list[-1])
    bounds = sorted(rules)
    return [float(b) for b in bounds], [rules[b] for b in bounds], last


def lr_lambda(step: int, name: str, num_warmup_steps: int = 0, num_training_steps: int = 0, num_cycles: float = 0.0,
              power: float = 1.0, lr_end: float = 1e-7, base_lr: float = 1.0, steps_per_step: int = 1, step_rules: Optional[str] = None) -> float:
    """Host evaluation of the multiplier the device applies (same arithmetic as csrc/optim.hip lr_lambda, in double); used for
    reporting only -- the step itself never reads it."""
    kind = K.SCHED_KINDS[name]
    warm, total, cycles, pw, end_ratio = float(num_warmup_steps), float(num_training_steps), float(num_cycles), float(power), lr_end / base_lr
    n = float(step)
    if kind == 0:
        return 1.0
    if kind == 6:
        bounds, mults, last = parse_step_rules(step_rules)
        for b, m in zip(bounds, mults):
            if n < b:
                return m
        return last
    if kind == 5:
        if n < warm:
            return n / max(1.0, warm)
        if n > total:
            return end_ratio
        return (1.0 - end_ratio) * (1.0 - (n - warm) / (total - warm)) ** pw + end_ratio
    if n < warm:
        return n / max(1.0, warm)
    if kind == 1:
        return 1.0
    if kind == 2:
        return max(0.0, (total - n) / max(1.0, total - warm))
    progress = (n - warm) / max(1.0, total - warm)
    if kind == 3:
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * cycles * 2.0 * progress)))
    if progress >= 1.0:
        return 0.0
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((cycles * progress) % 1.0))))


def get_scheduler(name, optimizer=None, step_rules: Optional[str] = None, num_warmup_steps: Optional[int] = None,
                  num_training_steps: Optional[int] = None, num_cycles: Optional[float] = None, power: float = 1.0,
                  last_epoch: int = -1, steps_per_step: Optional[int] = None) -> LRSchedule:
    """diffusers.optimization.get_scheduler with the Trainer in the `optimizer` seat (train_svd.py:807-813).

    The reference multiplies warmup / total steps by `accelerator.num_processes` because accelerate steps the scheduler that
    many times per optimizer step; pass the same products here -- `steps_per_step` defaults to the trainer's world size, which
    reproduces that stepping.  Same argument errors as diffusers; `piecewise_constant` (diffusers' step rules "m0:s0,...,m_last")
    is evaluated on the device from up to 8 rules stored behind the 16 state floats."""
    name = getattr(name, "value", name)                 # diffusers accepts its SchedulerType enum as well
    trainer = optimizer
    if trainer is None or not hasattr(trainer, "set_schedule"):
        raise TypeError("get_scheduler: pass the svd_xtend_amd Trainer as `optimizer`")
    if last_epoch != -1:
        raise NotImplementedError("get_scheduler: resume by loading the optimizer state (the schedule follows its step counter)")
    if name == "piecewise_constant":
        if not step_rules:
            raise ValueError("piecewise_constant requires `step_rules`, e.g. \"1:10,0.1:20,0.01:30,0.005\"")
        bounds, _, _ = parse_step_rules(step_rules)
        if len(bounds) > K.SCHED_MAX_RULES:
            raise ValueError(f"piecewise_constant: at most {K.SCHED_MAX_RULES} step rules have a device form (got {len(bounds)})")
        return LRSchedule(trainer, name=name, step_rules=step_rules,
                          steps_per_step=trainer.world if steps_per_step is None else steps_per_step)
    if name not in K.SCHED_KINDS:
        raise ValueError(f"{name} is not a valid SchedulerType")
    if name in _NEEDS_WARMUP and num_warmup_steps is None:
        raise ValueError(f"{name} requires `num_warmup_steps`, please provide that argument.")
    if name in _NEEDS_TOTAL and num_training_steps is None:
        raise ValueError(f"{name} requires `num_training_steps`, please provide that argument.")
    lr_end = 1e-7
    if name == "polynomial" and not trainer.lr > lr_end:
        raise ValueError(f"lr_end ({lr_end}) must be be smaller than initial lr ({trainer.lr})")
    cycles = _DEFAULT_CYCLES.get(name, 0.0) if num_cycles is None else num_cycles
    return LRSchedule(trainer, name=name, num_warmup_steps=num_warmup_steps or 0, num_training_steps=num_training_steps or 0,
                      num_cycles=cycles, power=power, lr_end=lr_end,
                      steps_per_step=trainer.world if steps_per_step is None else steps_per_step)
