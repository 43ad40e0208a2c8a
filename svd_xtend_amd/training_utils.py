"""EMA of the UNet weights (`from diffusers.training_utils import EMAModel`, /root/reference/train_svd.py:52; built at :677-679,
stepped after every optimizer step at :1053-1054, swapped in for validation at :1101-1104 / :1152-1154, saved / restored with the
checkpoints at :699-700, :709-714, copied into the final weights at :1170-1171).

Same surface and decay rule as diffusers' EMAModel; different storage.  Only trainable parameters can drift from their average,
so only they get a shadow: one flat float buffer laid out like the Trainer's master buffer, updated by ONE `svdx_ema_lerp` launch
per step (3.2 GB of HBM traffic for the 398 M trainables instead of a per-tensor loop over 1.5 G parameters).  A frozen
parameter's average is the parameter itself (diffusers copies it into its shadow on every step, to the same effect).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, List, Optional

import torch

from . import kernels as K

_EMA_KEYS = ("decay", "min_decay", "optimization_step", "update_after_step", "use_ema_warmup", "inv_gamma", "power")


class EMAModel:
    def __init__(self, parameters: Iterable[torch.nn.Parameter], decay: float = 0.9999, min_decay: float = 0.0,
                 update_after_step: int = 0, use_ema_warmup: bool = False, inv_gamma: float = 1.0, power: float = 2 / 3,
                 model_cls=None, model_config: Optional[Dict] = None, on_weights_changed=None, **unused):
        self._params: List[torch.nn.Parameter] = list(parameters)
        self.decay, self.min_decay, self.update_after_step = decay, min_decay, update_after_step
        self.use_ema_warmup, self.inv_gamma, self.power = use_ema_warmup, inv_gamma, power
        self.optimization_step = 0
        self.cur_decay_value = None
        self.model_cls, self.model_config = model_cls, model_config
        self.temp_stored_params = None
        # a prepared model reads 16-bit copies of its weights: pass `trainer.weights_changed` here and copy_to / restore call it
        self.on_weights_changed = on_weights_changed
        # built on first use: the reference constructs the EMA before it decides which parameters train (train_svd.py:677 vs
        # :735-766).  Until the first step() with a decay > 0 the average equals the weights, so nothing is lost by waiting.
        self._spans = None          # [(shadow_flat, param_flat_view, n)]
        self._shadow: Dict[int, torch.Tensor] = {}      # id(param) -> view of its shadow

    # ---- decay rule (EMAModel.get_decay) -------------------------------------------------------------------------
    def get_decay(self, optimization_step: int) -> float:
        step = max(0, optimization_step - self.update_after_step - 1)
        if step <= 0:
            return 0.0
        if self.use_ema_warmup:
            cur = 1 - (1 + step / self.inv_gamma) ** -self.power
        else:
            cur = (1 + step) / (10 + step)
        return max(min(cur, self.decay), self.min_decay)

    # ---- storage ----------------------------------------------------------------------------------------------------
    def _plan(self, parameters: List[torch.nn.Parameter]) -> None:
        """Group the trainable parameters by storage; parameters that tile one buffer (the Trainer's flat master buffer, gaps =
        alignment padding) become a single span."""
        train = [p for p in parameters if p.requires_grad]
        by_storage: Dict[int, List[torch.nn.Parameter]] = {}
        for p in train:
            if not p.data.is_contiguous() or p.dtype != torch.float32:
                raise ValueError("EMAModel: trainable parameters must be contiguous float32 (the float masters)")
            by_storage.setdefault(p.data.untyped_storage().data_ptr(), []).append(p)
        spans, shadow = [], {}
        for ps in by_storage.values():
            ps = sorted(ps, key=lambda q: q.data.storage_offset())
            lo = ps[0].data.storage_offset()
            hi = max(q.data.storage_offset() + q.numel() for q in ps)
            groups = [ps] if (hi - lo) <= 1.25 * sum(q.numel() for q in ps) + 64 * len(ps) else [[q] for q in ps]
            for grp in groups:
                lo = grp[0].data.storage_offset()
                lo4 = lo // 4 * 4                                   # 16-byte aligned start for the vector kernel
                hi = max(q.data.storage_offset() + q.numel() for q in grp)
                base = grp[0].data
                src = torch.empty(0, dtype=torch.float32, device=base.device).set_(base.untyped_storage(), lo4, (hi - lo4,))
                old = [self._shadow.get(id(q)) for q in grp]
                sh = src.clone()
                for q, o in zip(grp, old):
                    view = sh[q.data.storage_offset() - lo4:q.data.storage_offset() - lo4 + q.numel()].view(q.shape)
                    if o is not None:
                        view.copy_(o)                                # keep the average across a re-plan
                    shadow[id(q)] = view
                spans.append((sh, src, hi - lo4))
        self._spans, self._shadow = spans, shadow
        self._sig = self._signature(parameters)

    @staticmethod
    def _signature(parameters):
        return tuple((id(p), p.requires_grad, p.data.data_ptr()) for p in parameters)

    def _ensure(self, parameters) -> List[torch.nn.Parameter]:
        parameters = list(parameters)
        if self._spans is None or self._sig != self._signature(parameters):
            self._plan(parameters)
        return parameters

    @property
    def shadow_params(self) -> List[torch.Tensor]:
        ps = self._ensure(self._params)
        return [self._shadow.get(id(p), p.data) for p in ps]

    # ---- EMAModel surface ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, parameters) -> None:
        parameters = self._ensure(parameters)
        self.optimization_step += 1
        decay = self.get_decay(self.optimization_step)
        self.cur_decay_value = decay
        k = K.backend()
        for sh, src, n in self._spans:
            k.ema_lerp(sh, src, n, 1.0 - decay)

    @torch.no_grad()
    def copy_to(self, parameters) -> None:
        """Write the averages into the live weights (then `on_weights_changed`, see __init__)."""
        self._ensure(parameters)
        for sh, src, _ in self._spans:
            src.copy_(sh)
        if self.on_weights_changed is not None:
            self.on_weights_changed()

    def to(self, device=None, dtype=None) -> None:       # train_svd.py:713, 820 -- the shadow already lives beside the weights
        return None

    @torch.no_grad()
    def store(self, parameters) -> None:
        """Keep a copy of the current trainable weights (device side; diffusers parks all 1.5 G parameters on the host)."""
        self._ensure(parameters)
        self.temp_stored_params = [src.clone() for _, src, _ in self._spans]

    @torch.no_grad()
    def restore(self, parameters) -> None:
        if self.temp_stored_params is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
        self._ensure(parameters)
        for (_, src, _), c in zip(self._spans, self.temp_stored_params):
            src.copy_(c)
        self.temp_stored_params = None
        if self.on_weights_changed is not None:
            self.on_weights_changed()

    def state_dict(self) -> Dict:
        sd = {k: getattr(self, k) for k in _EMA_KEYS}
        sd["shadow_params"] = self.shadow_params
        return sd

    def load_state_dict(self, state_dict: Dict) -> None:
        sd = dict(state_dict)
        for k, lo, hi, typ in (("decay", 0.0, 1.0, float), ("min_decay", None, None, float), ("optimization_step", None, None, int),
                               ("update_after_step", None, None, int), ("use_ema_warmup", None, None, bool),
                               ("inv_gamma", None, None, (float, int)), ("power", None, None, (float, int))):
            v = sd.get(k, getattr(self, k))
            if not isinstance(v, typ) or (typ is int and isinstance(v, bool)):
                raise ValueError(f"Invalid {k}")
            if lo is not None and not (lo <= v <= hi):
                raise ValueError("Decay must be between 0 and 1")
            setattr(self, k, v)
        shadow = sd.get("shadow_params", None)
        if shadow is not None:
            ps = self._ensure(self._params)
            shadow = list(shadow)
            if len(shadow) != len(ps) or not all(isinstance(t, torch.Tensor) for t in shadow):
                raise ValueError("shadow_params must all be Tensors, one per parameter")
            with torch.no_grad():
                for p, t in zip(ps, shadow):
                    dst = self._shadow.get(id(p))
                    if dst is not None:
                        dst.copy_(t.to(dst.device, torch.float32))

    # ---- on-disk form: a diffusers model folder whose config.json also carries the EMA settings ---------------------------
    def _named(self):
        """Parameter names, from the model class on the meta device (no weights are allocated)."""
        if self.model_cls is None or self.model_config is None:
            raise ValueError("`save_pretrained` can only be used if `model_cls` and `model_config` were defined at __init__.")
        with torch.device("meta"):
            skeleton = self.model_cls.from_config(self.model_config)
        names = [n for n, _ in skeleton.named_parameters()]
        if len(names) != len(self._params):
            raise ValueError(f"EMAModel: {len(self._params)} parameters but {self.model_cls.__name__}(config) has {len(names)}")
        return names

    def save_pretrained(self, path: str) -> None:
        from safetensors.torch import save_file

        from .unet import CONFIG_NAME, WEIGHTS_NAME
        names = self._named()
        os.makedirs(path, exist_ok=True)
        cfg = {"_class_name": self.model_cls.__name__, "_svd_xtend_amd": True}
        cfg.update({k: (list(v) if isinstance(v, tuple) else v) for k, v in dict(self.model_config).items()})
        cfg.update({k: getattr(self, k) for k in _EMA_KEYS})
        with open(os.path.join(path, CONFIG_NAME), "w") as f:
            json.dump(cfg, f, indent=2)
        save_file({n: t.detach().cpu().contiguous() for n, t in zip(names, self.shadow_params)},
                  os.path.join(path, WEIGHTS_NAME.format(variant="")))

    @classmethod
    def from_pretrained(cls, path: str, model_cls) -> "EMAModel":
        _, ema_kwargs = model_cls.load_config(path, return_unused_kwargs=True)
        model = model_cls.from_pretrained(path)
        for p in model.parameters():
            p.requires_grad_(True)          # a stand-alone EMA holds every weight as data (diffusers clones them all)
        ema = cls(model.parameters(), model_cls=model_cls, model_config=model.config)
        ema._keepalive = model
        ema.load_state_dict({k: v for k, v in ema_kwargs.items() if k in _EMA_KEYS})
        return ema
