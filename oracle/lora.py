"""TEST INFRASTRUCTURE (CPU oracle) -- PARITY UNPINNED (peft is absent), like the diffusers blocks of oracle/unet.py.

LoRA as the reference uses it (config 5): /root/reference/train_svd_lora.py:659-674 builds
`LoraConfig(r=rank, lora_alpha=rank, init_lora_weights="gaussian", target_modules=["to_k","to_q","to_v","to_out.0"])` and calls
`unet.add_adapter(cfg)` (diffusers -> peft `inject_adapter_in_model`).  peft is not installed here and not vendored in the
reference, so its published behaviour is restated: every targeted nn.Linear is wrapped in a module that keeps the frozen layer as
`base_layer` and adds `lora_A["default"]` (Linear in->r, no bias, N(0, (1/r)^2) for "gaussian") and `lora_B["default"]`
(Linear r->out, no bias, zeros); forward = base(x) + lora_B(lora_A(x)) * (lora_alpha / r).  Parameter names therefore read
`...to_q.base_layer.weight`, `...to_q.lora_A.default.weight`, `...to_q.lora_B.default.weight`.
Self-pin: r = 64 adds 26,558,464 parameters to the SVD UNet (tests/test_oracle.py)."""
from typing import Sequence

import torch.nn as nn

TARGETS = ("to_k", "to_q", "to_v", "to_out.0")


class LoraLinear(nn.Module):
    def __init__(self, base: nn.Linear, r: int, lora_alpha: float, init: str = "gaussian"):
        super().__init__()
        self.base_layer = base
        self.r, self.lora_alpha, self.scaling = r, lora_alpha, lora_alpha / r
        dev, dt = base.weight.device, base.weight.dtype
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False, device=dev, dtype=dt)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False, device=dev, dtype=dt)})
        if dev.type != "meta":
            if init == "gaussian":
                nn.init.normal_(self.lora_A["default"].weight, std=1.0 / r)
            else:                                   # peft default: kaiming-uniform A
                nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=5 ** 0.5)
            nn.init.zeros_(self.lora_B["default"].weight)
        base.weight.requires_grad_(False)
        if base.bias is not None:
            base.bias.requires_grad_(False)

    @property
    def in_features(self):
        return self.base_layer.in_features

    @property
    def out_features(self):
        return self.base_layer.out_features

    @property
    def weight(self):
        return self.base_layer.weight

    @property
    def bias(self):
        return self.base_layer.bias

    def forward(self, x):
        return self.base_layer(x) + self.lora_B["default"](self.lora_A["default"](x)) * self.scaling


def add_adapter(model: nn.Module, r: int, lora_alpha: float = None, target_modules: Sequence[str] = TARGETS,
                init_lora_weights: str = "gaussian") -> int:
    """Wrap every nn.Linear whose dotted name ends with one of `target_modules`; returns the number of wrapped layers."""
    lora_alpha = r if lora_alpha is None else lora_alpha
    n = 0
    for name, mod in list(model.named_modules()):
        if isinstance(mod, nn.Linear) and name.endswith(tuple(target_modules)) and ".base_layer" not in name:
            parent_name, _, leaf = name.rpartition(".")
            parent = model.get_submodule(parent_name) if parent_name else model
            wrapped = LoraLinear(mod, r, lora_alpha, init_lora_weights)
            if leaf.isdigit():
                parent[int(leaf)] = wrapped
            else:
                setattr(parent, leaf, wrapped)
            n += 1
    return n
