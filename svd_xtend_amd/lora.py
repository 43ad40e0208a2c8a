"""LoRA adapters for the attention projections -- reference config 5 (train_svd_lora.py:659-674).

The reference builds `LoraConfig(r=rank, lora_alpha=rank, init_lora_weights="gaussian", target_modules=["to_k","to_q","to_v",
"to_out.0"])` and calls `unet.add_adapter(cfg)`; diffusers hands that to peft, which wraps every targeted nn.Linear so that
`y = base(x) + lora_B(lora_A(x)) * (lora_alpha / r)`.  peft is neither vendored in the reference nor installed here; its module
layout is reproduced (`base_layer`, `lora_A["default"]`, `lora_B["default"]`) so that parameter names -- and therefore
`named_parameters()` filters, optimizer groups and adapter checkpoints -- read exactly as with peft.  The holder has no forward:
the UNet's explicit fwd/bwd (unet.py) runs the adapter branch through ops.LoraOp / ops.SmallLoraOp."""
from dataclasses import dataclass
from typing import Dict, Sequence

import torch
import torch.nn as nn


@dataclass
class LoraConfig:
    """The fields of peft.LoraConfig that train_svd_lora.py:659-664 sets (any object with these attributes is accepted)."""
    r: int = 8
    lora_alpha: float = 8
    init_lora_weights: object = True            # True -> peft default (kaiming-uniform A), "gaussian" -> N(0, (1/r)^2); B = 0
    target_modules: Sequence[str] = ("to_k", "to_q", "to_v", "to_out.0")


class LoraLinear(nn.Module):
    """Parameter holder with peft's layout for one wrapped nn.Linear."""

    def __init__(self, base: nn.Linear, r: int, lora_alpha: float, init="gaussian"):
        super().__init__()
        self.base_layer = base
        self.r, self.lora_alpha, self.scaling = int(r), float(lora_alpha), float(lora_alpha) / int(r)
        dev = base.weight.device
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False, device=dev)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False, device=dev)})
        if init == "gaussian":
            nn.init.normal_(self.lora_A["default"].weight, std=1.0 / r)
        else:
            nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=5 ** 0.5)
        nn.init.zeros_(self.lora_B["default"].weight)
        base.weight.requires_grad_(False)
        if base.bias is not None:
            base.bias.requires_grad_(False)

    @property
    def A(self) -> nn.Parameter:
        return self.lora_A["default"].weight        # [r, in]

    @property
    def B(self) -> nn.Parameter:
        return self.lora_B["default"].weight        # [out, r]

    @property
    def weight(self):
        return self.base_layer.weight

    @property
    def bias(self):
        return self.base_layer.bias

    @property
    def in_features(self):
        return self.base_layer.in_features

    @property
    def out_features(self):
        return self.base_layer.out_features

    def forward(self, *a, **k):
        raise RuntimeError("LoraLinear is a parameter holder; the UNet runs it through its explicit fwd/bwd")


def base_linear(lin: nn.Module) -> nn.Linear:
    return lin.base_layer if isinstance(lin, LoraLinear) else lin


def inject(model: nn.Module, cfg) -> int:
    """Wrap every nn.Linear whose dotted name ends with one of cfg.target_modules (peft's suffix match)."""
    targets = tuple(cfg.target_modules)
    init = cfg.init_lora_weights
    n = 0
    for name, mod in list(model.named_modules()):
        if isinstance(mod, nn.Linear) and name.endswith(targets) and ".base_layer" not in name and ".lora_" not in name:
            parent_name, _, leaf = name.rpartition(".")
            parent = model.get_submodule(parent_name) if parent_name else model
            wrapped = LoraLinear(mod, cfg.r, cfg.lora_alpha, init)
            if leaf.isdigit():
                parent[int(leaf)] = wrapped
            else:
                setattr(parent, leaf, wrapped)
            n += 1
    return n


def lora_state_dict(model: nn.Module, prefix: str = "unet.") -> Dict[str, torch.Tensor]:
    """Adapter weights under the names the reference's `save_lora_weights` writes (train_svd_lora.py:1065-1074, 1148-1153):
    peft's state dict drops the adapter name (`...to_q.lora_A.weight`) and the pipeline prefixes `unet.`."""
    out = {}
    for name, p in model.named_parameters():
        if ".lora_A.default." in name or ".lora_B.default." in name:
            out[prefix + name.replace(".default.", ".")] = p.detach()
    return out


def load_lora_state_dict(model: nn.Module, sd: Dict[str, torch.Tensor], prefix: str = "unet.") -> None:
    own = dict(model.named_parameters())
    for k, v in sd.items():
        name = k[len(prefix):] if k.startswith(prefix) else k
        name = name.replace(".lora_A.weight", ".lora_A.default.weight").replace(".lora_B.weight", ".lora_B.default.weight")
        if name not in own:
            raise KeyError(f"unexpected adapter key {k}")
        with torch.no_grad():
            own[name].copy_(v.to(own[name].dtype))


LORA_WEIGHT_NAME_SAFE = "pytorch_lora_weights.safetensors"     # diffusers' file name for `save_lora_weights(safe_serialization=True)`


def save_lora_weights(save_directory: str, unet_lora_layers: Dict[str, torch.Tensor], safe_serialization: bool = True,
                      weight_name: str = None) -> str:
    """`StableVideoDiffusionPipeline.save_lora_weights` as the reference calls it (train_svd_lora.py:1069-1073, 1148-1152):
    one safetensors file of the `unet.`-prefixed adapter tensors (`lora_state_dict(unet)`).  Returns the path written."""
    import os

    from safetensors.torch import save_file
    if not safe_serialization:
        raise NotImplementedError("save_lora_weights: only safetensors output is supported")
    os.makedirs(save_directory, exist_ok=True)
    path = os.path.join(save_directory, weight_name or LORA_WEIGHT_NAME_SAFE)
    save_file({k: v.detach().cpu().contiguous() for k, v in unet_lora_layers.items()}, path)
    return path


def load_lora_weights(model: nn.Module, path: str, weight_name: str = None) -> None:
    """Read a `pytorch_lora_weights.safetensors` (file or the folder holding it) into an adapter-injected UNet."""
    import os

    from safetensors.torch import load_file
    if os.path.isdir(path):
        path = os.path.join(path, weight_name or LORA_WEIGHT_NAME_SAFE)
    load_lora_state_dict(model, load_file(path))
