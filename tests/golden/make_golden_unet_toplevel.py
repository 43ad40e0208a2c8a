"""Golden output of the reference's OWN top-level UNet class, executed in this container over the oracle's blocks.

/root/reference/src/unet_spatio_temporal_condition.py defines `UNetSpatioTemporalConditionModel` (constructor topology :71-246, forward
orchestration :357-490) but takes every block from diffusers, which is not installed.  This script executes that file unmodified
with stand-in `diffusers.*` modules whose block factories (`get_down_block`, `get_up_block`, `UNetMidBlockSpatioTemporal`,
`Timesteps`, `TimestepEmbedding`) return the ORACLE's restated blocks (oracle/unet.py).  What runs is therefore the reference's own
constructor and forward; what is pinned is everything the top level decides: per-block channel plumbing, the time / added-id
embedding sum, frame flattening, `repeat_interleave`, the skip-connection stack, the output head, and the state-dict key layout
(the oracle's weights are loaded with strict=True).  The blocks themselves stay parity-unpinned (oracle/unet.py header).
Nothing of the reference's source is written to the repo.  Usage (this container only): python tests/golden/make_golden_unet_toplevel.py
"""
import inspect
import logging as pylogging
import os
import sys
import types
from types import SimpleNamespace

import torch
import torch.nn as nn
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference/src/unet_spatio_temporal_condition.py"
CASES = [(2, 3, 16, 16, 11), (1, 4, 16, 24, 12)]            # (batch, frames, h, w, seed)


def toplevel_inputs(B, T, h, w, seed, cross_dim):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, T, 8, h, w, generator=g), torch.randn(B, generator=g), torch.randn(B, 1, cross_dim, generator=g),
            torch.tensor([[7.0, 127.0, 0.02]]).repeat(B, 1))


def stand_in_diffusers():
    from oracle import unet as O

    def register_to_config(init):
        def wrapped(self, *a, **kw):
            bound = inspect.signature(init).bind(self, *a, **kw)
            bound.apply_defaults()
            self.config = SimpleNamespace(**{k: v for k, v in bound.arguments.items() if k != "self"})
            init(self, *a, **kw)
        return wrapped

    def get_down_block(kind, num_layers, transformer_layers_per_block, in_channels, out_channels, temb_channels, add_downsample,
                       resnet_eps, cross_attention_dim, num_attention_heads, resnet_act_fn):
        if kind == "CrossAttnDownBlockSpatioTemporal":
            return O.CrossAttnDownBlockSpatioTemporal(in_channels, out_channels, temb_channels, num_layers, transformer_layers_per_block,
                                                      num_attention_heads, cross_attention_dim, add_downsample)
        assert kind == "DownBlockSpatioTemporal", kind
        return O.DownBlockSpatioTemporal(in_channels, out_channels, temb_channels, num_layers, add_downsample)

    def get_up_block(kind, num_layers, transformer_layers_per_block, in_channels, out_channels, prev_output_channel, temb_channels,
                     add_upsample, resnet_eps, resolution_idx, cross_attention_dim, num_attention_heads, resnet_act_fn):
        if kind == "CrossAttnUpBlockSpatioTemporal":
            return O.CrossAttnUpBlockSpatioTemporal(in_channels, out_channels, prev_output_channel, temb_channels, num_layers,
                                                    transformer_layers_per_block, num_attention_heads, cross_attention_dim, add_upsample)
        assert kind == "UpBlockSpatioTemporal", kind
        return O.UpBlockSpatioTemporal(in_channels, prev_output_channel, out_channels, temb_channels, num_layers, add_upsample)

    mods = {
        "diffusers": {},
        "diffusers.configuration_utils": dict(ConfigMixin=type("ConfigMixin", (), {}), register_to_config=register_to_config),
        "diffusers.loaders": dict(UNet2DConditionLoadersMixin=type("UNet2DConditionLoadersMixin", (), {}),
                                  PeftAdapterMixin=type("PeftAdapterMixin", (), {})),
        "diffusers.utils": dict(BaseOutput=type("BaseOutput", (), {}), logging=SimpleNamespace(get_logger=pylogging.getLogger)),
        "diffusers.models": {},
        "diffusers.models.attention_processor": dict(CROSS_ATTENTION_PROCESSORS=(), AttentionProcessor=object, AttnProcessor=object),
        "diffusers.models.embeddings": dict(TimestepEmbedding=O.TimestepEmbedding, Timesteps=O.Timesteps),
        "diffusers.models.modeling_utils": dict(ModelMixin=nn.Module),
        "diffusers.models.unets": {},
        "diffusers.models.unets.unet_3d_blocks": dict(UNetMidBlockSpatioTemporal=O.UNetMidBlockSpatioTemporal,
                                                      get_down_block=get_down_block, get_up_block=get_up_block),
    }
    out = {}
    for name, attrs in mods.items():
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        out[name] = m
    return out


def reference_class():
    saved = {k: sys.modules.get(k) for k in stand_in_diffusers()}
    sys.modules.update(stand_in_diffusers())
    try:
        mod = types.ModuleType("reference_unet_toplevel")
        sys.modules[mod.__name__] = mod                     # dataclasses looks the defining module up
        exec(compile(open(REF).read(), REF, "exec"), mod.__dict__)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod.UNetSpatioTemporalConditionModel


def main():
    from oracle.unet import TINY_CONFIG, UNetSpatioTemporalConditionOracle, scaled_init_
    Ref = reference_class()
    orc = UNetSpatioTemporalConditionOracle(**TINY_CONFIG)
    scaled_init_(orc, 0)
    ref = Ref(**TINY_CONFIG)
    ref.load_state_dict(orc.state_dict(), strict=True)          # same key layout, or this raises
    assert [n for n, _ in ref.named_parameters()] == [n for n, _ in orc.named_parameters()]
    out = {}
    with torch.no_grad():
        for i, (B, T, h, w, seed) in enumerate(CASES):
            x, t, ehs, ids = toplevel_inputs(B, T, h, w, seed, TINY_CONFIG["cross_attention_dim"])
            y = ref(x, t, ehs, added_time_ids=ids).sample
            yo = orc(x, t, ehs, added_time_ids=ids).sample
            print(f"case {i}: reference top level vs oracle top level max |diff| = {float((y - yo).abs().max()):.3e}")
            out[f"case{i}.sample"] = y.contiguous()
    # the trainable-set selection + optimizer construction of train_svd.py:758-773, executed as written on the reference class instance
    import ast
    tsrc = "/root/reference/train_svd.py"
    tree = ast.parse(open(tsrc).read())
    stmts = [n for n in ast.walk(tree) if isinstance(n, ast.stmt) and 758 <= n.lineno and n.end_lineno <= 773]
    top = sorted([n for n in stmts if not any(m is not n and any(c is n for c in ast.walk(m)) for m in stmts)], key=lambda n: n.lineno)
    args = SimpleNamespace(learning_rate=1e-4, adam_beta1=0.9, adam_beta2=0.999, adam_weight_decay=1e-2, adam_epsilon=1e-8)
    ns = dict(unet=ref, torch=torch, args=args, optimizer_cls=torch.optim.AdamW)
    exec(compile(ast.Module(body=top, type_ignores=[]), tsrc, "exec"), ns)
    chosen = [n for n, p in ref.named_parameters() if p.requires_grad]
    assert len(ns["parameters_list"]) == len(chosen) and isinstance(ns["optimizer"], torch.optim.AdamW)
    with open(os.path.join(HERE, "unet_toplevel_trainable_names.txt"), "w") as f:
        f.write("\n".join(chosen) + "\n")
    print("reference selection loop on the tiny model:", len(chosen), "trainable tensors")
    # the full-size constructor: the reference's own channel plumbing yields the published parameter count
    with torch.device("meta"):
        full = Ref()
    n_full = sum(p.numel() for p in full.parameters())
    n_train = sum(p.numel() for n, p in full.named_parameters() if "temporal_transformer_block" in n)
    print("full-size reference constructor:", n_full, "parameters,", n_train, "trainable by name")
    out["full_counts"] = torch.tensor([n_full, n_train])
    path = os.path.join(HERE, "unet_toplevel.safetensors")
    save_file(out, path)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
