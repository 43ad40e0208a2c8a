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


def lift(tree, lo, hi, filename):
    """Outermost statements lying entirely inside lines [lo, hi], compiled for exec."""
    import ast
    inside = [n for n in ast.walk(tree) if isinstance(n, ast.stmt) and n.lineno >= lo and n.end_lineno <= hi]
    top = sorted([n for n in inside if not any(m is not n and any(c is n for c in ast.walk(m)) for m in inside)], key=lambda n: n.lineno)
    return compile(ast.Module(body=top, type_ignores=[]), filename, "exec")


def reference_assembled_step(Ref, tree):
    """One whole optimizer step put together from the reference's own pieces, on the inputs of tests/golden/make_golden.py
    (SPEC there): the reference top-level class (over oracle blocks), its trainable-set loop and AdamW construction (:758-773), its
    nested `_get_add_time_ids` (:878-898), and the loop-body statements as written -- noising (:964-972; `rand_log_normal` is
    handed the batch's sigmas), concat (:992-1017, dropout off), the UNet call (:1020-1022), the loss (:1025-1036), backward (:1044),
    optimizer / scheduler / zero_grad (:1047-1049).  Compared against the oracle's step and stored."""
    import ast
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle.step import make_synthetic_batch
    from oracle.unet import TINY_CONFIG, UNetSpatioTemporalConditionOracle, scaled_init_
    B, T, h, w, seed, lr = 1, 4, 16, 16, 3, 1e-4                      # make_golden.py SPEC
    tsrc = "/root/reference/train_svd.py"
    orc = UNetSpatioTemporalConditionOracle(**TINY_CONFIG)
    scaled_init_(orc, seed)
    unet = Ref(**TINY_CONFIG)
    unet.load_state_dict(orc.state_dict(), strict=True)
    batch = make_synthetic_batch(B, T, h, w, seed + 1, cross_dim=TINY_CONFIG["cross_attention_dim"])
    ns = dict(torch=torch, unet=unet, optimizer_cls=torch.optim.AdamW, bsz=B, train_loss=0.0,
              args=SimpleNamespace(learning_rate=lr, adam_beta1=0.9, adam_beta2=0.999, adam_weight_decay=1e-2, adam_epsilon=1e-8,
                                   conditioning_dropout_prob=None, per_gpu_batch_size=B, gradient_accumulation_steps=1),
              accelerator=SimpleNamespace(device=torch.device("cpu"), backward=lambda loss: loss.backward(), gather=lambda t: t),
              lr_scheduler=SimpleNamespace(step=lambda: None), generator=None,
              rand_log_normal=lambda shape, loc, scale: batch["sigmas"].clone(),
              latents=batch["latents"], noise=batch["noise"], conditional_latents=batch["cond_latents"],
              encoder_hidden_states=batch["ehs"])
    exec(lift(tree, 758, 773, tsrc), ns)                                  # trainable set + optimizer
    nested = [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "_get_add_time_ids"]
    exec(compile(ast.Module(body=nested, type_ignores=[]), tsrc, "exec"), ns)
    ns["added_time_ids"] = ns["_get_add_time_ids"](7, 127, batch["cond_sigmas"][0], torch.float32, B)      # call site :981-987
    for lo, hi in ((964, 972), (992, 1017), (1020, 1022), (1025, 1036), (1039, 1041), (1044, 1044)):
        exec(lift(tree, lo, hi, tsrc), ns)
    grads = {n: p.grad.clone() for n, p in unet.named_parameters() if p.grad is not None}
    exec(lift(tree, 1047, 1049, tsrc), ns)
    after = {n: p.detach().clone() for n, p in unet.named_parameters() if p.requires_grad}
    assert all(p.grad is None or float(p.grad.abs().max()) == 0.0 for p in unet.parameters())           # zero_grad ran

    import e2e_checks
    o = e2e_checks.oracle_step(TINY_CONFIG, B, T, h, w, seed=seed, lr=lr, cross_dim=TINY_CONFIG["cross_attention_dim"])
    d_loss = abs(float(ns["loss"]) - o["loss"])
    d_pred = float((ns["model_pred"].detach() - o["pred"]).abs().max())
    d_grad = max(float((grads[n] - o["grads"][n]).abs().max()) for n in o["grads"])
    d_after = max(float((after[n] - o["params_after"][n]).abs().max()) for n in after)
    print(f"reference-assembled step vs oracle step: |d loss| {d_loss:.2e}, |d pred| {d_pred:.2e}, |d grad| {d_grad:.2e}, "
          f"|d params after AdamW| {d_after:.2e}; loss {float(ns['loss']):.7f}, train_loss {ns['train_loss']:.7f}")
    assert set(grads) == set(o["grads"]) and d_loss == 0.0 and d_pred == 0.0 and d_grad == 0.0 and d_after == 0.0
    names = sorted(grads)
    return {"step.loss": torch.tensor([float(ns["loss"])], dtype=torch.float64), "step.pred": ns["model_pred"].detach().contiguous(),
            "step.grad_norms": torch.tensor([float(grads[n].double().norm()) for n in names], dtype=torch.float64),
            "step.param_delta_norm": torch.tensor([float(sum((after[n] - orc.state_dict()[n]).double().pow(2).sum() for n in after)) ** 0.5],
                                                  dtype=torch.float64)}


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
    out.update(reference_assembled_step(Ref, tree))
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
