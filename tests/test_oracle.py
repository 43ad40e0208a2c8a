"""Self-pinning checks of the CPU oracle: the reference holds no golden vectors, so the
restatement is pinned structurally -- parameter inventory, trainable set, LoRA delta, key layout."""
import pytest
import torch

from oracle.step import edm_inputs, edm_loss, make_synthetic_batch, rand_log_normal
from oracle.unet import SVD_CONFIG, TINY_CONFIG, UNetSpatioTemporalConditionOracle, scaled_init_


@pytest.fixture(scope="module")
def meta_model():
    with torch.device("meta"):
        return UNetSpatioTemporalConditionOracle(**SVD_CONFIG)


def test_param_count_matches_published_svd_unet(meta_model):
    assert sum(p.numel() for p in meta_model.parameters()) == 1_524_623_082


def test_trainable_set_train_svd_761(meta_model):
    n = sum(p.numel() for k, p in meta_model.named_parameters() if "temporal_transformer_block" in k)
    assert n == 397_620_480


def test_lora_rank64_delta(meta_model):
    r, tot = 64, 0
    for name, mod in meta_model.named_modules():
        if isinstance(mod, torch.nn.Linear) and name.endswith(("to_q", "to_k", "to_v", "to_out.0")):
            tot += r * (mod.in_features + mod.out_features)
    assert tot == 26_558_464


def test_state_dict_key_layout(meta_model):
    keys = set(meta_model.state_dict().keys())
    for k in ["conv_in.weight", "time_embedding.linear_1.weight", "add_embedding.linear_2.bias",
              "down_blocks.0.resnets.0.spatial_res_block.norm1.weight",
              "down_blocks.0.resnets.1.temporal_res_block.conv1.weight",
              "down_blocks.0.resnets.0.time_mixer.mix_factor",
              "down_blocks.1.resnets.0.spatial_res_block.conv_shortcut.weight",
              "down_blocks.0.attentions.0.time_pos_embed.linear_1.weight",
              "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_out.0.bias",
              "down_blocks.0.attentions.0.temporal_transformer_blocks.0.ff_in.net.0.proj.weight",
              "down_blocks.0.attentions.0.temporal_transformer_blocks.0.ff.net.2.bias",
              "down_blocks.2.downsamplers.0.conv.weight", "mid_block.attentions.0.proj_out.weight",
              "mid_block.resnets.1.temporal_res_block.time_emb_proj.bias", "up_blocks.0.upsamplers.0.conv.weight",
              "up_blocks.3.attentions.2.time_mixer.mix_factor", "conv_norm_out.weight", "conv_out.bias"]:
        assert k in keys, k
    assert "down_blocks.3.downsamplers.0.conv.weight" not in keys and "up_blocks.3.upsamplers.0.conv.weight" not in keys
    sd = meta_model.state_dict()
    assert sd["down_blocks.0.resnets.0.temporal_res_block.conv1.weight"].shape == (320, 320, 3, 1, 1)
    assert sd["up_blocks.1.resnets.2.spatial_res_block.conv1.weight"].shape == (1280, 1920, 3, 3)
    assert sd["down_blocks.0.attentions.0.temporal_transformer_blocks.0.attn2.to_k.weight"].shape == (320, 1024)
    assert sum(1 for k in keys if k.endswith("mix_factor")) == 38


def test_tiny_forward_shape_and_loss_finite():
    torch.manual_seed(0)
    m = UNetSpatioTemporalConditionOracle(**TINY_CONFIG)
    scaled_init_(m, 0)
    b = make_synthetic_batch(2, 3, 16, 24, 1, cross_dim=64)
    unet_in, ts, ehs, ids, noisy, sig = edm_inputs(b)
    assert unet_in.shape == (2, 3, 8, 16, 24) and ts.shape == (2,) and ids.shape == (2, 3)
    out = m(unet_in, ts, ehs, added_time_ids=ids).sample
    assert out.shape == (2, 3, 4, 16, 24)
    assert torch.isfinite(edm_loss(out, noisy, b["latents"], sig))


def test_config1_as_written_is_infeasible():
    """BASELINE.json config 1 (256x160 -> latent 20x32): 20 -> 10 -> 5 -> 3, upsample gives 6 != 5."""
    m = UNetSpatioTemporalConditionOracle(**TINY_CONFIG)
    b = make_synthetic_batch(1, 2, 20, 32, 1, cross_dim=64)
    unet_in, ts, ehs, ids, _, _ = edm_inputs(b)
    with pytest.raises(RuntimeError):
        m(unet_in, ts, ehs, added_time_ids=ids)


def test_rand_log_normal_matches_reference_formula():
    g = torch.Generator().manual_seed(5)
    s = rand_log_normal([1000], loc=0.7, scale=1.6, generator=g)
    assert (s > 0).all() and abs(float(s.log().mean()) - 0.7) < 0.2


def test_oracle_reproduces_committed_golden_step():
    """tests/golden/tiny_step.safetensors (made by tests/golden/make_golden.py) freezes one seeded oracle train step, full-parameter
    and LoRA: the restatement must keep reproducing it (same torch CPU kernels -> tight tolerance)."""
    import os
    import sys

    from safetensors.torch import load_file
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import e2e_checks
    from oracle.unet import TINY_CONFIG
    gold = load_file(os.path.join(here, "golden", "tiny_step.safetensors"))
    for tag, r in (("full", 0), ("lora8", 8)):
        ref = e2e_checks.oracle_step(TINY_CONFIG, 1, 4, 16, 16, seed=3, lr=1e-4, cross_dim=TINY_CONFIG["cross_attention_dim"], lora_r=r)
        assert abs(ref["loss"] - float(gold[f"{tag}.loss"])) <= 1e-5 * abs(ref["loss"]), tag
        assert torch.allclose(ref["pred"], gold[f"{tag}.pred"], atol=1e-4, rtol=1e-4), tag
        names = open(os.path.join(here, "golden", f"tiny_step_{tag}_grad_names.txt")).read().split()
        assert names == sorted(ref["grads"]), tag
        norms = torch.tensor([float(ref["grads"][n].double().norm()) for n in names], dtype=torch.float64)
        assert torch.allclose(norms, gold[f"{tag}.grad_norms"], rtol=1e-3, atol=1e-9), tag


def test_chunked_attention_equals_the_explicit_form(monkeypatch):
    """`oracle.unet._ChunkedAttention` (the memory-frugal path the oracle takes for config 4's 9216-pixel level) against the explicit
    softmax(QK^T s)V of `Attention.forward`: the same module, the same input, once under the byte limit and once over it -- output and
    every gradient (input, q/k/v/out weights), with a sequence that is not a multiple of the chunk and cross-attention-shaped keys."""
    from oracle import unet as ou
    torch.manual_seed(3)
    monkeypatch.setattr(ou._ChunkedAttention, "CHUNK", 48)
    for s, kv, cross in ((130, 130, None), (40, 7, 24)):
        att = ou.Attention(64, heads=2, dim_head=32, cross_attention_dim=cross).double()
        x = torch.randn(3, s, 64, dtype=torch.float64, requires_grad=True)
        ctx = None if cross is None else torch.randn(3, kv, cross, dtype=torch.float64)
        w = torch.randn(3, s, 64, dtype=torch.float64)
        outs = []
        for limit in (1 << 40, 0):
            monkeypatch.setattr(ou.Attention, "SCORE_BYTES_LIMIT", limit)
            att.zero_grad()
            x.grad = None
            y = att(x, ctx)
            (y * w).sum().backward()
            outs.append((y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in att.parameters()]))
        (y0, gx0, gp0), (y1, gx1, gp1) = outs
        assert torch.allclose(y0, y1, rtol=1e-12, atol=1e-13)
        assert torch.allclose(gx0, gx1, rtol=1e-10, atol=1e-12)
        for a, b in zip(gp0, gp1):
            assert torch.allclose(a, b, rtol=1e-10, atol=1e-12)


def test_recompute_mode_changes_nothing(monkeypatch):
    """oracle.unet.RECOMPUTE (torch.utils.checkpoint around every resnet / transformer module, used for the cached references of config 4's
    upper levels) against the default mode: loss, prediction and every gradient of one seeded step, bit for bit."""
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import e2e_checks
    from oracle import unet as ou
    a = e2e_checks.oracle_step(TINY_CONFIG, 1, 3, 16, 16, seed=3, lr=1e-4, cross_dim=TINY_CONFIG["cross_attention_dim"], with_pred_after=False)
    monkeypatch.setattr(ou, "RECOMPUTE", True)
    b = e2e_checks.oracle_step(TINY_CONFIG, 1, 3, 16, 16, seed=3, lr=1e-4, cross_dim=TINY_CONFIG["cross_attention_dim"], with_pred_after=False)
    assert a["loss"] == b["loss"] and torch.equal(a["pred"], b["pred"])
    assert sorted(a["grads"]) == sorted(b["grads"]) and all(torch.equal(a["grads"][n], b["grads"][n]) for n in a["grads"])
