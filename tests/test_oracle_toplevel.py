"""The UNet's top level against the reference's OWN class.  tests/golden/make_golden_unet_toplevel.py executes
/root/reference/src/unet_spatio_temporal_condition.py unmodified in this container, with diffusers' block factories standing in as
the oracle's blocks, loads the oracle's weights into it (strict) and stores its outputs: constructor topology and forward
orchestration of the oracle -- and, through the emulated kernels, of the product -- are held to the reference's code."""
import os
import sys

import torch
from safetensors.torch import load_file

from oracle.unet import TINY_CONFIG, UNetSpatioTemporalConditionOracle, scaled_init_

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _gold():
    return load_file(os.path.join(HERE, "golden", "unet_toplevel.safetensors"))


def test_oracle_top_level_equals_reference_class_output():
    from make_golden_unet_toplevel import CASES, toplevel_inputs
    g = _gold()
    orc = UNetSpatioTemporalConditionOracle(**TINY_CONFIG)
    scaled_init_(orc, 0)
    with torch.no_grad():
        for i, (B, T, h, w, seed) in enumerate(CASES):
            x, t, ehs, ids = toplevel_inputs(B, T, h, w, seed, TINY_CONFIG["cross_attention_dim"])
            y = orc(x, t, ehs, added_time_ids=ids).sample
            assert y.shape == g[f"case{i}.sample"].shape == (B, T, 4, h, w)
            assert float((y - g[f"case{i}.sample"]).abs().max()) <= 1e-6, i


def test_reference_constructor_counts():
    """Parameter counts produced by the reference's own constructor at its default (SVD) configuration."""
    assert _gold()["full_counts"].tolist() == [1_524_623_082, 397_620_480]


def test_trainable_selection_equals_reference_loop():
    """select_trainable (product) and trainable_names (oracle) against the parameter list the reference's own loop
    (train_svd.py:758-766, executed on the reference class instance by the golden script) put into its optimizer."""
    from oracle.unet import trainable_names
    from svd_xtend_amd.train import select_trainable
    from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
    want = open(os.path.join(HERE, "golden", "unet_toplevel_trainable_names.txt")).read().split()
    assert len(want) > 100 and all("temporal_transformer_block" in n for n in want)
    assert list(trainable_names(UNetSpatioTemporalConditionOracle(**TINY_CONFIG))) == want
    m = UNetSpatioTemporalConditionModel(**TINY_CONFIG)
    assert select_trainable(m) == want
    assert [n for n, p in m.named_parameters() if p.requires_grad] == want


def test_product_forward_equals_reference_class_output(emu_backend):
    from make_golden_unet_toplevel import CASES, toplevel_inputs
    from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
    g = _gold()
    orc = UNetSpatioTemporalConditionOracle(**TINY_CONFIG)
    scaled_init_(orc, 0)
    m = UNetSpatioTemporalConditionModel(**TINY_CONFIG)
    m.load_state_dict(orc.state_dict(), strict=True)
    for p in m.parameters():
        p.requires_grad_(False)
    m._requested_dtype = torch.float32              # the emulated kernels store activations in the requested dtype
    m.prepare()
    with torch.no_grad():
        for i, (B, T, h, w, seed) in enumerate(CASES):
            x, t, ehs, ids = toplevel_inputs(B, T, h, w, seed, TINY_CONFIG["cross_attention_dim"])
            y = m(x, t, ehs, added_time_ids=ids).sample
            ref = g[f"case{i}.sample"]
            assert float((y - ref).abs().max()) <= 2e-4 * float(ref.abs().max()), (i, float((y - ref).abs().max()))
