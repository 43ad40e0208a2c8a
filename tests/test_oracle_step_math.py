"""The train-step arithmetic that lives in the reference file itself, against values computed by the reference's own statements
(tests/golden/make_golden_step_math.py lifts train_svd.py:964-972, :992-1017, :1020, :1025-1036 out of `main()` and runs them here):
the oracle's restatement (oracle/step.py) and the product's host-side data prep (svd_xtend_amd.train.edm_prepare /
conditioning_dropout) must reproduce them."""
import os
import sys

import torch
from safetensors.torch import load_file

from oracle.step import conditioning_dropout, edm_inputs, edm_loss

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def _cases():
    from make_golden_step_math import CASES, case_inputs
    g = load_file(os.path.join(HERE, "golden", "step_math.safetensors"))
    for i, (bsz, T, h, w, D, prob, seed) in enumerate(CASES):
        inp = case_inputs(bsz, T, h, w, D, seed)
        random_p = torch.rand(bsz, generator=torch.Generator().manual_seed(2000 + seed)) if prob is not None else None
        yield i, prob, inp, random_p, {k.split(".", 1)[1]: v for k, v in g.items() if k.startswith(f"case{i}.")}


def test_oracle_step_math_matches_reference_statements():
    n_masked = 0
    for i, prob, inp, random_p, gold in _cases():
        sig = gold["sigmas"].reshape(-1)
        ehs, cond = inp["encoder_hidden_states"], inp["conditional_latents"]
        if prob is not None:
            ehs, cond = conditioning_dropout(random_p, ehs, cond, prob)
            n_masked += int((ehs.abs().sum((1, 2)) == 0).sum()) + int((cond.abs().sum((1, 2, 3)) == 0).sum())
        batch = dict(latents=inp["latents"], noise=inp["noise"], cond_latents=cond, ehs=ehs, sigmas=sig, cond_sigmas=torch.ones(len(sig)))
        unet_in, ts, ehs_out, _, noisy, sig5 = edm_inputs(batch)
        assert torch.equal(noisy, gold["noisy_latents"]), i
        assert torch.equal(ts, gold["timesteps"]), i
        assert torch.equal(unet_in, gold["inp_noisy_latents"]), i
        assert torch.equal(ehs_out, gold["encoder_hidden_states"]), i
        loss = edm_loss(inp["model_pred"], noisy, inp["latents"], sig5)
        assert torch.equal(loss, gold["loss"]), (i, float(loss), float(gold["loss"]))
    assert n_masked >= 4            # the seeded cases exercise both masks


def test_add_time_ids_match_reference_function():
    from oracle.step import get_add_time_ids
    g = load_file(os.path.join(HERE, "golden", "step_math.safetensors"))
    assert torch.equal(get_add_time_ids(7, 127, torch.tensor(0.0625), torch.float32, 3), g["add_time_ids"])
    assert g["add_time_ids"].tolist() == [[7.0, 127.0, 0.0625]] * 3


def test_product_data_prep_matches_reference_statements():
    from svd_xtend_amd.train import conditioning_dropout as product_dropout
    from svd_xtend_amd.train import edm_prepare
    for i, prob, inp, random_p, gold in _cases():
        ehs, cond = inp["encoder_hidden_states"], inp["conditional_latents"]
        if prob is not None:
            ehs, cond = product_dropout(random_p, ehs, cond, prob)
            assert torch.equal(ehs, gold["encoder_hidden_states"]), i
        unet_in, ts, noisy = edm_prepare(inp["latents"], inp["noise"], cond, gold["sigmas"].reshape(-1))
        assert torch.equal(noisy, gold["noisy_latents"]) and torch.equal(unet_in, gold["inp_noisy_latents"]), i
        assert float((ts - gold["timesteps"]).abs().max()) <= 1e-6, i       # vectorised log instead of a Python loop over sigmas


def test_emulated_loss_kernel_matches_reference_loss(emu_backend):
    """svdx_edm_loss (through its emulation on CPU; the GPU kernel is checked against the emulation in test_kernels_gpu.py)."""
    from svd_xtend_amd import kernels as K
    k = K.backend()
    for i, prob, inp, random_p, gold in _cases():
        B, T, C, h, w = inp["model_pred"].shape
        pred_rows = inp["model_pred"].permute(0, 1, 3, 4, 2).reshape(B * T * h * w, C).contiguous()        # rows [B*T*HW, C]
        st = torch.zeros(K.OPT_STATE_FLOATS)
        st[1] = 1.0
        loss, dpred = torch.zeros(1), torch.zeros(B * T * h * w, C)
        k.edm_loss(pred_rows, C, gold["noisy_latents"].contiguous(), inp["latents"].contiguous(), gold["sigmas"].reshape(-1).contiguous(),
                   loss, dpred, B, T, C, h * w, st)
        assert abs(float(loss) - float(gold["loss"])) <= 2e-6 * float(gold["loss"]), (i, float(loss), float(gold["loss"]))
