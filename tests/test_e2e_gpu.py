"""`-m gpu`: train-step parity of the HIP UNet against the CPU oracle (loss rel err <= 1e-3 for fp16)."""
import pytest
import torch

gpu = pytest.mark.gpu


@gpu
def test_train_step_matches_oracle_tiny():
    import e2e_checks
    res = e2e_checks.run_all(verbose=True)
    for key, r in res.items():
        assert "error" not in r, f"{key}: {r}"
        tol = 1e-3 if "float16" in key else 8e-3     # north_star tolerance is stated for fp16; bf16 has 8x less mantissa
        assert r["loss_rel"] <= tol, f"{key}: loss rel err {r['loss_rel']:.3e}"
        assert r["grad_cos_min"] >= (0.99 if "float16" in key else 0.95), f"{key}: {r}"
