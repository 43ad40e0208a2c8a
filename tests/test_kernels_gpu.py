"""`-m gpu`: every libsvdx kernel against its fp32 torch emulation, through the C-ABI."""
import math
import os

import pytest
import torch

gpu = pytest.mark.gpu
RING = (16, 18, 20, 23, 25)  # ring-staged tile variants: between them every instantiation of gemm_v4_kernel (kernel_checks.RING_VARIANTS: N % 160 picks the 160- or 128-wide sibling)
GROUPS = (["gemm_tn", "gemm_tn_s3", "gemm_tn_s4", "gemm_tn_v18", "gemm_geglu", "gemm_geglu_v17", "gemm_geglu_v18", "gemm_geglu_v21", "gemm_geglu_v26", "gemm_plain_v1"]
          + [f"gemm_{k}_v{v}" for v in (4, 6) + RING for k in ("plain", "gather")]
          + [f"gemm_gn_v{v}" for v in (4, 6, 8, 16, 18, 22, 23, 24, 26)]     # svdx_gemm_gn: GroupNorm statistics from the store loop of every tile family
          + [f"gemm_{k}_v{v}" for v in (27, 28) for k in ("plain", "gather")] + ["gemm_geglu_v27"]     # tuner candidates (ops.STAGED_TILES)
          + [f"gemm_{k}_v{v}" for v in (32, 34) for k in ("plain", "gather", "gn", "geglu")]     # two-role eight-wave tiles (gemm_v5_kernel)
          + [f"gemm_{k}_v36" for k in ("plain", "gather", "gn")]     # round 6: 144 x 160 six-wave tile stepping 140 rows
          + ["large_offsets", "small", "groupnorm", "layernorm", "attention", "temporal_attention", "tsa", "encoders", "elementwise", "optim"])


@pytest.fixture(scope="module")
def pair():
    import kernel_checks as kc
    from svd_xtend_amd import kernels as K
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return kc.Pair(K.backend(), torch.device("cuda"))


@gpu
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
@pytest.mark.parametrize("group", GROUPS)
def test_kernel_group(pair, group, dt):
    import kernel_checks as kc
    fns = {"gemm_tn": lambda: kc.check_gemm_tn(pair, dt), "gemm_tn_s3": lambda: kc.check_gemm_tn(pair, dt, 3), "gemm_tn_s4": lambda: kc.check_gemm_tn(pair, dt, 4),
           "gemm_geglu": lambda: kc.check_gemm_geglu(pair, dt), "gemm_plain_v1": lambda: kc.check_gemm_plain(pair, dt, 1),
           "large_offsets": lambda: kc.check_large_offsets(pair, dt) if dt == torch.float16 else [],
           "small": lambda: kc.check_small(pair, dt), "groupnorm": lambda: kc.check_groupnorm(pair, dt),
           "layernorm": lambda: kc.check_layernorm(pair, dt), "attention": lambda: kc.check_attention(pair, dt),
           "temporal_attention": lambda: kc.check_temporal_attention(pair, dt),
           "tsa": lambda: kc.check_tsa(pair, dt), "encoders": lambda: kc.check_encoders(pair, dt),
           "elementwise": lambda: kc.check_elementwise(pair, dt), "optim": lambda: kc.check_optim(pair, dt)}
    for v in (18,):
        fns[f"gemm_tn_v{v}"] = lambda v=v: kc.check_gemm_tn(pair, dt, v)
    for v in (4, 6, 8, 16, 18, 22, 23, 24, 26, 32, 34, 36):
        fns[f"gemm_gn_v{v}"] = lambda v=v: kc.check_gemm_gn(pair, dt, v)
    for v in (17, 18, 21, 26, 27, 32, 34):
        fns[f"gemm_geglu_v{v}"] = lambda v=v: kc.check_gemm_geglu(pair, dt, v)
    for v in (4, 6, 27, 28, 32, 34, 36) + RING:
        fns[f"gemm_plain_v{v}"] = lambda v=v: kc.check_gemm_plain(pair, dt, v)
        fns[f"gemm_gather_v{v}"] = lambda v=v: kc.check_gemm_gather(pair, dt, v)
    bad = [(l, e, t) for l, e, t in fns[group]() if not (e <= t and math.isfinite(e))]
    assert not bad, f"{len(bad)} mismatches, first: {bad[:5]}"


@gpu
def test_extension_is_loaded_and_native():
    """The product path must be the HIP library (no silent fallback)."""
    from svd_xtend_amd import kernels as K
    be = K.backend()
    assert isinstance(be, K.HipBackend) and be.lib.svdx_device_ok() == 1


@gpu
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_edm_loss_kernel_matches_reference_statements(dt):
    """svdx_edm_loss against the loss computed by the reference's own statements (tests/golden/step_math.safetensors, made by
    tests/golden/make_golden_step_math.py from train_svd.py:1025-1036); the prediction is rounded to the activation dtype first."""
    import os
    import sys

    from safetensors.torch import load_file
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    from make_golden_step_math import CASES, case_inputs
    from svd_xtend_amd import kernels as K
    k, dev = K.backend(), torch.device("cuda")
    g = load_file(os.path.join(here, "golden", "step_math.safetensors"))
    for i, (bsz, T, h, w, D, prob, seed) in enumerate(CASES):
        inp = case_inputs(bsz, T, h, w, D, seed)
        C = 4
        pred16 = inp["model_pred"].permute(0, 1, 3, 4, 2).reshape(bsz * T * h * w, C).to(dt)
        sig5 = g[f"case{i}.sigmas"]
        # the reference loss on the SAME rounded prediction (the golden value itself uses the float prediction)
        pr = pred16.float().view(bsz, T, h, w, C).permute(0, 1, 4, 2, 3)
        c_out, c_skip = -sig5 / ((sig5 ** 2 + 1) ** 0.5), 1 / (sig5 ** 2 + 1)
        wgt = (1 + sig5 ** 2) * (sig5 ** -2.0)
        want = (wgt * (pr * c_out + c_skip * g[f"case{i}.noisy_latents"] - inp["latents"]) ** 2).reshape(bsz, -1).mean(1).mean()
        st = torch.zeros(K.OPT_STATE_FLOATS, device=dev)
        st[1] = 1.0
        loss = torch.zeros(1, device=dev)
        dpred = torch.zeros(bsz * T * h * w, 64, dtype=dt, device=dev)
        k.edm_loss(pred16.to(dev).contiguous(), C, g[f"case{i}.noisy_latents"].to(dev).contiguous(), inp["latents"].to(dev).contiguous(),
                   sig5.reshape(-1).to(dev).contiguous(), loss, dpred, bsz, T, C, h * w, st)
        torch.cuda.synchronize()
        assert abs(float(loss) - float(want)) <= 1e-5 * float(want), (i, float(loss), float(want))
        # and within the rounding of the prediction of the golden loss itself
        assert abs(float(loss) - float(g[f"case{i}.loss"])) <= (2e-2 if dt == torch.bfloat16 else 3e-3) * float(g[f"case{i}.loss"]), i


@gpu
@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_tsa_outputs_are_fully_written_and_reproducible(dt):
    """The fused temporal self-attention writes every row of n, stats, q/k/v, o and h1 (outputs pre-filled with NaN come back
    finite) and gives the same bits on every launch whatever the buffers held before -- the regression test of the wide-store data
    hazard (tests/test_store_hazard.py): the image side copy once lost ~1 element per million of `o` at the small widths."""
    from svd_xtend_amd import kernels as K
    k = K.backend()
    dev = torch.device("cuda")

    def run(B, T, HW, heads, fill):
        C, M = heads * 64, B * T * HW
        g = torch.Generator().manual_seed(5)
        x = torch.randn(M, C, generator=g).to(dt).to(dev)
        gamma, beta = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
        wqkv = (torch.randn(3 * C, C, generator=g) * C ** -0.5).to(dt).to(dev)
        wo = (torch.randn(C, C, generator=g) * C ** -0.5).to(dt).to(dev)
        bo, cvec = (0.1 * torch.randn(C, generator=g)).to(dev), torch.randn(B, C, generator=g).to(dev)
        outs = [torch.full((M, C), fill, dtype=dt, device=dev), torch.full((M, 2), fill, device=dev),
                torch.full((M, 3 * C), fill, dtype=dt, device=dev), torch.full((M, C), fill, dtype=dt, device=dev),
                torch.full((M, C), fill, dtype=dt, device=dev)]
        k.tsa_fwd(x, gamma, beta, 1e-5, wqkv, wo, bo, cvec, C, T * HW, 0, *outs, B, T, HW, C, heads, 0.125)
        torch.cuda.synchronize()
        return outs

    for shape in [(1, 4, 256, 1), (1, 4, 64, 2), (2, 3, 96, 2), (1, 14, 640, 5)]:
        ref = run(*shape, fill=float("nan"))
        assert all(bool(torch.isfinite(t.float()).all()) for t in ref), shape
        for rep in range(6):
            again = run(*shape, fill=float(rep))
            for name, a, b in zip(("n", "stats", "qkv", "o", "h1"), ref, again):
                assert torch.equal(a, b), (shape, rep, name)
