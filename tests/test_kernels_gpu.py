"""`-m gpu`: every libsvdx kernel against its fp32 torch emulation, through the C-ABI."""
import math

import pytest
import torch

gpu = pytest.mark.gpu
GROUPS = ["gemm_tn", "gemm_geglu", "gemm_plain_v4", "gemm_gather_v4", "gemm_plain_v6", "gemm_gather_v6", "gemm_plain_v0", "gemm_plain_v1", "gemm_gather_v0", "gemm_gather_v1", "small", "groupnorm", "layernorm",
          "attention", "temporal_attention", "elementwise", "optim"]


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
    fns = {"gemm_tn": lambda: kc.check_gemm_tn(pair, dt), "gemm_geglu": lambda: kc.check_gemm_geglu(pair, dt),
           "gemm_plain_v4": lambda: kc.check_gemm_plain(pair, dt, 4), "gemm_gather_v4": lambda: kc.check_gemm_gather(pair, dt, 4),
           "gemm_plain_v6": lambda: kc.check_gemm_plain(pair, dt, 6), "gemm_gather_v6": lambda: kc.check_gemm_gather(pair, dt, 6),
           "gemm_plain_v0": lambda: kc.check_gemm_plain(pair, dt, 0), "gemm_plain_v1": lambda: kc.check_gemm_plain(pair, dt, 1),
           "gemm_gather_v0": lambda: kc.check_gemm_gather(pair, dt, 0), "gemm_gather_v1": lambda: kc.check_gemm_gather(pair, dt, 1),
           "small": lambda: kc.check_small(pair, dt), "groupnorm": lambda: kc.check_groupnorm(pair, dt),
           "layernorm": lambda: kc.check_layernorm(pair, dt), "attention": lambda: kc.check_attention(pair, dt),
           "temporal_attention": lambda: kc.check_temporal_attention(pair, dt),
           "elementwise": lambda: kc.check_elementwise(pair, dt), "optim": lambda: kc.check_optim(pair, dt)}
    bad = [(l, e, t) for l, e, t in fns[group]() if not (e <= t and math.isfinite(e))]
    assert not bad, f"{len(bad)} mismatches, first: {bad[:5]}"


@gpu
def test_extension_is_loaded_and_native():
    """The product path must be the HIP library (no silent fallback)."""
    from svd_xtend_amd import kernels as K
    be = K.backend()
    assert isinstance(be, K.HipBackend) and be.lib.svdx_device_ok() == 1
