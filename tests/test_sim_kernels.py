"""CPU (`-m "not gpu"`): the HIP kernel sources themselves, executed lane by lane by the wave64 functional simulator of tests/sim/
(fibers per thread, rendezvous per cross-lane instruction, LDS-DMA landing at the `s_waitcnt` that retires it), against the same fp32
emulation and the same CPU oracle the `-m gpu` tests use.  This checks the kernels' LOGIC -- fragment layouts, swizzles, tile maps,
barrier and wait placement -- on a box without a GPU; it says nothing about speed, and the GPU tests remain the parity tests proper.

SVDX_SIM_FULL=1 adds every kernel group in both dtypes (about nine minutes on eight cores) and the mutation tests that show the
simulator notices a missing wait / barrier."""
import math
import os
import shutil
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "sim"))
FULL = os.environ.get("SVDX_SIM_FULL") == "1"

QUICK = [("gemm_plain_v4", "f16"), ("gemm_gn_v6", "f16"), ("gemm_gn_v23", "bf16"), ("gemm_gather_v16", "f16"), ("gemm_plain_v23", "bf16"), ("gemm_tn", "bf16"), ("gemm_tn_v18", "f16"), ("gemm_tn_flat", "f16"), ("gemm_tn_v18_flat", "bf16"), ("gemm_geglu_v26", "f16"),
         ("groupnorm", "f16"), ("layernorm", "bf16"), ("temporal_attention", "f16"), ("temporal_attention", "bf16"), ("tsa", "f16"), ("small", "f16"),
         ("encoders", "f16"), ("elementwise", "bf16"), ("optim", "f16")]
DT = {"f16": torch.float16, "bf16": torch.bfloat16}


@pytest.fixture(scope="module")
def sim_pair():
    import kernel_checks as kc
    from backend import SimBackend
    return kc.Pair(SimBackend(), torch.device("cpu"))


def _all_cases():
    import run_checks
    names = list(run_checks.groups(None, None))
    return [(n, d) for n in names for d in ("f16", "bf16")]


@pytest.mark.parametrize("group,dt", _all_cases() if FULL else QUICK)
def test_kernel_group_on_simulator(sim_pair, group, dt):
    import run_checks
    res = run_checks.groups(sim_pair, DT[dt])[group]()
    bad = [(l, e, t) for l, e, t in res if not (e <= t and math.isfinite(e))]
    assert res and not bad, f"{len(bad)} mismatches, first: {bad[:5]}"


def test_train_step_on_simulator_matches_oracle_tiny():
    """The tiny SVD UNet topology: forward + EDM loss + backward + AdamW through the simulated kernels against the CPU oracle (what
    __graft_entry__.smoke() checks on the GPU), same bar: loss within 1e-3 relative, every gradient's cosine >= 0.99."""
    import e2e_checks
    from backend import SimBackend
    from oracle.unet import TINY_CONFIG
    from svd_xtend_amd import kernels as K
    prev = K._backend
    K._set_backend_for_tests(SimBackend())
    try:
        cfg = TINY_CONFIG
        ref = e2e_checks.oracle_step(cfg, 1, 4, 16, 16, seed=3, lr=1e-4, cross_dim=cfg["cross_attention_dim"])
        got = e2e_checks.compare(ref, e2e_checks.product_step(ref, cfg, torch.float16, torch.device("cpu"), 1e-4))
    finally:
        K._set_backend_for_tests(prev)
    assert got["loss_rel"] <= 1e-3 and got["grad_cos_min"] >= 0.99, got


def test_batched_skinny_launches_equal_single_launches_on_simulator():
    """Two fp16 optimizer steps of the tiny topology with the table-driven skinny launches and with one launch each end in identical
    bits (the GPU suite runs the same comparison on the hardware)."""
    import e2e_checks
    from backend import SimBackend
    from svd_xtend_amd import kernels as K
    prev = K._backend
    K._set_backend_for_tests(SimBackend())
    try:
        a, b = e2e_checks.batched_vs_single_small_launches(dev=torch.device("cpu"))
        la, lb = e2e_checks.batched_vs_single_small_launches(dev=torch.device("cpu"), dtype=torch.bfloat16, steps=1, lora_r=8) if FULL else (None, None)
    finally:
        K._set_backend_for_tests(prev)
    e2e_checks.assert_batched_equals_single(a, b)
    if FULL:                                                   # config 5's adapters on the cross-attention value path: four + three stages
        e2e_checks.assert_batched_equals_single(la, lb, lora=True)


@pytest.mark.skipif(not FULL, reason="SVDX_SIM_FULL=1 (about two minutes)")
def test_lora_step_and_replay_on_simulator():
    """Reference config 5's path (adapters through the dual-operand GEMM loop, zero-padded rank 8) against the CPU oracle, and two runs of
    two bf16 optimizer steps ending in identical bits -- the GPU suite's LoRA and determinism tests, on the simulated kernels."""
    import e2e_checks
    from backend import SimBackend
    from svd_xtend_amd import kernels as K
    prev = K._backend
    K._set_backend_for_tests(SimBackend())
    try:
        dev = torch.device("cpu")
        for key, r in e2e_checks.run_lora(dev=dev, ranks=(8,)).items():
            e2e_checks.assert_parity(key, r)
        a = e2e_checks.run_steps(dev=dev, dtype=torch.bfloat16, steps=2)
        b = e2e_checks.run_steps(dev=dev, dtype=torch.bfloat16, steps=2)
        assert a["loss"] == b["loss"] and all(torch.equal(a[k], b[k]) for k in ("p", "m", "v"))
    finally:
        K._set_backend_for_tests(prev)


@pytest.mark.skipif(not FULL, reason="SVDX_SIM_FULL=1 (about a minute)")
def test_drop_in_autograd_route_on_simulator():
    """INTEGRATION.md's minimal-change route -- `unet(...).sample`, `loss.backward()` through `_UNetFn` on autograd's worker thread, the
    host's own torch.optim.AdamW, `refresh_trainable()` -- on the simulated kernels against the oracle's step."""
    import e2e_checks
    from backend import SimBackend
    from svd_xtend_amd import kernels as K
    prev = K._backend
    K._set_backend_for_tests(SimBackend())
    try:
        e2e_checks.assert_parity("autograd route", e2e_checks.autograd_route(dev=torch.device("cpu"), dtype=torch.float16))
    finally:
        K._set_backend_for_tests(prev)


@pytest.mark.parametrize("dt", ["f16", "bf16"])
def test_random_shapes_on_simulator(sim_pair, dt):
    """tests/sim/fuzz.py: seeded random shapes, strides, epilogue modes, tile variants and split factors of every kernel family against
    the emulation -- the shapes a caller of the C-ABI may pass, not only the UNet's (this is how the GEGLU-backward partial-tile hole
    behind svdx_gemm's argument check was found)."""
    import fuzz
    bad = fuzz.run(sim_pair, DT[dt], list(fuzz.FAMILIES), 60 if FULL else 10, seed=1 if dt == "f16" else 2, verbose=False)
    assert not bad, bad[:5]


def test_geglu_backward_refuses_partial_column_tiles(sim_pair):
    """The GEGLU-backward epilogue lives in the coalesced store path only (whole column tiles): a 128-wide tile variant on N = 320 is an
    argument error, and the host rule never asks for one."""
    from svd_xtend_amd import kernels as K
    from svd_xtend_amd import ops
    dt, M, C, F = torch.float16, 70, 64, 320
    dy, W2t, pre = torch.zeros(M, C, dtype=dt), torch.zeros(F, C, dtype=dt), torch.zeros(M, 2 * F, dtype=dt)
    for v in (8, 21, 26, 18, 24):
        with pytest.raises(K.SvdxError, match="GEGLU bwd"):
            sim_pair.impl.gemm(dy, W2t, torch.zeros(M, 2 * F, dtype=dt), M, F, C, C, C, 2 * F, variant=v, epilogue=K.EPI_GEGLU_BWD, aux_in=pre, aux_dim=F)
    for M_ in (70, 560, 2240, 35840):
        assert ops.choose_geglu_variant(M_, 320, 64, fwd=False) == 4
        assert ops.GEGLU_TWO_PER_CU not in ops.geglu_candidates(M_, 320, 64, fwd=False)
        assert ops.choose_geglu_variant(M_, 1280, 320, fwd=False) in (21, ops.GEGLU_TWO_PER_CU)


def test_simulated_library_is_not_the_product_library():
    """The product binding refuses to run without a GPU; the simulator build is a separate file that only tests construct."""
    import build_sim
    from svd_xtend_amd import kernels as K
    assert os.path.realpath(build_sim.LIB) != os.path.realpath(K.LIB_PATH)
    pkg = os.path.join(HERE, "..", "svd_xtend_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            text = open(os.path.join(pkg, f)).read()
            assert "libsvdx_sim" not in text and "tests.sim" not in text and "build_sim" not in text, f
    if not torch.cuda.is_available():
        with pytest.raises(K.SvdxError):
            K.HipBackend()


MUTATIONS = {
    # a counted wait that leaves one tile too many in flight: the ring's next stage is read before its LDS-DMA has landed
    "lax_vmcnt": ('if (t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPIECE) : "memory");',
                  'if (t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPIECE) : "memory");', "gemm_plain_v16"),
    # no workgroup barrier between the K-steps of the ring: a fast wave overwrites the stage a slow wave still reads
    "no_barrier": ("__builtin_amdgcn_s_barrier();", "/* removed */", "gemm_plain_v16"),
}


@pytest.mark.skipif(not FULL, reason="SVDX_SIM_FULL=1: rebuilds the simulator library from a mutated copy of gemm.hip (about a minute each)")
@pytest.mark.parametrize("name", sorted(MUTATIONS))
def test_simulator_notices_mutation(tmp_path, name):
    import subprocess
    old, new, group = MUTATIONS[name]
    csrc = os.path.join(HERE, "..", "svd_xtend_amd", "csrc")
    mut = tmp_path / "csrc"
    mut.mkdir()
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".cpp")):
            shutil.copy(os.path.join(csrc, f), mut / f)
    src = (mut / "gemm.hip").read_text()
    assert old in src
    (mut / "gemm.hip").write_text(src.replace(old, new))
    env = dict(os.environ, SVDX_SIM_CSRC=str(mut), SVDX_SIM_OUT=str(tmp_path / "out"))
    r = subprocess.run([sys.executable, os.path.join(HERE, "sim", "run_checks.py"), group, "--f16"], env=env, capture_output=True, text=True)
    assert r.returncode == 1 and " bad " in r.stdout and "  0 bad" not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
