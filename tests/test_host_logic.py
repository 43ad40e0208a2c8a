"""Host-side orchestration (explicit forward/backward schedule, packing, flat buffers, optimizer plumbing) checked
against the CPU oracle, with the libsvdx kernels replaced by their torch emulation (tests/emul.py).  At fp32 storage
the hand-written backward must reproduce autograd to rounding; at fp16/bf16 storage the north-star tolerance applies."""
import pytest
import torch

import e2e_checks
from oracle.step import edm_inputs, edm_loss, make_optimizer, make_synthetic_batch
from oracle.unet import TINY_CONFIG, UNetSpatioTemporalConditionOracle, scaled_init_
from svd_xtend_amd.train import Trainer, select_trainable
from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel

CPU = torch.device("cpu")


def build_pair(seed=0):
    orc = UNetSpatioTemporalConditionOracle(**TINY_CONFIG)
    scaled_init_(orc, seed)
    m = UNetSpatioTemporalConditionModel(**TINY_CONFIG)
    m.load_state_dict(orc.state_dict(), strict=True)
    return orc, m


def test_state_dict_keys_and_shapes_match_diffusers_layout():
    orc, m = build_pair()
    a, b = orc.state_dict(), m.state_dict()
    assert set(a) == set(b)
    assert all(a[k].shape == b[k].shape for k in a)
    assert select_trainable(m) == [n for n, _ in orc.named_parameters() if "temporal_transformer_block" in n]
    assert m.add_embedding.linear_1.in_features == 3 * m.config.addition_time_embed_dim      # train_svd.py:887-889


@pytest.mark.parametrize("B,T,h,w", [(1, 3, 16, 16), (1, 4, 16, 16), (1, 4, 32, 32), (2, 2, 16, 24)])
def test_fp32_train_step_matches_oracle(emu_backend, B, T, h, w):
    orc, m = build_pair(1)
    batch = make_synthetic_batch(B, T, h, w, 7, cross_dim=64)
    opt = make_optimizer(orc, lr=1e-3)
    unet_in, ts, ehs, ids, noisy, sig = edm_inputs(batch)
    pred = orc(unet_in, ts, ehs, added_time_ids=ids).sample
    loss = edm_loss(pred, noisy, batch["latents"], sig)
    loss.backward()
    grads = {n: p.grad.clone() for n, p in orc.named_parameters() if p.grad is not None}
    tr = Trainer(m, dtype=torch.float32, lr=1e-3)
    tr.zero_grad()
    tr.forward_backward(unet_in, ts, ehs, ids, noisy, batch["latents"], batch["sigmas"])
    assert abs(float(tr.last_loss()) - float(loss)) / float(loss) < 1e-5
    for n, p in m.named_parameters():
        if p.requires_grad:
            g = grads[n]
            assert float((p.grad - g).abs().max()) <= 2e-4 * float(g.abs().max()) + 1e-7, n
    # KV-length-1 cross attention: to_q / to_k and the LayerNorm feeding to_q get exactly zero gradient
    z = [n for n, p in m.named_parameters() if p.requires_grad and ("attn2.to_q" in n or "attn2.to_k" in n or "norm2" in n)]
    assert z and all(float(dict(m.named_parameters())[n].grad.abs().max()) == 0.0 for n in z)


def test_drop_in_forward_and_autograd_boundary(emu_backend):
    """`unet(...).sample` + `loss.backward()` as in train_svd.py:1021-1044 reaches the hand-written backward."""
    orc, m = build_pair(2)
    select_trainable(m)
    tr = Trainer(m, dtype=torch.float32)
    batch = make_synthetic_batch(1, 3, 16, 16, 5, cross_dim=64)
    unet_in, ts, ehs, ids, noisy, sig = edm_inputs(batch)
    out = m(unet_in, ts, ehs, added_time_ids=ids).sample
    ref = orc(unet_in, ts, ehs, added_time_ids=ids).sample
    assert out.shape == ref.shape and torch.allclose(out, ref, atol=2e-5)
    tup = m(unet_in, ts, ehs, ids, return_dict=False)
    assert isinstance(tup, tuple) and tup[0].shape == ref.shape
    tr.zero_grad()
    loss = edm_loss(m(unet_in, ts, ehs, ids).sample, noisy, batch["latents"], sig)
    loss.backward()
    make_optimizer(orc)
    edm_loss(ref, noisy, batch["latents"], sig).backward()
    name = "mid_block.attentions.0.temporal_transformer_blocks.0.ff.net.2.weight"
    g_ref = dict(orc.named_parameters())[name].grad
    g = dict(m.named_parameters())[name].grad
    assert torch.allclose(g, g_ref, atol=1e-5 * float(g_ref.abs().max()) + 1e-8)


def test_minimal_change_route_with_torch_optimizer(emu_backend):
    """INTEGRATION.md section 1, first route: host script keeps loss.backward() and torch.optim.AdamW."""
    orc, m = build_pair(8)
    select_trainable(m)
    m.prepare(torch.float32)
    opt_ref = make_optimizer(orc, lr=1e-3)
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-2)
    batch = make_synthetic_batch(1, 2, 16, 16, 9, cross_dim=64)
    unet_in, ts, ehs, ids, noisy, sig = edm_inputs(batch)
    for _ in range(2):
        opt.zero_grad(set_to_none=False)
        edm_loss(m(unet_in, ts, ehs, added_time_ids=ids).sample, noisy, batch["latents"], sig).backward()
        opt.step()
        m.refresh_trainable()
        edm_loss(orc(unet_in, ts, ehs, added_time_ids=ids).sample, noisy, batch["latents"], sig).backward()
        opt_ref.step()
        opt_ref.zero_grad()
    out = m(unet_in, ts, ehs, ids).sample
    ref = orc(unet_in, ts, ehs, added_time_ids=ids).sample
    assert torch.allclose(out, ref, atol=5e-4), float((out - ref).abs().max())


@pytest.mark.parametrize("dt,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
def test_low_precision_storage_meets_loss_tolerance(emu_backend, dt, tol):
    ref = e2e_checks.oracle_step(TINY_CONFIG, 1, 3, 16, 16, seed=3, lr=1e-4, cross_dim=64)
    got = e2e_checks.compare(ref, e2e_checks.product_step(ref, TINY_CONFIG, dt, CPU, 1e-4))
    assert got["loss_rel"] <= tol and got["grad_cos_min"] > 0.99 - (0.04 if dt == torch.bfloat16 else 0), got


def test_loss_scale_skips_step_on_overflow(emu_backend):
    _, m = build_pair(4)
    tr = Trainer(m, dtype=torch.float16, lr=1e-3, init_scale=1024.0)
    before = tr.p_flat.clone()
    tr.zero_grad()
    tr.g_flat[5] = float("inf")
    tr.optimizer_step()
    assert torch.equal(tr.p_flat, before)                       # step skipped
    assert float(tr.opt_state[1]) == 512.0 and float(tr.opt_state[0]) == 0.0
    tr.zero_grad()
    tr.g_flat[:tr.n_flat] = 1.0
    tr.optimizer_step()
    assert not torch.equal(tr.p_flat, before) and float(tr.opt_state[0]) == 1.0


def test_flat_buffers_alias_parameters(emu_backend):
    _, m = build_pair(5)
    tr = Trainer(m, dtype=torch.float32)
    for p, off in zip(tr.params, tr.offsets):
        assert p.data.data_ptr() == tr.p_flat.data_ptr() + 4 * off
        assert p.grad.data_ptr() == tr.g_flat.data_ptr() + 4 * off
    # fused QKV weights of a temporal block are adjacent, so one weight-grad GEMM covers them
    blk = m.mid_block.attentions[0].temporal_transformer_blocks[0]
    assert blk.attn1.qkv.w_grad is not None and blk.attn1.qkv.w_grad.numel() == 3 * blk.dim * blk.dim


def test_backward_stops_before_first_trainable_block(emu_backend):
    _, m = build_pair(6)
    Trainer(m, dtype=torch.float32)
    kinds = [(k, getattr(mod, "need_dx", None)) for k, mod in m.steps if k in ("res", "attn")]
    assert kinds[0] == ("res", False) and kinds[1] == ("attn", False) and kinds[2] == ("res", True)


def test_pretrained_folder_round_trip(tmp_path):
    """`save_pretrained` / `from_pretrained` / `register_to_config` (train_svd.py:651-656, 698-725): diffusers folder layout,
    diffusers key names, strict load, config usable both as attributes and as a mapping."""
    import json

    from oracle.unet import TINY_CONFIG, UNetSpatioTemporalConditionOracle, scaled_init_
    from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
    orc = UNetSpatioTemporalConditionOracle(**TINY_CONFIG)
    scaled_init_(orc, 9)
    m = UNetSpatioTemporalConditionModel(**TINY_CONFIG)
    m.load_state_dict(orc.state_dict(), strict=True)
    m.save_pretrained(str(tmp_path / "ckpt" / "unet"))
    m.save_pretrained(str(tmp_path / "ckpt" / "unet"), variant="fp16")
    cfg = json.load(open(tmp_path / "ckpt" / "unet" / "config.json"))
    assert cfg["block_out_channels"] == list(TINY_CONFIG["block_out_channels"])
    m2 = UNetSpatioTemporalConditionModel.from_pretrained(str(tmp_path / "ckpt"), subfolder="unet")
    assert m2.config.addition_time_embed_dim == TINY_CONFIG["addition_time_embed_dim"]
    assert set(m2.state_dict()) == set(orc.state_dict())
    assert all(torch.equal(v, m2.state_dict()[k]) for k, v in m.state_dict().items())
    m3 = UNetSpatioTemporalConditionModel.from_pretrained(str(tmp_path / "ckpt"), subfolder="unet", variant="fp16")
    k0 = "down_blocks.0.attentions.0.temporal_transformer_blocks.0.attn1.to_q.weight"
    assert torch.equal(m3.state_dict()[k0], m.state_dict()[k0].half().float())
    m3.register_to_config(**m2.config)                   # train_svd.py:723
    assert m3.config["num_frames"] == m2.config.num_frames


@pytest.mark.parametrize("r", [64, 8])
def test_lora_fp32_train_step_matches_oracle(emu_backend, r):
    """Config 5 (train_svd_lora.py:655-674): adapters on every to_q/to_k/to_v/to_out.0, everything else frozen.  Loss and every
    adapter gradient against autograd on the oracle's peft restatement; B is randomised so that dA is exercised too."""
    from oracle.lora import add_adapter
    from svd_xtend_amd.lora import LoraConfig
    orc, m = build_pair(3)
    for p in orc.parameters():
        p.requires_grad_(False)
    n_wrapped = add_adapter(orc, r, r)
    g = torch.Generator().manual_seed(0)
    for n, p in orc.named_parameters():
        if ".lora_B." in n:
            p.data.copy_(torch.randn(p.shape, generator=g) * 0.05)
    assert m.add_adapter(LoraConfig(r=r, lora_alpha=r, init_lora_weights="gaussian")) == n_wrapped
    assert [n for n, _ in m.named_parameters()] == [n for n, _ in orc.named_parameters()]      # peft naming
    m.load_state_dict(orc.state_dict(), strict=True)
    batch = make_synthetic_batch(1, 3, 16, 16, 11, cross_dim=64)
    unet_in, ts, ehs, ids, noisy, sig = edm_inputs(batch)
    pred = orc(unet_in, ts, ehs, added_time_ids=ids).sample
    loss = edm_loss(pred, noisy, batch["latents"], sig)
    loss.backward()
    grads = {n: p.grad.clone() for n, p in orc.named_parameters() if p.grad is not None}
    tr = Trainer(m, dtype=torch.float32, lr=1e-3)
    assert all((".lora_" in n) == p.requires_grad for n, p in m.named_parameters())
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == sum(p.numel() for p in orc.parameters() if p.requires_grad)
    tr.zero_grad()
    tr.forward_backward(unet_in, ts, ehs, ids, noisy, batch["latents"], batch["sigmas"])
    assert abs(float(tr.last_loss()) - float(loss)) / float(loss) < 1e-5
    checked = 0
    for n, p in m.named_parameters():
        if p.requires_grad:
            gr = grads.get(n)
            if gr is None or float(gr.abs().max()) == 0.0:      # attn2.to_q / to_k adapters: KV length 1 -> exactly zero
                assert float(p.grad.abs().max()) == 0.0, n
                continue
            assert float((p.grad - gr).abs().max()) <= 3e-4 * float(gr.abs().max()) + 1e-7, n
            checked += 1
    assert checked >= n_wrapped            # A and B of every projection that can receive gradient
    tr.optimizer_step()                    # AdamW on the adapters + re-pack of their 16-bit copies
    tr.zero_grad()
    tr.forward_backward(unet_in, ts, ehs, ids, noisy, batch["latents"], batch["sigmas"])
    assert float(tr.last_loss()) < float(loss)          # one lr = 1e-3 step on the same batch lowers the loss
    # packed adapter copies: at r = 64 the flat layout regroups the factors (A_q A_k A_v A_out B_q ...) so that the stacked operands
    # are views of the 16-bit twin and AdamW maintains their transposes; the padded rank keeps the re-packing path
    from svd_xtend_amd.ops import LoraOp
    ops_ = [l for kind, mod in m.steps if kind == "attn" for blk in list(mod.transformer_blocks) + list(mod.temporal_transformer_blocks)
            for l in blk.attn1.loras if isinstance(l, LoraOp)]
    assert ops_
    if r == 64:
        assert all(l.fast for l in ops_)
        assert any(l.J == 3 and l.seg_ok for l in ops_) and any(l.J == 3 and not l.seg_ok for l in ops_)     # C = 128 / C = 64 levels
        for l in ops_:
            assert l.A3.data_ptr() == tr.rt.act_view(l.mods[0].A.data).data_ptr()
            for j, mod in enumerate(l.mods):            # after the optimizer step: twins and transposes written by AdamW are current
                a16 = mod.A.data.to(tr.rt.dt)
                assert torch.equal(l.A3[j * l.rp:(j + 1) * l.rp], a16) and torch.equal(l.A3T[:, j * l.rp:(j + 1) * l.rp], a16.t())
                assert torch.equal(l.BTp[j], mod.B.data.to(tr.rt.dt).t())
        offs = sorted(zip(tr.offsets, tr.params), key=lambda t: t[0])
        assert [o for o, _ in offs] != tr.offsets          # layout order differs from the optimizer's parameter order
    else:
        assert not any(l.fast for l in ops_)


def test_to_dtype_keeps_float_masters(emu_backend):
    """`unet.to(device, dtype=torch.float16)` as the reference scripts do must not destroy the fp32 masters."""
    _, m = build_pair(4)
    m2 = m.to(torch.device("cpu"), dtype=torch.bfloat16)
    assert m2 is m and all(p.dtype == torch.float32 for p in m.parameters()) and m._requested_dtype == torch.bfloat16
    assert m.half()._requested_dtype == torch.float16
    m.to(dtype=torch.float32)
    select_trainable(m)
    m._requested_dtype = torch.float32         # the CPU emulator stores activations in the requested dtype
    m.prepare()
    assert m.rt.dt == torch.float32


def test_zero_grad_then_stale_gradients_are_overwritten(emu_backend):
    """Trainer.zero_grad() clears only the atomically-accumulated slots; the big matrices are STORED by the next backward sweep.
    Two consecutive steps on different batches must therefore give the gradients of the second batch alone -- through
    Trainer.backward() and through the autograd route (`loss.backward()`), and a second backward without zero_grad must add."""
    _, m = build_pair(6)
    tr = Trainer(m, dtype=torch.float32, lr=1e-3)
    b1, b2 = make_synthetic_batch(1, 2, 16, 16, 21, cross_dim=64), make_synthetic_batch(1, 2, 16, 16, 22, cross_dim=64)

    def args(b):
        unet_in, ts, ehs, ids, noisy, sig = edm_inputs(b)
        return (unet_in, ts, ehs, ids, noisy, b["latents"], b["sigmas"]), sig

    a1, _ = args(b1)
    a2, sig2 = args(b2)
    tr.g_flat.fill_(7.0)                      # garbage everywhere: nothing may survive zero_grad + backward
    tr.zero_grad()
    tr.forward_backward(*a2)
    ref = tr.g_flat[:tr.n_flat].clone()
    assert float(ref.abs().max()) < 7.0 and torch.isfinite(ref).all()
    tr.micro = 0
    tr.zero_grad()
    tr.forward_backward(*a1)                  # stale = gradients of batch 1
    tr.micro = 0
    tr.zero_grad()
    tr.forward_backward(*a2)
    assert torch.equal(tr.g_flat[:tr.n_flat], ref)
    tr.micro = 0
    tr.zero_grad()                            # autograd route after Trainer.zero_grad()
    loss = edm_loss(m(a2[0], a2[1], a2[2], added_time_ids=a2[3]).sample, a2[4], b2["latents"], sig2)
    loss.backward()
    assert torch.allclose(tr.g_flat[:tr.n_flat], ref, rtol=1e-5, atol=1e-7)
    tr.forward_backward(*a2)                  # no zero_grad: accumulates
    tr.micro = 0
    assert torch.allclose(tr.g_flat[:tr.n_flat], 2 * ref, rtol=1e-5, atol=1e-7)


def test_gemm_config_rules_for_the_c2_shapes():
    """ops.choose_cfg: the cost model picks what the in-situ sweeps of round 3 measured as best on the shapes it
    was fitted to, and never a split that leaves a slice without K-tiles."""
    from svd_xtend_amd.ops import TILE_OF_VARIANT, choose_cfg

    class RT:
        gemm_variant, split_k = 4, True
    rt = RT()
    assert choose_cfg(rt, 35840, 320, 2880, 320, 320) == (1, 6)          # 64x40 level: 448 two-stage 160 x 160 tiles, two per CU
    assert choose_cfg(rt, 35840, 320, 320, 320) == (1, 6)
    assert choose_cfg(rt, 8960, 640, 5760, 640, 640) == (1, 22)          # 32x20 level, N = 640: 47 x 5 = 235 eight-wave 192 x 128 ring tiles on 256 CUs
    assert choose_cfg(rt, 8960, 640, 2560, 640) == (1, 22)               #   (256-row tiles: 175)
    assert choose_cfg(rt, 8960, 1280, 5760, 1280, 640) == (1, 6)         # N = 1280: 448 two-stage 160 x 160 tiles, two per CU
    assert choose_cfg(rt, 8960, 1920, 640, 1920) == (1, 7)               # q/k/v projection: 840 two-stage 128 x 160 tiles
    assert choose_cfg(rt, 2240, 1280, 11520, 1280, 1280) == (2, 22)      # 16x10 level, long K: 120 tiles of 192 x 128 x 2 slices
    assert choose_cfg(rt, 2240, 1280, 3840, 1280, 1280) == (1, 24)       # medium / short K: 24 x 10 = 240 four-wave ring tiles of 96 x 128, no split
    assert choose_cfg(rt, 2240, 1280, 1280, 1280) == (1, 24)             #   (128-row tiles: 180)
    assert choose_cfg(rt, 560, 1280, 3840, 1280, 1280) == (3, 24)        # 8x5 level: 6 x 10 tiles of 96 x 128, 3 slices
    assert choose_cfg(rt, 2457600, 128, 1152, 128, 128) == (1, 27)       # conditioner widths (VAE): round 6's in-situ table, not the UNet-fitted model
    assert choose_cfg(rt, 614400, 256, 2304, 256, 256) == (1, 18) and choose_cfg(rt, 38400, 512, 4608, 512, 512) == (1, 27)
    s, v = choose_cfg(rt, 560, 1280, 11520, 1280, 1280)
    assert v in (22, 23) and 6 <= s <= 10
    for (M, N, Kd) in [(560, 1280, 11520), (560, 1280, 1280), (2240, 640, 5760), (8960, 1920, 640), (300, 960, 192), (64, 640, 1280), (1000, 4, 576)]:
        s, v = choose_cfg(rt, M, N, Kd, N)
        kt = Kd // 64
        assert v in TILE_OF_VARIANT and s >= 1 and (s == 1 or -(-kt // s) * (s - 1) < kt), (M, N, Kd, s, v)    # every split owns at least one K-tile
    rt.gemm_variant = 1
    assert choose_cfg(rt, 2240, 1280, 11520, 1280)[1] == 1


def test_split_rules_never_leave_an_empty_slice():
    """svdx_gemm / svdx_gemm_tn refuse a slab split whose last slices own no K-tiles / rows (their slabs would stay unwritten and the
    reducing launch would add uninitialised memory): the host rules must never ask for one, on any problem of configs 2, 4 and 5."""
    from svd_xtend_amd.ops import _tn_formula, choose_cfg

    class RT:
        gemm_variant, split_k = 4, True
    rt = RT()
    rows = [35840, 8960, 2240, 560, 230400, 57600, 14400, 3600, 6144, 1536, 384, 96]
    widths = [64, 128, 192, 256, 320, 640, 960, 1280, 1920, 2560, 3840, 5120, 10240, 2880, 5760, 8640, 11520, 17280, 23040]
    for M in rows:
        for N in widths:
            for Kd in widths:
                sk, _ = _tn_formula(M, N, Kd)
                r = -(-M // 64)
                assert sk == 1 or -(-r // sk) * (sk - 1) < r, ("tn", M, N, Kd, sk)
                if N <= 10240:
                    s, v = choose_cfg(rt, M, N, Kd, N)
                    kt = Kd // 64
                    assert s == 1 or -(-kt // s) * (s - 1) < kt, ("nt", M, N, Kd, s, v)


def test_gemm_tuner_picks_fastest_candidate_per_problem():
    from svd_xtend_amd.ops import GemmTuner
    tn = GemmTuner(rounds=2)

    class Ev:
        def __init__(self, t):
            self.t = t

        def elapsed_time(self, other):
            return other.t - self.t

        def synchronize(self):
            pass
    cost = {"a": {1: 5.0, 2: 3.0, 3: 4.0}, "b": {10: 1.0, 20: 2.0}}
    steps = 0
    while tn.active:
        for key, cands in cost.items():
            cfg, idx = tn.pick(key, lambda c=cands: list(c))
            tn.record(key, idx, Ev(0.0), Ev(cands[cfg]))
        steps += 1
        tn.end_step()
        assert steps < 20
    assert tn.table == {"a": 2, "b": 10} and steps == 2 * 3


def test_lora_adapter_state_dict_round_trip():
    """Adapter weights under the reference's on-disk names (train_svd_lora.py:1065-1074: `unet.<module>.lora_A.weight`)."""
    from svd_xtend_amd.lora import LoraConfig, load_lora_state_dict, lora_state_dict
    _, m = build_pair(8)
    n = m.add_adapter(LoraConfig(r=4, lora_alpha=4, init_lora_weights="gaussian"))
    sd = lora_state_dict(m)
    assert len(sd) == 2 * n and all(k.startswith("unet.") and (".lora_A.weight" in k or ".lora_B.weight" in k) for k in sd)
    k0 = "unet.down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q.lora_A.weight"
    assert k0 in sd and sd[k0].shape == (4, 64)
    _, m2 = build_pair(8)
    m2.add_adapter(LoraConfig(r=4, lora_alpha=4, init_lora_weights="gaussian"))
    load_lora_state_dict(m2, {k: v.clone() for k, v in sd.items()})
    sd2 = lora_state_dict(m2)
    assert all(torch.equal(sd[k], sd2[k]) for k in sd)
    with pytest.raises(KeyError):
        load_lora_state_dict(m2, {"unet.nope.lora_A.weight": torch.zeros(1)})


def test_geglu_tile_rule():
    """ops.choose_geglu_variant: the two-per-CU eight-wave tile under every GEGLU epilogue with M >= 1024 except the 32x20-level forward
    (256 x 256), ring tiles below; the rule's answer is always among the candidates the in-situ tuner sweeps."""
    from svd_xtend_amd.ops import GEGLU_TWO_PER_CU, choose_geglu_variant, geglu_candidates
    for (M, C) in [(35840, 320), (8960, 640), (2240, 1280), (560, 1280), (230400, 320), (57600, 640), (14400, 1280), (3600, 1280)]:
        F = 4 * C
        f, b = choose_geglu_variant(M, 2 * F, C), choose_geglu_variant(M, F, C, fwd=False)
        assert f in geglu_candidates(M, 2 * F, C) and b in geglu_candidates(M, F, C, fwd=False), (M, C, f, b)
        if M < 1024:
            assert GEGLU_TWO_PER_CU not in (f, b)
        else:
            assert b == GEGLU_TWO_PER_CU and f == (18 if 4096 <= M < 16384 else GEGLU_TWO_PER_CU)


def test_tile_table_matches_the_kernel_dispatch():
    """ops.TILE_OF_VARIANT (what the cost model believes a variant's tile is) against the template arguments csrc/gemm.hip dispatches that
    variant to: rows = 16 * MB * WGM, columns = 32 * NB, stages = NSTG, waves = 2 * WGM."""
    import os
    import re
    from svd_xtend_amd.ops import GEGLU_TWO_PER_CU, STAGED_TILES, TILE_OF_VARIANT
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "svd_xtend_amd", "csrc", "gemm.hip")).read()
    seen = {}
    for line in src.splitlines():
        m = re.match(r"\s*((?:case \d+: )+)(.*)", line)
        if not m or "launch_gemm_v4<" not in m.group(2):
            continue
        variants = [int(v) for v in re.findall(r"case (\d+):", m.group(1))]
        tiles = [tuple(int(x) for x in t) for t in re.findall(r"launch_gemm_v4<T, (\d+), (\d+), (\d+), (\d+)>", m.group(2))]
        assert tiles, line
        for v in variants:
            seen[v] = tiles
    # the two-role tiles: `if (variant == V && ...) return launch_gemm_v5<T, MF, NF>(p, st);` -> rows = 32 * MF, columns = 64 * NF, two K-tiles, eight waves
    for v, mf, nf in re.findall(r"if \(variant == (\d+) &&[^\n]*launch_gemm_v5<T, (\d+), (\d+)>", src):
        seen[int(v)] = [("v5", int(mf), int(nf))]
    # variant 36: `if (variant == 36) { if (...) return launch_gemm_v4<T, NB, MB, WGM, NSTG, MSTEP>(p, st);`: the table holds the row STEP, the tile computes 16 MB WGM rows
    m36 = re.search(r"if \(variant == 36\) \{\s*if \([^\n]*launch_gemm_v4<T, (\d+), (\d+), (\d+), (\d+), (\d+)>", src)
    assert m36, "variant 36 dispatch not found"
    nb, mb, wgm, nstg, mstep = (int(x) for x in m36.groups())
    assert STAGED_TILES[36] == (mstep, 32 * nb, nstg, 2 * wgm) and mstep <= 16 * mb * wgm == 144
    assert not set(STAGED_TILES) & set(TILE_OF_VARIANT)
    for v, (bm, bn, stages, waves) in list(TILE_OF_VARIANT.items()) + [kv for kv in STAGED_TILES.items() if kv[0] != 36]:
        if v < 16:
            continue                                   # 6 / 7 / 8: the two-stage four-wave defaults of launch_gemm_v4<T, NB, MB>
        if seen[v][0][0] == "v5":
            geo = {(32 * mf, 64 * nf, 2, 8) for _, mf, nf in seen[v]}
        else:
            geo = {(16 * mb * wgm, 32 * nb, nstg, 2 * wgm) for nb, mb, wgm, nstg in seen[v]}
        assert (bm, bn, stages, waves) in geo, (v, (bm, bn, stages, waves), geo)
    assert {(16 * mb * wgm, 32 * nb, nstg, 2 * wgm) for nb, mb, wgm, nstg in seen[GEGLU_TWO_PER_CU]} == {(192, 128, 2, 8)}


def test_attention_processor_plumbing_and_forward_chunking():
    """The processor get / set surface and enable_forward_chunking of the reference class
    (/root/reference/src/unet_spatio_temporal_condition.py:248-321, 328-355): same keys ("<attention module path>.processor", one per
    attention layer: 16 transformers x (spatial + temporal) x (attn1 + attn2) = 64), same argument checks; weights are untouched."""
    from oracle.unet import TINY_CONFIG
    from svd_xtend_amd.unet import HipAttnProcessor, UNetSpatioTemporalConditionModel
    m = UNetSpatioTemporalConditionModel(**TINY_CONFIG)
    keys_before = list(m.state_dict())
    procs = m.attn_processors
    assert len(procs) == 64 and all(k.endswith(".processor") for k in procs)
    attn_modules = {n for n, _ in m.named_modules() if n.endswith(("attn1", "attn2"))}
    assert {k[:-len(".processor")] for k in procs} == attn_modules
    assert "down_blocks.0.attentions.0.temporal_transformer_blocks.0.attn2.processor" in procs and "mid_block.attentions.0.transformer_blocks.0.attn1.processor" in procs
    one = HipAttnProcessor()
    m.set_attn_processor(one)
    assert all(p is one for p in m.attn_processors.values())
    m.set_attn_processor({k: HipAttnProcessor() for k in procs})          # a dict keyed like attn_processors
    assert len({id(p) for p in m.attn_processors.values()}) == 64
    with pytest.raises(ValueError, match="does not match"):
        m.set_attn_processor({"x.processor": one})
    with pytest.raises(ValueError):
        m.set_attn_processor(object())                                    # no foreign attention implementations
    m.set_default_attn_processor()
    m.enable_forward_chunking()                                            # default chunk size 1 over the batch dimension
    blocks = [b for b in m.modules() if hasattr(b, "set_chunk_feed_forward")]
    assert len(blocks) == 32 and all((b._chunk_size, b._chunk_dim) == (1, 0) for b in blocks)
    m.enable_forward_chunking(4, dim=1)
    assert all((b._chunk_size, b._chunk_dim) == (4, 1) for b in blocks)
    with pytest.raises(ValueError, match="either 0 or 1"):
        m.enable_forward_chunking(dim=2)
    assert list(m.state_dict()) == keys_before


def test_deferred_skinny_gradients_are_final_at_the_block_hook(emu_backend):
    """Runtime.flush_deferred: the skinny gradient launches of a transformer block (cross-attention value path, LayerNorm affine
    reductions) are queued; when the per-block hook consumes the block's gradients (gradient buckets, graph cuts) they run before
    it, otherwise they wait for the end of the sweep -- same gradients either way."""
    orc, m = build_pair(3)
    batch = make_synthetic_batch(1, 2, 16, 16, 9, cross_dim=64)
    unet_in, ts, ehs, ids, noisy, _ = edm_inputs(batch)
    tr = Trainer(m, dtype=torch.float32, lr=1e-3)
    assert tr.rt.batch_small
    args = (unet_in, ts, ehs, ids, noisy, batch["latents"], batch["sigmas"])
    tr.zero_grad()
    tr.forward_backward(*args)
    final = tr.g_flat.clone()
    watched = [n for n, p in m.named_parameters() if p.requires_grad and ("attn2.to_v" in n or "attn2.to_out" in n or ".norm1." in n)]
    params = dict(m.named_parameters())
    off = {id(p): o for p, o in zip(tr.params, tr.offsets)}
    seen = {}

    def hook(flushed):
        def cb(module):
            ids_ = {id(p) for p in module.parameters()}
            for n in watched:
                p = params[n]
                if id(p) in ids_:
                    seen[(flushed, n)] = torch.equal(p.grad.reshape(-1), final[off[id(p)]:off[id(p)] + p.numel()])
        return cb
    for flushed in (True, False):
        tr.zero_grad()
        tr.micro = 0
        tr.forward_loss(*args)
        tr.backward(on_block=hook(flushed), block_grads_final=flushed)
        assert torch.equal(tr.g_flat, final)
        assert not tr.rt.deferred_pending
    assert watched and all(seen[(True, n)] for n in watched)                       # final when the hook says it reads them
    assert not all(seen[(False, n)] for n in watched)                              # still queued otherwise


def test_three_step_trajectory_matches_oracle(emu_backend):
    """e2e_checks.trajectory_vs_oracle on the emulated kernels at fp32 storage: three consecutive optimizer steps reproduce the oracle's
    losses and every tensor's accumulated update (the GPU form of this test runs in fp16, tests/test_e2e_gpu.py)."""
    r = e2e_checks.trajectory_vs_oracle(TINY_CONFIG, (1, 3, 16, 16), dtype=torch.float32, steps=3, lr=1e-3, dev=CPU)
    assert r["opt_steps"] == 3.0 and max(r["loss_rel"]) < 2e-5, r
    assert r["update_cos_min"] > 0.999 and 0.99 < r["update_norm_ratio_min"] <= r["update_norm_ratio_max"] < 1.01, r


def test_trajectory_check_sees_a_zeroed_gradient(emu_backend, monkeypatch):
    """The mutation round 4 lived with: the skinny gradient chain of the temporal blocks' cross-attention value path reads a
    zeroed vector.  First-step loss parity cannot see it; the trajectory check must."""
    from svd_xtend_amd import ops
    real = ops.Runtime.flush_deferred

    def broken(self, *a, **kw):
        self._q_outer = []                              # drop the queued outer products (dW of attn2.to_v / to_out) instead of running them
        return real(self, *a, **kw)
    monkeypatch.setattr(ops.Runtime, "flush_deferred", broken)
    r = e2e_checks.trajectory_vs_oracle(TINY_CONFIG, (1, 3, 16, 16), dtype=torch.float32, steps=3, lr=1e-3, dev=CPU)
    assert max(r["loss_rel"][:1]) < 2e-5, r            # the first loss is blind to it ...
    assert r["update_cos_min"] < 0.9, r                # ... the updates are not


def test_folded_inf_check_equals_the_full_pass(emu_backend):
    """GradScaler's inf check where the gradients are written (Runtime.fold_finite: svdx_gemm_tn / svdx_grad_finalize_batch raise
    opt_state[3], svdx_check_finite_spans covers the accumulated slots) against the 1.59 GB pass of svdx_check_finite: a loss scale that
    overflows fp16 in the backward sweep must skip the step and halve the scale in both forms, a sane one must step in both, and the two
    trajectories must be the same bits."""
    batch = make_synthetic_batch(1, 3, 16, 16, 7, cross_dim=64)
    unet_in, ts, ehs, ids, noisy, _ = edm_inputs(batch)
    b = dict(unet_in=unet_in, timesteps=ts, ehs=ehs, added_time_ids=ids, noisy_latents=noisy, target=batch["latents"], sigmas=batch["sigmas"])
    out = {}
    for fold in (True, False):
        _, m = build_pair(4)
        tr = Trainer(m, dtype=torch.float16, lr=1e-3, init_scale=2.0 ** 30)
        tr.rt.fold_finite = fold
        assert tr.rt.found_inf is not None and tr.finite_spans is not None
        k, names = tr.rt.k, []
        for nm in ("check_finite", "check_finite_spans"):
            setattr(k, nm, (lambda f, nm: lambda *a, **kw: (names.append("svdx_" + nm), f(*a, **kw))[1])(getattr(k, nm), nm))
        states = []
        for _ in range(8):                      # 2^30 overflows; the scale halves until the sweep is finite, then the steps are taken
            tr.step(b)
            states.append((float(tr.opt_state[0]), float(tr.opt_state[1])))
        for nm in ("check_finite", "check_finite_spans"):
            k.__dict__.pop(nm, None)
        assert ("svdx_check_finite_spans" in names) == fold and ("svdx_check_finite" in names) == (not fold)
        out[fold] = (states, tr.p_flat.clone())
    assert out[True][0] == out[False][0], (out[True][0], out[False][0])
    assert out[True][0][0][0] == 0.0 and out[True][0][0][1] == 2.0 ** 29          # the first step was skipped, the scale halved
    assert out[True][0][-1][0] >= 1.0                                            # ... and steps were taken once the sweep was finite
    assert torch.equal(out[True][1], out[False][1])


def test_reference_dtype_adamw_is_torch_adamw_on_bf16_tensors():
    """Trainer(lora_param_dtype="reference") / svdx_adamw* param_mode 1: the reference's LoRA recipe under --mixed_precision bf16 keeps adapters,
    gradients and optimizer state as bf16 tensors (/root/reference/train_svd_lora.py:666-674, torch.optim.AdamW at :766-772).  The emulation of
    that mode (which the GPU kernel is held to in kernel_checks.check_optim) against torch.optim.AdamW ITSELF stepping bf16 CPU tensors:
    parameters and both moments bit for bit over five steps."""
    import emul
    be = emul.EmuBackend()
    g = torch.Generator().manual_seed(3)
    n = 4096
    p0 = (torch.randn(n, generator=g) * 0.05).to(torch.bfloat16)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    p, m, v = p0.float().clone(), torch.zeros(n), torch.zeros(n)
    pa = torch.zeros(n, dtype=torch.bfloat16)
    st = torch.tensor([0, 1.0, 0, 0, 1, 1, 1, 0, 1.0] + [0.0] * 27)
    for step in range(5):
        grad = torch.randn(n, generator=g) * 0.01
        ref_p.grad = grad.to(torch.bfloat16)                 # the reference's gradient is a bf16 tensor
        opt.step()
        be.optim_prep(st, 0.9, 0.999, 2.0, 0.5, 2000, 0)
        be.adamw(p, grad, m, v, n, 1e-3, 0.9, 0.999, 1e-8, 1e-2, 1.0, st, pa, param_mode=1)
        state = opt.state[ref_p]
        assert torch.equal(p, ref_p.detach().float()), (step, float((p - ref_p.detach().float()).abs().max()))
        assert torch.equal(m, state["exp_avg"].float()) and torch.equal(v, state["exp_avg_sq"].float()), step
        assert torch.equal(pa, ref_p.detach())
    # and the fp32-master default is NOT that trajectory (the deviation DESIGN documents): most 1e-3 steps round away on a bf16 parameter
    assert float((p - p0.float()).abs().max()) > 0


def test_trainer_reference_lora_dtype_keeps_bf16_parameters_and_state(emu_backend):
    """Trainer(lora_param_dtype="reference"): the adapters start as bf16 numbers (the reference creates them in a bf16 UNet), every optimizer
    step leaves parameters and both moments bf16-valued, the trajectory differs from the fp32-master default, and each step IS
    torch.optim.AdamW on bf16 tensors fed the bf16-rounded gradient of that step (the op-level pin is the test above)."""
    from svd_xtend_amd.lora import LoraConfig
    batch = make_synthetic_batch(1, 3, 16, 16, 11, cross_dim=64)
    unet_in, ts, ehs, ids, noisy, _ = edm_inputs(batch)
    b = dict(unet_in=unet_in, timesteps=ts, ehs=ehs, added_time_ids=ids, noisy_latents=noisy, target=batch["latents"], sigmas=batch["sigmas"])

    def make(mode):
        _, m = build_pair(3)
        for p in m.parameters():
            p.requires_grad_(False)
        torch.manual_seed(5)
        m.add_adapter(LoraConfig(r=8, lora_alpha=8, init_lora_weights="gaussian"))
        g = torch.Generator().manual_seed(0)
        for n, p in m.named_parameters():
            if ".lora_B." in n:
                p.data.copy_(torch.randn(p.shape, generator=g) * 0.05)
        return Trainer(m, dtype=torch.bfloat16, lr=1e-3, lora_param_dtype=mode)

    with pytest.raises(ValueError):
        _, m0 = build_pair(3)
        Trainer(m0, dtype=torch.float16, lora_param_dtype="reference")
    ref, dflt = make("reference"), make(None)
    is_bf16 = lambda t: torch.equal(t, t.to(torch.bfloat16).float())      # noqa: E731
    n = ref.n_flat
    assert is_bf16(ref.p_flat[:n]) and not is_bf16(dflt.p_flat[:n])
    shadow = torch.nn.Parameter(ref.p_flat[:n].to(torch.bfloat16).clone())
    opt = torch.optim.AdamW([shadow], lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    for _ in range(3):
        ref.zero_grad()
        ref.forward_backward(**b)
        shadow.grad = ref.g_flat[:n].to(torch.bfloat16)                 # bf16 run: no loss scale
        ref.optimizer_step()
        opt.step()
        dflt.step(b)
        assert is_bf16(ref.p_flat[:n]) and is_bf16(ref.m_flat[:n]) and is_bf16(ref.v_flat[:n])
        assert torch.equal(ref.p_flat[:n], shadow.detach().float())
    assert float((ref.p_flat[:n] - dflt.p_flat[:n]).abs().max()) > 0


def test_big_reference_fingerprint_is_enforced():
    """tests/golden/big_ref_fingerprints.json (config 4's 9216-pixel level: loss, seeded-weight fingerprint, prediction norm, 104 gradient
    norms) is what a cached or recomputed big oracle reference must reproduce: e2e_checks.check_big_ref_fingerprint accepts a reference
    that carries these numbers and refuses one whose loss or one gradient moved."""
    import json
    import os
    import pytest
    import torch
    import e2e_checks
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_ref_fingerprints.json")))
    tag = "L0 320ch 72x128 T=25 seed=11"
    w = want[tag.replace(" ", "_")]
    assert w["n_grads"] == 104 and len(w["grad_l2"]) == 104 and abs(w["loss"] - 0.769970178604126) < 1e-12
    unit = lambda n: torch.tensor([float(n)])                                            # noqa: E731  (a tensor whose L2 norm is n)
    ref = dict(loss=w["loss"], sd0_fingerprint=w["sd0_fingerprint"], pred=unit(w["pred_l2"]), grads={k: unit(v) for k, v in w["grad_l2"].items()})
    assert e2e_checks.check_big_ref_fingerprint(tag, ref)
    assert not e2e_checks.check_big_ref_fingerprint("no such case", ref)
    with pytest.raises(AssertionError):
        e2e_checks.check_big_ref_fingerprint(tag, dict(ref, loss=w["loss"] * 1.001))
    k0 = next(iter(w["grad_l2"]))
    with pytest.raises(AssertionError):
        e2e_checks.check_big_ref_fingerprint(tag, dict(ref, grads=dict(ref["grads"], **{k0: unit(w["grad_l2"][k0] * 1.01)})))
