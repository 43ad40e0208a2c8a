"""Rows "next" of SURVEY.md 8(f), ranks 3-4: checkpoint-N resume, the learning-rate schedule and the EMA of the weights, checked
on CPU against oracle/schedule.py (kernels replaced by tests/emul.py).  The schedule oracle itself is pinned here against
transformers.optimization, the module diffusers.optimization was derived from."""
import math
import os

import pytest
import torch

from oracle.schedule import SCHEDULES, lr_lambda, lr_trajectory
from oracle.schedule import EMAModel as OracleEMA
from oracle.step import edm_inputs, make_synthetic_batch
from oracle.unet import TINY_CONFIG, UNetSpatioTemporalConditionOracle, scaled_init_
from svd_xtend_amd import checkpoint as ckpt
from svd_xtend_amd.optimization import get_scheduler
from svd_xtend_amd.train import Trainer
from svd_xtend_amd.training_utils import EMAModel
from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel


def build(seed=0):
    orc = UNetSpatioTemporalConditionOracle(**TINY_CONFIG)
    scaled_init_(orc, seed)
    m = UNetSpatioTemporalConditionModel(**TINY_CONFIG)
    m.load_state_dict(orc.state_dict(), strict=True)
    return m


def batch_of(seed, B=1, T=2, h=16, w=16):
    b = make_synthetic_batch(B, T, h, w, seed, cross_dim=64)
    unet_in, ts, ehs, ids, noisy, _ = edm_inputs(b)
    return dict(unet_in=unet_in, timesteps=ts, ehs=ehs, added_time_ids=ids, noisy_latents=noisy, target=b["latents"], sigmas=b["sigmas"])


# ---- the oracle's schedules against transformers (installed) -------------------------------------------------------------------
def test_schedule_oracle_is_pinned_to_transformers():
    import transformers.optimization as TO
    p = torch.nn.Parameter(torch.zeros(1))

    def traj(fn, n, **kw):
        opt = torch.optim.SGD([p], lr=1.0)
        sched, out = fn(opt, **kw), []
        for _ in range(n):
            out.append(opt.param_groups[0]["lr"])
            opt.step()
            sched.step()
        return out
    cases = [("constant_with_warmup", TO.get_constant_schedule_with_warmup, dict(num_warmup_steps=5)),
             ("linear", TO.get_linear_schedule_with_warmup, dict(num_warmup_steps=3, num_training_steps=20)),
             ("cosine", TO.get_cosine_schedule_with_warmup, dict(num_warmup_steps=3, num_training_steps=20)),
             ("cosine", TO.get_cosine_schedule_with_warmup, dict(num_warmup_steps=0, num_training_steps=17, num_cycles=1.5)),
             ("cosine_with_restarts", TO.get_cosine_with_hard_restarts_schedule_with_warmup,
              dict(num_warmup_steps=3, num_training_steps=20, num_cycles=3)),
             ("polynomial", TO.get_polynomial_decay_schedule_with_warmup, dict(num_warmup_steps=3, num_training_steps=20, power=2.0)),
             ("polynomial", TO.get_polynomial_decay_schedule_with_warmup, dict(num_warmup_steps=0, num_training_steps=9))]
    for name, fn, kw in cases:
        ref = traj(fn, 26, **kw)
        mine = [lr_lambda(name, i, **kw) for i in range(26)]
        assert max(abs(a - b) for a, b in zip(ref, mine)) <= 1e-12, name
    assert all(lr_lambda("constant", i) == 1.0 for i in range(5))


# ---- device-side schedule: the lr every optimizer step runs at ---------------------------------------------------------------------
@pytest.mark.parametrize("name", SCHEDULES)
def test_lr_schedule_on_device_follows_oracle(emu_backend, name):
    m = build(0)
    tr = Trainer(m, dtype=torch.float16, lr=1e-3, init_scale=256.0)
    kw = dict(num_warmup_steps=2 * 3, num_training_steps=9 * 3)      # x num_processes, train_svd.py:810-812
    if name == "cosine_with_restarts":
        kw["num_cycles"] = 2
    if name == "polynomial":
        kw["power"] = 2.0
    sched = get_scheduler(name, optimizer=tr, steps_per_step=3, **kw)
    okw = {k: v for k, v in kw.items()}
    if name == "polynomial":
        okw.update(lr_init=1e-3, lr_end=1e-7)
    skipped = {4}
    want = lr_trajectory(name, 1e-3, 12, num_processes=3, skipped=skipped, **okw)
    k = tr.rt.k
    for i in range(12):
        assert abs(sched.get_last_lr()[0] - want[i]) <= 1e-9, (name, i)        # lr the coming step will use
        tr.g_flat.zero_()
        if i in skipped:
            tr.g_flat[3] = float("inf")
        k.check_finite(tr.g_flat, tr.n_flat, tr.opt_state)
        k.optim_prep(tr.opt_state, 0.9, 0.999, 2.0, 0.5, 2000, 1)
        assert float(tr.opt_state[7]) == (1.0 if i in skipped else 0.0)
        if i not in skipped:
            assert abs(1e-3 * float(tr.opt_state[8]) - want[i]) <= 1e-9 + 2e-6 * want[i], (name, i)
        sched.step()
    assert sched.last_epoch == 11 * 3
    sd = sched.state_dict()
    assert sd["last_epoch"] == 33 and sd["base_lrs"] == [1e-3]
    sched.load_state_dict(sd)
    with pytest.raises(ValueError):
        sched.load_state_dict({**sd, "last_epoch": 5})


def test_get_scheduler_argument_errors(emu_backend):
    tr = Trainer(build(0), dtype=torch.float32, lr=1e-3)
    with pytest.raises(ValueError, match="requires `num_warmup_steps`"):
        get_scheduler("linear", optimizer=tr, num_training_steps=10)
    with pytest.raises(ValueError, match="requires `num_training_steps`"):
        get_scheduler("cosine", optimizer=tr, num_warmup_steps=1)
    with pytest.raises(ValueError):
        get_scheduler("no_such_schedule", optimizer=tr)
    with pytest.raises(ValueError, match="requires `step_rules`"):
        get_scheduler("piecewise_constant", optimizer=tr)
    with pytest.raises(ValueError, match="at most 8"):
        get_scheduler("piecewise_constant", optimizer=tr, step_rules=",".join(f"1:{i}" for i in range(1, 11)) + ",0.5")
    with pytest.raises(TypeError):
        get_scheduler("constant", optimizer=torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=0.1))
    s = get_scheduler("constant", optimizer=tr)                 # the reference's default (--lr_scheduler constant)
    assert s.get_last_lr() == [1e-3] and float(tr.opt_state[9]) == 0.0


def test_piecewise_constant_schedule_on_device(emu_backend):
    """diffusers' get_piecewise_constant_schedule (train_svd.py:807-812 passes --lr_scheduler through): the rule string of its docstring,
    evaluated by svdx_optim_prep from the rules stored behind the 16 state floats, against the rule restated here -- with num_processes
    scheduler steps per optimizer step and a skipped step that must not advance it."""
    tr = Trainer(build(0), dtype=torch.float16, lr=1e-3, init_scale=256.0)
    sched = get_scheduler("piecewise_constant", optimizer=tr, step_rules="1:10,0.1:20,0.01:30,0.005", steps_per_step=3)

    def rule(n):
        return 1.0 if n < 10 else 0.1 if n < 20 else 0.01 if n < 30 else 0.005
    k, n = tr.rt.k, 0
    for i in range(14):
        skip = i == 5
        assert abs(sched.get_last_lr()[0] - 1e-3 * rule(n)) <= 1e-12, i
        tr.g_flat.zero_()
        if skip:
            tr.g_flat[3] = float("inf")
        k.check_finite(tr.g_flat, tr.n_flat, tr.opt_state)
        k.optim_prep(tr.opt_state, 0.9, 0.999, 2.0, 0.5, 2000, 1)
        if not skip:
            assert abs(float(tr.opt_state[8]) - rule(n)) <= 1e-7, (i, n)
            n += 3
        sched.step()
    assert sched.last_epoch == 13 * 3


def test_scheduled_step_moves_weights_by_scheduled_lr(emu_backend):
    """First step of a warmup runs at lr * lambda(0) = 0: weights must not move; the second step moves them."""
    m = build(1)
    tr = Trainer(m, dtype=torch.float32, lr=1e-2, weight_decay=0.0)
    get_scheduler("constant_with_warmup", optimizer=tr, num_warmup_steps=2)
    b = batch_of(3)
    before = tr.p_flat.clone()
    tr.step(b)
    assert torch.equal(tr.p_flat, before) and float(tr.opt_state[0]) == 1.0
    tr.step(b)
    d = (tr.p_flat - before)[:tr.n_flat].abs().max()
    assert 0 < float(d) <= 0.5e-2 * 1.001 * 3.2          # Adam step 2 at half the base lr (|m_hat / sqrt(v_hat)| <= ~3.2 at step 2)


# ---- EMA ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kw", [dict(), dict(decay=0.9, use_ema_warmup=True, inv_gamma=1.0, power=0.75), dict(update_after_step=2, min_decay=0.3)])
def test_ema_follows_oracle(emu_backend, kw):
    m = build(2)
    ema = EMAModel(m.parameters(), model_cls=UNetSpatioTemporalConditionModel, model_config=m.config, **kw)   # before the trainable set exists
    tr = Trainer(m, dtype=torch.float32, lr=1e-2)
    ema.on_weights_changed = tr.weights_changed
    params = list(m.parameters())
    ref = OracleEMA([p.detach().clone().requires_grad_(p.requires_grad) for p in params], **kw)
    b = batch_of(5)
    for i in range(5):
        tr.step(b)
        ema.step(m.parameters())
        ref.step(params)
        assert ema.cur_decay_value == ref.cur_decay_value
    assert len(ema._spans) == 1 and ema._spans[0][2] >= tr.n_flat - 64         # one launch over the flat master buffer
    for s, r, p in zip(ema.shadow_params, ref.shadow_params, params):
        assert float((s - r).abs().max()) <= 1e-6 * float(r.abs().max()) + 1e-9
    assert any(float((s - p.data).abs().max()) > 0 for s, p in zip(ema.shadow_params, params) if p.requires_grad)
    # validation swap (train_svd.py:1101-1104, 1152-1154): store, copy_to, restore; the kernels' 16-bit copies follow
    live = tr.p_flat.clone()
    w16 = tr.rt.w16_flat.clone()
    ema.store(m.parameters())
    ema.copy_to(m.parameters())
    assert not torch.equal(tr.p_flat, live) and not torch.equal(tr.rt.w16_flat, w16)
    for s, p in zip(ema.shadow_params, params):
        assert torch.equal(s, p.data)
    ema.restore(m.parameters())
    assert torch.equal(tr.p_flat, live) and torch.equal(tr.rt.w16_flat, w16)
    with pytest.raises(RuntimeError):
        ema.restore(m.parameters())
    sd = ema.state_dict()
    assert sd["optimization_step"] == 5 and len(sd["shadow_params"]) == len(params)


def test_ema_save_and_from_pretrained(emu_backend, tmp_path):
    m = build(3)
    ema = EMAModel(m.parameters(), decay=0.95, model_cls=UNetSpatioTemporalConditionModel, model_config=m.config)
    tr = Trainer(m, dtype=torch.float32, lr=1e-2)
    b = batch_of(6)
    for _ in range(3):
        tr.step(b)
        ema.step(m.parameters())
    ema.save_pretrained(str(tmp_path / "unet_ema"))
    back = EMAModel.from_pretrained(str(tmp_path / "unet_ema"), UNetSpatioTemporalConditionModel)
    assert back.decay == 0.95 and back.optimization_step == 3
    for a, b_ in zip(back.shadow_params, ema.shadow_params):
        assert torch.equal(a.cpu(), b_.cpu())
    # the folder is a loadable model whose weights are the averages
    avg = UNetSpatioTemporalConditionModel.from_pretrained(str(tmp_path / "unet_ema"))
    for (n, p), s in zip(avg.named_parameters(), ema.shadow_params):
        assert torch.equal(p.data, s.cpu()), n
    ema2 = EMAModel(m.parameters(), model_cls=UNetSpatioTemporalConditionModel, model_config=m.config)
    ema2.load_state_dict(back.state_dict())                       # load_model_hook, train_svd.py:709-714
    assert ema2.decay == 0.95 and ema2.optimization_step == 3
    for a, b_ in zip(ema2.shadow_params, ema.shadow_params):
        assert torch.equal(a, b_)
    with pytest.raises(ValueError):
        ema2.load_state_dict({"decay": 1.5})


# ---- checkpoint-N ------------------------------------------------------------------------------------------------------------------
def test_checkpoint_directory_rules(tmp_path):
    out = str(tmp_path)
    assert ckpt.latest_checkpoint(out) is None
    for n in (500, 1000, 1500, 10000):
        os.makedirs(os.path.join(out, f"checkpoint-{n}"))
    os.makedirs(os.path.join(out, "logs"))
    assert ckpt.latest_checkpoint(out) == "checkpoint-10000"                 # numeric, not lexicographic (train_svd.py:907)
    assert ckpt.latest_checkpoint(out, "some/where/checkpoint-500") == "checkpoint-500"
    assert ckpt.global_step_of("checkpoint-1500") == 1500
    assert ckpt.rotate_checkpoints(out, None) == [] and ckpt.rotate_checkpoints(out, 5) == []
    assert ckpt.rotate_checkpoints(out, 3) == ["checkpoint-500", "checkpoint-1000"]   # leaves limit - 1 before the new save
    assert sorted(os.listdir(out)) == ["checkpoint-10000", "checkpoint-1500", "logs"]


def test_checkpoint_rules_match_reference_statements(tmp_path):
    """Decisions recorded from the reference's own resume (:901-925) and rotation (:1062-1090) statements
    (tests/golden/make_golden_checkpoint_rules.py) on six directory layouts."""
    import json
    rules = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoint_rules.json")))
    assert len(rules) == 6
    for i, r in enumerate(rules):
        root = str(tmp_path / f"case{i}")
        for d in r["dirs"]:
            os.makedirs(os.path.join(root, d))
        name = ckpt.latest_checkpoint(root, r["resume_from_checkpoint"])
        assert ([name] if name else []) == r["loaded"], r
        if name:
            step = ckpt.global_step_of(name)
            assert step == r["global_step"]
            assert ckpt.resume_position(step, 300, 2) == (r["first_epoch"], r["resume_step"])
        assert ckpt.rotate_checkpoints(root, r["checkpoints_total_limit"]) == r["removed"], r
        os.makedirs(os.path.join(root, f"checkpoint-{r['save_at_step']}"))
        assert sorted(os.listdir(root)) == sorted(set(r["dirs"]) - set(r["removed"]) | set(r["created"]))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_resume_continues_the_same_trajectory(emu_backend, tmp_path, dtype):
    """5 steps straight == 3 steps, save_state, fresh process state, load_state, 2 steps (weights, moments, loss scale, lr)."""
    kw = dict(dtype=dtype, lr=1e-2, init_scale=64.0, growth_interval=2)
    skw = dict(num_warmup_steps=2, num_training_steps=8)
    batches = [batch_of(10 + i) for i in range(5)]

    a = Trainer(build(4), **kw)
    sa = get_scheduler("cosine", optimizer=a, **skw)
    ema_a = EMAModel(a.model.parameters(), decay=0.9, model_cls=UNetSpatioTemporalConditionModel, model_config=a.model.config)
    for b in batches:
        a.step(b)
        ema_a.step(a.model.parameters())

    c = Trainer(build(4), **kw)
    sc = get_scheduler("cosine", optimizer=c, **skw)
    ema_c = EMAModel(c.model.parameters(), decay=0.9, model_cls=UNetSpatioTemporalConditionModel, model_config=c.model.config)
    for b in batches[:3]:
        c.step(b)
        ema_c.step(c.model.parameters())
    path = str(tmp_path / "checkpoint-3")
    c.save_state(path, ema=ema_c, scheduler=sc)
    want = {"unet", "unet_ema", "optimizer.bin", "scheduler.bin", "random_states_0.pkl"} | ({"scaler.pt"} if dtype == torch.float16 else set())
    assert set(os.listdir(path)) == want

    d = Trainer(build(99), **kw)                                    # different weights: everything must come from the folder
    sd_ = get_scheduler("cosine", optimizer=d, **skw)
    ema_d = EMAModel(d.model.parameters(), decay=0.5, model_cls=UNetSpatioTemporalConditionModel, model_config=d.model.config)
    d.load_state(path, ema=ema_d, scheduler=sd_)
    assert ema_d.decay == 0.9 and ema_d.optimization_step == 3
    assert torch.equal(d.opt_state.cpu()[:3], c.opt_state.cpu()[:3])
    for b in batches[3:]:
        d.step(b)
        ema_d.step(d.model.parameters())
    assert torch.equal(d.p_flat, a.p_flat) and torch.equal(d.m_flat, a.m_flat) and torch.equal(d.v_flat, a.v_flat)
    assert torch.equal(d.opt_state.cpu(), a.opt_state.cpu())
    assert sd_.get_last_lr() == sa.get_last_lr()
    for x, y in zip(ema_d.shadow_params, ema_a.shadow_params):
        assert torch.equal(x, y)
    # frozen weights were re-packed from the folder as well
    for (n, p), (_, q) in zip(d.model.named_parameters(), a.model.named_parameters()):
        assert torch.equal(p.data, q.data), n


def test_resume_with_lora_adapters(emu_backend, tmp_path):
    """train_svd_lora.py uses the same save / load hooks (:693-715): the unet/ folder then carries peft's key names."""
    from svd_xtend_amd.lora import LoraConfig
    # r = 64: the flat layout regroups the adapter factors, optimizer.bin keeps the optimizer's (named_parameters) order
    cfg = LoraConfig(r=64, lora_alpha=64, init_lora_weights="gaussian", target_modules=["to_k", "to_q", "to_v", "to_out.0"])

    def fresh(seed):
        m = build(seed)
        for p in m.parameters():
            p.requires_grad_(False)
        torch.manual_seed(seed)
        m.add_adapter(cfg)
        return Trainer(m, dtype=torch.float32, lr=1e-2)
    batches = [batch_of(30 + i) for i in range(3)]
    a = fresh(7)
    for b in batches:
        a.step(b)
    c = fresh(7)
    for b in batches[:2]:
        c.step(b)
    path = str(tmp_path / "checkpoint-2")
    c.save_state(path)
    from safetensors.torch import load_file
    keys = set(load_file(os.path.join(path, "unet", "diffusion_pytorch_model.safetensors")))
    assert any(k.endswith("to_q.lora_A.default.weight") for k in keys) and any(k.endswith("to_q.base_layer.weight") for k in keys)
    osd = torch.load(os.path.join(path, "optimizer.bin"), weights_only=False)
    names = [n for n, p in c.model.named_parameters() if p.requires_grad]
    assert all(tuple(osd["state"][i]["exp_avg"].shape) == tuple(dict(c.model.named_parameters())[n].shape) for i, n in enumerate(names))
    d = fresh(8)
    d.load_state(path)
    d.step(batches[2])
    assert torch.equal(d.p_flat, a.p_flat) and torch.equal(d.m_flat, a.m_flat) and float(d.opt_state[0]) == 3.0


def test_optimizer_bin_is_a_torch_adamw_state_dict(emu_backend, tmp_path):
    """optimizer.bin written here loads into torch.optim.AdamW over the same parameter list, and one written by torch loads here."""
    tr = Trainer(build(5), dtype=torch.float32, lr=1e-3)
    b = batch_of(20)
    tr.step(b)
    tr.step(b)
    sd = ckpt.optimizer_state_dict(tr)
    twins = [torch.nn.Parameter(p.detach().clone()) for p in tr.params]
    opt = torch.optim.AdamW(twins, lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    opt.load_state_dict(sd)
    assert float(opt.state[twins[0]]["step"]) == 2.0
    # one more step on both sides from the same gradients
    tr.zero_grad()
    tr.forward_backward(**b)
    for t, p in zip(twins, tr.params):
        t.grad = p.grad.detach().clone()
    opt.step()
    tr.optimizer_step()
    for t, p in zip(twins, tr.params):
        assert float((t.data - p.data).abs().max()) <= 1e-6
    # and back: torch's state into a fresh trainer
    tr2 = Trainer(build(5), dtype=torch.float32, lr=1e-3)
    ckpt.load_optimizer_state_dict(tr2, opt.state_dict())
    assert float(tr2.opt_state[0]) == 3.0
    assert float((tr2.m_flat - tr.m_flat).abs().max()) <= 1e-7 and float((tr2.v_flat - tr.v_flat).abs().max()) <= 1e-9
    with pytest.raises(ValueError, match="hyper-parameters differ"):
        ckpt.load_optimizer_state_dict(Trainer(build(5), dtype=torch.float32, lr=5e-4), opt.state_dict())


def test_lora_weights_file_round_trip(emu_backend, tmp_path):
    from svd_xtend_amd.lora import LORA_WEIGHT_NAME_SAFE, LoraConfig, load_lora_weights, lora_state_dict, save_lora_weights
    cfg = LoraConfig(r=8, lora_alpha=8, init_lora_weights="gaussian", target_modules=["to_k", "to_q", "to_v", "to_out.0"])
    m = build(6)
    m.add_adapter(cfg)
    for n, p in m.named_parameters():
        if ".lora_B." in n:
            torch.nn.init.normal_(p, std=0.02)
    path = save_lora_weights(str(tmp_path), lora_state_dict(m))
    assert os.path.basename(path) == LORA_WEIGHT_NAME_SAFE
    m2 = build(6)
    m2.add_adapter(cfg)
    load_lora_weights(m2, str(tmp_path))
    for (n, p), (_, q) in zip(m.named_parameters(), m2.named_parameters()):
        assert torch.equal(p.data, q.data), n
    assert math.isfinite(float(sum(p.sum() for p in m2.parameters())))
