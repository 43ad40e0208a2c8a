"""End-to-end parity: MI355X-native UNet train step (HIP kernels) vs the CPU oracle on identical seeded
weights / latents / sigmas.  Metric (BASELINE.json north_star): |loss_gpu - loss_cpu| / loss_cpu <= 1e-3 (fp16)."""
import copy
import json
import os
import time

import torch

from oracle.step import edm_inputs, edm_loss, make_optimizer, make_synthetic_batch
from oracle.unet import TINY_CONFIG, UNetSpatioTemporalConditionOracle as _Oracle, no_default_init, scaled_init_
from svd_xtend_amd.train import Trainer
from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel as _Product


def UNetSpatioTemporalConditionOracle(**cfg):
    """Every caller below fills all parameters right after (scaled_init_): skip the constructors' own kaiming draws (16-33 s per model at the real widths)."""
    with no_default_init():
        return _Oracle(**cfg)


def UNetSpatioTemporalConditionModel(**cfg):
    """The product model, constructed for a `load_state_dict(strict=True)` that follows at once."""
    with no_default_init():
        return _Product(**cfg)


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a @ b) / (a.norm() * b.norm() + 1e-30))


def oracle_step(cfg, B, T, h, w, seed, lr, cross_dim, lora_r=0, orc=None, with_pred_after=True, more_steps=0):
    """One optimizer step of the CPU oracle on seeded weights and a seeded batch: everything `compare` holds the product to.
    more_steps > 0: the oracle keeps stepping on the same batch (the trajectory test's reference) -- `traj` then carries every step's loss
    and the accumulated update of each trainable tensor, and the forward of step 2 IS the prediction of the updated weights (`pred_after`
    costs no pass of its own)."""
    t0 = time.time()
    if orc is None:
        orc = UNetSpatioTemporalConditionOracle(**cfg)
        scaled_init_(orc, seed)
    if lora_r:                                   # config 5: adapters are the trainable set (B randomised so dA is non-zero)
        from oracle.lora import add_adapter
        for p in orc.parameters():
            p.requires_grad_(False)
        torch.manual_seed(seed + 5)              # peft's "gaussian" init of A draws from the global generator
        add_adapter(orc, lora_r, lora_r)
        gen = torch.Generator().manual_seed(seed + 17)
        for n, p in orc.named_parameters():
            if ".lora_B." in n:
                p.data.copy_(torch.randn(p.shape, generator=gen) * 0.05)
    batch = make_synthetic_batch(B, T, h, w, seed + 1, cross_dim=cross_dim)
    opt = make_optimizer(orc, lr=lr) if not lora_r else torch.optim.AdamW([p for p in orc.parameters() if p.requires_grad], lr=lr,
                                                                          betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    sd0 = copy.deepcopy(orc.state_dict())
    t1 = time.time()
    unet_in, ts, ehs, ids, noisy, sig = edm_inputs(batch)
    pred = orc(unet_in, ts, ehs, added_time_ids=ids).sample
    loss = edm_loss(pred, noisy, batch["latents"], sig)
    t2 = time.time()
    loss.backward()
    t3 = time.time()
    grads = {n: p.grad.clone() for n, p in orc.named_parameters() if p.grad is not None}
    opt.step()
    params_after = {n: p.detach().clone() for n, p in orc.named_parameters() if p.requires_grad}
    pred_after, traj = None, None
    if more_steps:
        p0 = {n: sd0[n] for n in params_after}
        losses = [float(loss.detach())]
        for k in range(more_steps):
            opt.zero_grad()
            pk = orc(unet_in, ts, ehs, added_time_ids=ids).sample
            if k == 0:
                pred_after = pk.detach()
            lk = edm_loss(pk, noisy, batch["latents"], sig)
            lk.backward()
            opt.step()
            losses.append(float(lk.detach()))
        traj = dict(losses=losses, update={n: p.detach() - p0[n] for n, p in orc.named_parameters() if p.requires_grad})
    elif with_pred_after:
        with torch.no_grad():                    # the prediction of the UPDATED weights (checks the optimizer step end to end)
            pred_after = orc(unet_in, ts, ehs, added_time_ids=ids).sample
    t4 = time.time()
    print(f"[oracle_step] {T}x{h}x{w} build {t1 - t0:.1f}s forward {t2 - t1:.1f}s backward {t3 - t2:.1f}s rest {t4 - t3:.1f}s "
          f"({torch.get_num_threads()} threads)", flush=True)
    return dict(sd0=sd0, batch=batch, inputs=(unet_in, ts, ehs, ids, noisy), loss=float(loss.detach()), pred=pred.detach(),
                pred_after=pred_after, lr=lr, grads=grads, params_after=params_after, traj=traj)


BIG_REF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "_big")


def check_big_ref_fingerprint(tag, ref, rel=2e-5):
    """Hold a big oracle reference (cached file or fresh computation) to the fingerprint this repository committed for it
    (tests/golden/big_ref_fingerprints.json, written by tests/golden/make_big_fingerprint.py): loss, seeded-weight fingerprint, the L2 norm
    of the prediction and of every gradient.  The cache itself is untracked (145 MB); what it must contain is not.  rel: fp32 oracle
    results move in the last digits with the host's thread count (reduction order)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_ref_fingerprints.json")
    if not os.path.exists(path):
        return False
    want = json.load(open(path)).get(tag.replace(" ", "_"))
    if want is None:
        return False
    close = lambda a, b: abs(a - b) <= rel * max(abs(a), abs(b), 1e-30)      # noqa: E731
    assert close(ref["loss"], want["loss"]), (tag, "loss", ref["loss"], want["loss"])
    assert close(ref["sd0_fingerprint"], want["sd0_fingerprint"]), (tag, "seeded weights", ref["sd0_fingerprint"], want["sd0_fingerprint"])
    assert close(float(ref["pred"].double().norm()), want["pred_l2"]), (tag, "prediction norm")
    assert len(ref["grads"]) == want["n_grads"], (tag, len(ref["grads"]), want["n_grads"])
    for k, v in ref["grads"].items():
        assert close(float(v.double().norm()), want["grad_l2"][k]), (tag, k, float(v.double().norm()), want["grad_l2"][k])
    return True


def oracle_step_cached(tag, cfg, B, T, h, w, seed, lr, cross_dim):
    """`oracle_step` with its result kept on disk (tests/golden/_big/<tag>.pt: git-ignored, travels with the gpurun snapshot) for the
    cases whose CPU oracle takes minutes and tens of GB -- config 4's upper levels: computed once wherever there is a host for it
    (`SVDX_SAVE_BIG_REF=1 python tests/golden/make_big_refs.py`), compared on the GPU box without burning GPU-minutes on CPU work.  The
    seeded weights are NOT stored: they are re-drawn from `seed` and held to the stored fingerprint."""
    path = os.path.join(BIG_REF_DIR, tag.replace(" ", "_") + ".pt")
    if os.path.exists(path):
        ref = torch.load(path, weights_only=False)
        orc = UNetSpatioTemporalConditionOracle(**cfg)
        scaled_init_(orc, seed)
        sd0 = copy.deepcopy(orc.state_dict())
        fp = float(sum(v.double().abs().sum() for v in sd0.values()))
        if abs(fp - ref["sd0_fingerprint"]) <= 1e-9 * abs(fp):
            check_big_ref_fingerprint(tag, ref)
            ref["sd0"] = sd0
            ref["cached"] = path
            return ref
        print(f"[e2e_checks] {path}: seeded weights differ from the stored fingerprint ({fp} vs {ref['sd0_fingerprint']}); recomputing", flush=True)
    ref = oracle_step(cfg, B, T, h, w, seed=seed, lr=lr, cross_dim=cross_dim)
    check_big_ref_fingerprint(tag, dict(ref, sd0_fingerprint=float(sum(v.double().abs().sum() for v in ref["sd0"].values()))))
    if os.environ.get("SVDX_SAVE_BIG_REF") == "1":
        os.makedirs(BIG_REF_DIR, exist_ok=True)
        keep = {k: v for k, v in ref.items() if k != "sd0"}
        keep["sd0_fingerprint"] = float(sum(v.double().abs().sum() for v in ref["sd0"].values()))
        torch.save(keep, path)
    return ref


_SESSION_REFS = {}


def oracle_steps_shared(cfg, B, T, h, w, seed, lr, cross_dim, steps=3):
    """`steps` oracle steps on one batch, computed once per pytest session: the 64x40-level block test reads step 1 (loss, gradients,
    updated weights, and step 2's forward as the prediction after the update), the trajectory test reads all of them -- one ~25 s CPU
    run of the 35840-row level instead of four."""
    key = (repr(sorted(cfg.items())), B, T, h, w, seed, lr, cross_dim, steps)
    if key not in _SESSION_REFS:
        _SESSION_REFS[key] = oracle_step(cfg, B, T, h, w, seed=seed, lr=lr, cross_dim=cross_dim, more_steps=steps - 1)
    return _SESSION_REFS[key]


def product_step(ref, cfg, dtype, dev, lr, lora_r=0):
    m = UNetSpatioTemporalConditionModel(**cfg)
    if lora_r:
        from svd_xtend_amd.lora import LoraConfig
        m.add_adapter(LoraConfig(r=lora_r, lora_alpha=lora_r, init_lora_weights="gaussian"))
    m.load_state_dict(ref["sd0"], strict=True)
    m.to(dev)
    tr = Trainer(m, dtype=dtype, lr=lr)
    unet_in, ts, ehs, ids, noisy = (t.to(dev) for t in ref["inputs"])
    b = ref["batch"]
    with torch.no_grad():
        pred0 = m(unet_in, ts, ehs, ids).sample.float().cpu()
    tr.zero_grad()
    tr.forward_backward(unet_in, ts, ehs, ids, noisy, b["latents"].to(dev), b["sigmas"].to(dev))
    loss = float(tr.last_loss())
    scale = float(tr.opt_state[1])
    grads = {n: (p.grad.detach().float().cpu() / scale) for n, p in m.named_parameters() if p.requires_grad}
    tr.optimizer_step()
    params = {n: p.detach().float().cpu() for n, p in m.named_parameters() if p.requires_grad}
    with torch.no_grad():
        pred = m(unet_in, ts, ehs, ids).sample.float().cpu()
    return dict(loss=loss, grads=grads, params_after=params, pred=pred0, pred_after=pred, state=tr.opt_state.cpu().tolist())


def compare(ref, got):
    out = dict(loss_ref=ref["loss"], loss=got["loss"], loss_rel=abs(got["loss"] - ref["loss"]) / abs(ref["loss"]))
    cos = {n: cosine(got["grads"][n], g) for n, g in ref["grads"].items() if n in got["grads"] and float(g.abs().max()) > 1e-12}
    out["grad_cos_min"] = min(cos.values())
    out["grad_cos_worst"] = min(cos, key=cos.get)
    gn_ref = sum(float(g.double().pow(2).sum()) for g in ref["grads"].values()) ** 0.5
    gn = sum(float(g.double().pow(2).sum()) for g in got["grads"].values()) ** 0.5
    out["grad_norm_rel"] = abs(gn - gn_ref) / gn_ref
    out["n_grads"] = len(cos)
    out["param_max_diff"] = max(float((got["params_after"][n] - p).abs().max()) for n, p in ref["params_after"].items())
    n_el = sum(p.numel() for p in ref["params_after"].values())
    out["param_mean_diff"] = sum(float((got["params_after"][n] - p).abs().sum()) for n, p in ref["params_after"].items()) / n_el
    out["lr"] = ref.get("lr")
    out["pred_rel_l2"] = rel_l2(got["pred"], ref["pred"]) if "pred" in got else None
    out["pred_after_rel_l2"] = rel_l2(got["pred_after"], ref["pred_after"]) if ref.get("pred_after") is not None else None
    out["opt_state"] = got["state"]
    return out


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _log_measured(key, r) -> None:
    """One line per oracle comparison into gpurun_out/parity_measured.jsonl (when that directory exists: the GPU box): what the bars of
    `assert_parity` are set against."""
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if not os.path.isdir(out) or "error" in r:
        return
    import json
    keep = {k: r.get(k) for k in ("loss_rel", "grad_cos_min", "grad_cos_worst", "grad_norm_rel", "pred_rel_l2", "pred_after_rel_l2", "param_max_diff",
                                  "param_mean_diff", "lr", "n_grads")}
    with open(os.path.join(out, "parity_measured.jsonl"), "a") as f:
        f.write(json.dumps({"case": key, **keep}) + "\n")


def assert_parity(key, r, bf16=None):
    """The acceptance bar of every oracle comparison (north_star: noise-prediction MSE within 1e-3 relative at fp16; bf16 carries
    8x less mantissa).  One AdamW step moves a weight by at most ~lr, and where the gradient is at rounding level its sign -- and
    with it the whole update -- may differ: the worst weight is bounded by 2.5 lr, the mean weight must agree far below lr."""
    assert "error" not in r, f"{key}: {r}"
    bf16 = ("bfloat16" in key) if bf16 is None else bf16          # NB: "float16" is a substring of "bfloat16"
    _log_measured(key, r)
    assert r["loss_rel"] <= (8e-3 if bf16 else 1e-3), f"{key}: loss rel err {r['loss_rel']:.3e}"
    # gradient direction of every trainable tensor: measured >= 0.99998 (fp16) / >= 0.9998 (bf16) over the suite (profiles/r6_parity_measured.jsonl);
    # rounds 1-5 held 0.99 / 0.95 -- two orders looser than the data.  A mis-routed or mis-scaled gradient gives ~0, a dropped launch < 0.9.
    assert r["grad_cos_min"] >= (0.995 if bf16 else 0.9995), f"{key}: {r}"
    # prediction (the tensor the reference's loss is built from): 16-bit storage through ~100 chained layers.  Measured at fp16 on the full
    # model: 1.1e-3 (c2, bench.py's seed) / 1.3e-3 (c1') / 2.28e-3 (c2, this suite's seed: round 5's test_full_topology_c2_matches_oracle);
    # 1.96e-3 on the tiny LoRA topology; bf16 up to 1.57e-2.  Fixed since round 5 (3e-3 / 2e-2); north_star's own metric is the loss above
    assert r["pred_rel_l2"] is None or r["pred_rel_l2"] <= (2e-2 if bf16 else 3e-3), f"{key}: {r}"
    assert r["pred_after_rel_l2"] is None or r["pred_after_rel_l2"] <= (2e-2 if bf16 else 3e-3), f"{key}: {r}"
    if r.get("lr"):
        assert r["param_max_diff"] <= 2.5 * r["lr"], f"{key}: {r}"
        assert r["param_mean_diff"] <= (0.2 if bf16 else 0.05) * r["lr"], f"{key}: {r}"


def run_all(verbose=False, dev=None):
    dev = dev or torch.device("cuda")
    cfg = TINY_CONFIG
    res = {}
    for (B, T, h, w) in [(1, 4, 16, 16), (2, 3, 16, 24)]:
        t0 = time.time()
        ref = oracle_step(cfg, B, T, h, w, seed=3, lr=1e-4, cross_dim=cfg["cross_attention_dim"])
        for dt in (torch.float16, torch.bfloat16):
            key = f"tiny B={B} T={T} {h}x{w} {str(dt).split('.')[-1]}"
            try:
                res[key] = compare(ref, product_step(ref, cfg, dt, dev, 1e-4))
            except Exception as e:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                res[key] = {"error": repr(e)[:400]}
            if verbose:
                print(key, res[key], f"{time.time() - t0:.1f}s", flush=True)
    return res


def graphed_vs_eager(dev=None, dtype=torch.float16, steps=3):
    """The chained-hipGraph step (train.GraphedStep) must walk the same trajectory as eager `Trainer.step` calls."""
    from svd_xtend_amd.train import GraphedStep
    dev = dev or torch.device("cuda")
    cfg = TINY_CONFIG
    orc = UNetSpatioTemporalConditionOracle(**cfg)
    scaled_init_(orc, 5)
    b = make_synthetic_batch(1, 3, 16, 16, 77, cross_dim=cfg["cross_attention_dim"])
    unet_in, ts, ehs, ids, noisy, _ = edm_inputs(b)
    batch = {k: v.to(dev) for k, v in dict(unet_in=unet_in, timesteps=ts, ehs=ehs, added_time_ids=ids, noisy_latents=noisy,
                                           target=b["latents"], sigmas=b["sigmas"]).items()}
    out = []
    for mode in ("eager", "graph"):
        m = UNetSpatioTemporalConditionModel(**cfg)
        m.load_state_dict(orc.state_dict(), strict=True)
        m.to(dev)
        tr = Trainer(m, dtype=dtype, lr=1e-3)
        if mode == "eager":
            for _ in range(steps):
                tr.step(batch)
            segs = 0
        else:
            gs = GraphedStep(tr, batch, cut_blocks=True)   # its warm-up pass is one real step; one rank rehearses the multi-rank chain
            for _ in range(steps - 1):
                gs()
            segs = len(gs.graphs)
        torch.cuda.synchronize()
        out.append(dict(p=tr.p_flat.clone(), loss=float(tr.last_loss()), state=tr.opt_state.cpu().tolist(), segments=segs))
    e, g = out
    return dict(loss_eager=e["loss"], loss_graph=g["loss"], param_max_diff=float((e["p"] - g["p"]).abs().max()),
                param_mean_diff=float((e["p"] - g["p"]).abs().mean()),
                param_abs_max=float(e["p"].abs().max()), opt_steps=(e["state"][0], g["state"][0]), segments=g["segments"])


def plan_vs_graph(dev=None, dtype=torch.float16, steps=4):
    """One captured step replayed `steps - 1` times as hipGraphs vs. the launch plan recorded during the same kind of capture replayed through
    svdx_plan_replay (the capture's warm-up pass is step 1 of both)."""
    from svd_xtend_amd.train import GraphedStep
    dev = dev or torch.device("cuda")
    cfg = TINY_CONFIG
    sd = seeded_weights(cfg, 5)
    b = make_synthetic_batch(1, 3, 16, 16, 77, cross_dim=cfg["cross_attention_dim"])
    unet_in, ts, ehs, ids, noisy, _ = edm_inputs(b)
    out = []
    for use_plan in (False, True):
        batch = {k: v.to(dev) for k, v in dict(unet_in=unet_in, timesteps=ts, ehs=ehs, added_time_ids=ids, noisy_latents=noisy,
                                               target=b["latents"], sigmas=b["sigmas"]).items()}
        m = UNetSpatioTemporalConditionModel(**cfg)
        m.load_state_dict(sd, strict=True)
        m.to(dev)
        tr = Trainer(m, dtype=dtype, lr=1e-3)
        gs = GraphedStep(tr, batch, record_plan=use_plan)
        for _ in range(steps - 1):
            gs.replay_plan() if use_plan else gs()
        torch.cuda.synchronize()
        out.append(dict(p=tr.p_flat.clone(), m=tr.m_flat.clone(), loss=float(tr.loss_slot.cpu()), state=tr.opt_state.cpu().tolist(),
                        launches=gs.plan.launches if use_plan else 0, bytes=gs.plan.host_bytes if use_plan else 0))
    g, p = out
    return dict(loss_graph=g["loss"], loss_plan=p["loss"], param_max_diff=float((g["p"] - p["p"]).abs().max()),
                m_max_diff=float((g["m"] - p["m"]).abs().max()), opt_steps=(g["state"][0], p["state"][0]),
                plan_launches=p["launches"], plan_host_bytes=p["bytes"])


def seeded_weights(cfg, seed):
    """State dict of the oracle topology under scaled_init_(seed) (the product's keys are the same)."""
    orc = UNetSpatioTemporalConditionOracle(**cfg)
    scaled_init_(orc, seed)
    return orc.state_dict()


def replays_with_traffic_between(dev=None, dtype=torch.float16, replays=4, disturb=True, cfg=None, geom=(1, 3, 16, 16), lora_r=0, sd=None):
    """`replays` replays of the captured step with -- when `disturb` -- the things a real training loop does between two replays: a new
    batch copied into the captured input tensors (here: the same values, so the trajectory must not move), the loss read on the host,
    ATen launches, a matmul.  Returns the flat parameters, the loss slot and the optimizer state.  lora_r: config 5's trainable set
    (a rank below the K granule of 64 also captures the padded-rank re-layout of the adapters, ops.LoraOp.refresh)."""
    from svd_xtend_amd.train import GraphedStep
    dev = dev or torch.device("cuda")
    cfg = cfg or TINY_CONFIG
    B, T, h, w = geom
    sd = sd if sd is not None else seeded_weights(cfg, 5)       # sd: the caller's seeded_weights(cfg, 5), drawn once for both of its runs
    b = make_synthetic_batch(B, T, h, w, 77, cross_dim=cfg["cross_attention_dim"])
    unet_in, ts, ehs, ids, noisy, _ = edm_inputs(b)
    host = dict(unet_in=unet_in, timesteps=ts, ehs=ehs, added_time_ids=ids, noisy_latents=noisy, target=b["latents"], sigmas=b["sigmas"])
    batch = {k: v.to(dev) for k, v in host.items()}
    m = UNetSpatioTemporalConditionModel(**cfg)
    m.load_state_dict(sd, strict=True)
    if lora_r:
        from svd_xtend_amd.lora import LoraConfig
        torch.manual_seed(11)
        m.add_adapter(LoraConfig(r=lora_r, lora_alpha=lora_r, init_lora_weights="gaussian"))
        gen = torch.Generator().manual_seed(23)
        for n, p in m.named_parameters():
            if ".lora_B." in n:
                p.data.copy_(torch.randn(p.shape, generator=gen) * 0.05)
    m.to(dev)
    tr = Trainer(m, dtype=dtype, lr=1e-3)
    gs = GraphedStep(tr, batch)
    losses = []
    big = torch.ones(1 << 20, device=dev)
    a = torch.ones(64, 64, device=dev, dtype=torch.float16)
    for i in range(replays):
        gs()
        if disturb:
            losses.append(float(tr.last_loss()))                 # an ATen launch + a D2H copy + a host sync
            for k, v in host.items():
                batch[k].copy_(v)                                # the next batch: H2D copies into the captured tensors
            _ = big + 1                                          # eager launches with allocations of their own
            _ = a @ a
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return dict(p=tr.p_flat.clone(), loss=float(tr.loss_slot.cpu()), state=tr.opt_state.cpu().tolist(), losses=losses)


def run_steps(dev=None, dtype=torch.float16, steps=3, seed=5, lora_r=0, rt_attrs=None):
    """`steps` eager optimizer steps of the tiny topology from seeded weights on a seeded batch; returns the final state.
    lora_r: config 5's trainable set (adapters on the attention projections, B randomised so that every gradient is non-zero)."""
    dev = dev or torch.device("cuda")
    cfg = TINY_CONFIG
    orc = UNetSpatioTemporalConditionOracle(**cfg)
    scaled_init_(orc, seed)
    b = make_synthetic_batch(1, 3, 16, 16, 77, cross_dim=cfg["cross_attention_dim"])
    unet_in, ts, ehs, ids, noisy, _ = edm_inputs(b)
    batch = {k: v.to(dev) for k, v in dict(unet_in=unet_in, timesteps=ts, ehs=ehs, added_time_ids=ids, noisy_latents=noisy,
                                           target=b["latents"], sigmas=b["sigmas"]).items()}
    m = UNetSpatioTemporalConditionModel(**cfg)
    m.load_state_dict(orc.state_dict(), strict=True)
    if lora_r:
        from svd_xtend_amd.lora import LoraConfig
        torch.manual_seed(seed + 5)
        m.add_adapter(LoraConfig(r=lora_r, lora_alpha=lora_r, init_lora_weights="gaussian"))
        gen = torch.Generator().manual_seed(seed + 17)
        for n, p in m.named_parameters():
            if ".lora_B." in n:
                p.data.copy_(torch.randn(p.shape, generator=gen) * 0.05)
    m.to(dev)
    tr = Trainer(m, dtype=dtype, lr=1e-3)
    for name, val in (rt_attrs or {}).items():      # Runtime switches (ops.Runtime), set before the first step
        assert hasattr(tr.rt, name), name
        setattr(tr.rt, name, val)
    for _ in range(steps - 1):
        tr.step(batch)
    tr.rt.k.launch_log = log = []              # entry names of the last step's launches
    try:
        tr.step(batch)
    finally:
        tr.rt.k.launch_log = None
    if dev.type == "cuda":
        torch.cuda.synchronize()
    id2name = {id(p): n for n, p in m.named_parameters()}
    layout = sorted((o, p.numel(), id2name.get(id(p), "?")) for o, p in zip(tr.offsets, tr.params))      # (offset, numel, name) of the flat buffers
    return dict(p=tr.p_flat.clone(), m=tr.m_flat.clone(), v=tr.v_flat.clone(), loss=float(tr.last_loss()), launches=[e[0] for e in log],
                layout=layout)


def batched_vs_single_small_launches(dev=None, dtype=torch.float16, steps=2, lora_r=0):
    """The table-driven skinny launches (Runtime.batch_small: the cross-attention vector chain up front, its gradient chain and the
    LayerNorm affine-gradient reductions at the end of the sweep) against one launch each (batch_small=False): identical bits after
    `steps` optimizer steps, and the launches they save.  Returns (batched, single) run_steps results."""
    a = run_steps(dev=dev, dtype=dtype, steps=steps, lora_r=lora_r, rt_attrs=dict(batch_small=True))
    b = run_steps(dev=dev, dtype=dtype, steps=steps, lora_r=lora_r, rt_attrs=dict(batch_small=False))
    return a, b


def assert_batched_equals_single(a, b, lora=False, exact=True):
    """exact: bit equality -- both forms inline the same device function; asserted on the simulator and, since round 4 looked at the
    hardware result (tools/batched_diff.py: not one element differs), on the GPU too.  exact=False (rounding level) is kept for builds
    whose two instantiations a compiler schedules differently."""
    if exact:
        assert a["loss"] == b["loss"] and all(torch.equal(a[k], b[k]) for k in ("p", "m", "v")), "batched skinny launches changed the step's bits"
    else:
        assert abs(a["loss"] - b["loss"]) <= 1e-6 * abs(b["loss"]), (a["loss"], b["loss"])
        for k in ("p", "m", "v"):
            d = float((a[k] - b[k]).abs().max())
            assert d <= 1e-5 * float(b[k].abs().max()) + 1e-12, (k, d)
    na, nb = a["launches"], b["launches"]
    # base training: 2 forward stages + 1 transposed stage, the outer products, the LayerNorm reductions; adapters (frozen LayerNorms):
    # 4 forward stages + 3 transposed stages, the outer products
    want = (7, 1, 0) if lora else (3, 1, 1)
    assert (na.count("svdx_small_linear_batch"), na.count("svdx_outer_acc_batch"), na.count("svdx_ln_param_reduce_batch")) == want, \
        {n: na.count(n) for n in set(na) if "batch" in n}
    assert not any("batch" in n for n in nb)
    assert na.count("svdx_outer_acc") == 0 and nb.count("svdx_outer_acc") > 0
    assert len(na) < len(nb), (len(na), len(nb))


def resume_vs_straight(tmpdir, dev=None, dtype=torch.float16, steps=4, cut=2):
    """`steps` optimizer steps in one go vs `cut` steps, save_state, a fresh trainer from other weights, load_state, the rest:
    weights, Adam moments, loss scale, the device-side lr schedule (graph-replayed steps after the resume) and the EMA."""
    import os

    from svd_xtend_amd.optimization import get_scheduler
    from svd_xtend_amd.train import GraphedStep
    from svd_xtend_amd.training_utils import EMAModel
    dev = dev or torch.device("cuda")
    cfg = TINY_CONFIG
    b = make_synthetic_batch(1, 3, 16, 16, 78, cross_dim=cfg["cross_attention_dim"])
    unet_in, ts, ehs, ids, noisy, _ = edm_inputs(b)
    batch = {k: v.to(dev) for k, v in dict(unet_in=unet_in, timesteps=ts, ehs=ehs, added_time_ids=ids, noisy_latents=noisy,
                                           target=b["latents"], sigmas=b["sigmas"]).items()}

    def fresh(seed):
        orc = UNetSpatioTemporalConditionOracle(**cfg)
        scaled_init_(orc, seed)
        m = UNetSpatioTemporalConditionModel(**cfg)
        m.load_state_dict(orc.state_dict(), strict=True)
        m.to(dev)
        tr = Trainer(m, dtype=dtype, lr=1e-3)
        sched = get_scheduler("cosine", optimizer=tr, num_warmup_steps=1, num_training_steps=steps + 2)
        ema = EMAModel(m.parameters(), decay=0.8, model_cls=_Product, model_config=m.config,
                       on_weights_changed=tr.weights_changed)
        return tr, sched, ema

    a, sa, ema_a = fresh(6)
    lrs_a = []
    for _ in range(steps):
        a.step(batch)
        ema_a.step(a.model.parameters())
        lrs_a.append(1e-3 * float(a.opt_state[8]))
    c, sc, ema_c = fresh(6)
    for _ in range(cut):
        c.step(batch)
        ema_c.step(c.model.parameters())
    path = os.path.join(str(tmpdir), f"checkpoint-{cut}")
    c.save_state(path, ema=ema_c, scheduler=sc)
    d, sd_, ema_d = fresh(123)
    d.load_state(path, ema=ema_d, scheduler=sd_)
    gs = GraphedStep(d, batch, cut_blocks=True)  # its warm-up pass is one real step
    ema_d.step(d.model.parameters())
    lrs_d = [1e-3 * float(d.opt_state[8])]
    for _ in range(steps - cut - 1):
        gs()
        ema_d.step(d.model.parameters())
        lrs_d.append(1e-3 * float(d.opt_state[8]))
    torch.cuda.synchronize()
    n = a.n_flat
    sh_a = torch.cat([t.reshape(-1) for t, p in zip(ema_a.shadow_params, a.model.parameters()) if p.requires_grad])
    sh_d = torch.cat([t.reshape(-1) for t, p in zip(ema_d.shadow_params, d.model.parameters()) if p.requires_grad])
    # EMA swap on the prepared model: predictions change, and come back after restore
    with torch.no_grad():
        y0 = d.model(batch["unet_in"], batch["timesteps"], batch["ehs"], batch["added_time_ids"]).sample.float().clone()
        packed0 = (d.p_flat.clone(), d.rt.w16_flat.clone(), d.rt.wt16_flat[:d.rt.wt_pos].clone())
        ema_d.store(d.model.parameters())
        ema_d.copy_to(d.model.parameters())
        y1 = d.model(batch["unet_in"], batch["timesteps"], batch["ehs"], batch["added_time_ids"]).sample.float().clone()
        ema_d.restore(d.model.parameters())
        y2 = d.model(batch["unet_in"], batch["timesteps"], batch["ehs"], batch["added_time_ids"]).sample.float().clone()
        packed2 = (d.p_flat, d.rt.w16_flat, d.rt.wt16_flat[:d.rt.wt_pos])
        restored = all(torch.equal(x, y) for x, y in zip(packed0, packed2))
    return dict(weights_restored_exactly=restored, files=sorted(os.listdir(path)), opt_steps=(float(a.opt_state[0]), float(d.opt_state[0])),
                scale=(float(a.opt_state[1]), float(d.opt_state[1])), lrs_straight=lrs_a[cut:], lrs_resumed=lrs_d,
                param_max_diff=float((a.p_flat[:n] - d.p_flat[:n]).abs().max()),
                param_mean_diff=float((a.p_flat[:n] - d.p_flat[:n]).abs().mean()),
                m_rel=float((a.m_flat - d.m_flat).norm() / a.m_flat.norm()),
                ema_max_diff=float((sh_a - sh_d).abs().max()), ema_vs_live=float((sh_d - d.p_flat[:n][:sh_d.numel()]).abs().max()) if sh_d.numel() == n else -1.0,
                swap_changes_pred=float((y1 - y0).abs().max()), restore_pred_diff=float((y2 - y0).abs().max()),
                ema_step=ema_d.optimization_step, segments=len(gs.graphs))


def run_lora(verbose=False, dev=None, ranks=(64, 8)):
    """Config 5 on the tiny topology: LoRA adapters (train_svd_lora.py:659-674) through libsvdx vs the oracle's peft restatement."""
    dev = dev or torch.device("cuda")
    cfg = TINY_CONFIG
    res = {}
    for r in ranks:
        ref = oracle_step(cfg, 1, 3, 16, 16, seed=4, lr=1e-4, cross_dim=cfg["cross_attention_dim"], lora_r=r)
        for dt in (torch.bfloat16, torch.float16):
            key = f"lora r={r} {str(dt).split('.')[-1]}"
            try:
                res[key] = compare(ref, product_step(ref, cfg, dt, dev, 1e-4, lora_r=r))
            except Exception as e:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                res[key] = {"error": repr(e)[:400]}
            if verbose:
                print(key, res[key], flush=True)
    return res


# ----------------------------------------------------------------------------------------------------------------------
# Real widths.  One-level UNets whose every block has the channel width, head count, frame count and pixel count of ONE resolution
# level of the benched configuration (c2: 14 frames, latent 40x64): the same GEMM problems (M = 35840 / 8960 / 2240 / 560 rows,
# N and K multiples of 320), hence the same tile / split-K choices of `ops.choose_cfg`, the T = 14 -> 16 padding of the temporal
# attention, spatial sequences of 2560 / 640 / 160 / 40, the 2C -> C resnets behind a skip concat with their 1x1 shortcut.
# layers_per_block = 1 keeps the CPU oracle at seconds per case.
# ----------------------------------------------------------------------------------------------------------------------
def level_config(C, heads, cross_dim=1024, layers=1, num_frames=14):
    return dict(in_channels=8, out_channels=4, down_block_types=("CrossAttnDownBlockSpatioTemporal",),
                up_block_types=("CrossAttnUpBlockSpatioTemporal",), block_out_channels=(C,), addition_time_embed_dim=256,
                projection_class_embeddings_input_dim=768, layers_per_block=layers, cross_attention_dim=cross_dim,
                transformer_layers_per_block=1, num_attention_heads=(heads,), num_frames=num_frames)


C2_LEVELS = {           # name: (C, heads, h, w) at T = 14
    "L0 320ch 40x64": (320, 5, 40, 64),
    "L1 640ch 20x32": (640, 10, 20, 32),
    "L2 1280ch 10x16": (1280, 20, 10, 16),
    "L3 1280ch 5x8": (1280, 20, 5, 8),
}


C4_LEVELS = {           # reference config 4 (25 frames of 1024 x 576, latent 72 x 128) at T = 25: its two deepest levels, and -- opt-in, the CPU
    "L0 320ch 72x128": (320, 5, 72, 128),         # the 9216-pixel top level (230,400 rows; spatial attention over S = 9216 through the oracle's chunked path)
    "L1 640ch 36x64": (640, 10, 36, 64),          # oracle needs minutes and ~20 GB for it -- the 2304-pixel level (57,600 rows of 640 channels)
    "L2 1280ch 18x32": (1280, 20, 18, 32),
    "L3 1280ch 9x16": (1280, 20, 9, 16),
}


def run_levels(levels=None, dtypes=(torch.float16,), T=14, lora_r=0, verbose=False, dev=None, seed=11, table=None):
    dev = dev or torch.device("cuda")
    res = {}
    for name, (C, heads, h, w) in (table or C2_LEVELS).items():
        if levels is not None and name.split()[0] not in levels:
            continue
        cfg = level_config(C, heads, num_frames=T)
        t0 = time.time()
        if lora_r:
            ref = oracle_step(cfg, 1, T, h, w, seed=seed, lr=1e-4, cross_dim=cfg["cross_attention_dim"], lora_r=lora_r)
        elif table is None and T == 14 and h * w == 2560:      # shared with test_three_step_trajectory_matches_oracle[L0]
            ref = oracle_steps_shared(cfg, 1, T, h, w, seed, 1e-4, cfg["cross_attention_dim"])
        else:
            ref = oracle_step_cached(f"{name} T={T} seed={seed}", cfg, 1, T, h, w, seed, 1e-4, cfg["cross_attention_dim"])
        t_or = time.time() - t0
        for dt in dtypes:
            key = f"{name} T={T} {'lora r=%d ' % lora_r if lora_r else ''}{str(dt).split('.')[-1]}"
            try:
                res[key] = compare(ref, product_step(ref, cfg, dt, dev, 1e-4, lora_r=lora_r))
            except Exception as e:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                res[key] = {"error": repr(e)[:400]}
            if verbose:
                print(key, res[key], f"oracle {t_or:.1f}s total {time.time() - t0:.1f}s", flush=True)
        del ref
    return res


def run_full_c1(dtypes=(torch.float16, torch.bfloat16), verbose=False, dev=None):
    """c1' (SURVEY.md 8d): the FULL 1,524,623,082-parameter topology on one 8-frame 256x192 clip (latent 24x32), the oracle's
    weights through the HIP path: loss, every trainable tensor's gradient, prediction, updated weights."""
    from oracle.unet import SVD_CONFIG
    dev = dev or torch.device("cuda")
    t0 = time.time()
    ref = oracle_step(SVD_CONFIG, 1, 8, 24, 32, seed=0, lr=1e-4, cross_dim=1024)
    t_or = time.time() - t0
    res = {}
    for dt in dtypes:
        key = f"c1' full topology 8x24x32 {str(dt).split('.')[-1]}"
        try:
            res[key] = compare(ref, product_step(ref, SVD_CONFIG, dt, dev, 1e-4))
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            res[key] = {"error": repr(e)[:400]}
        if verbose:
            print(key, res[key], f"oracle {t_or:.1f}s total {time.time() - t0:.1f}s", flush=True)
        torch.cuda.empty_cache() if dev.type == "cuda" else None
    return res


def host_can_run_c2_oracle():
    """One CPU-oracle step at the benched shape (14 x 512x320) keeps ~60 GB of fp32 activations and is ~25 TFLOP: the same gate as
    bench.py's `cpu_baseline.c2` leg (>= 32 cores, >= 110 GB free)."""
    try:
        import psutil
        avail = psutil.virtual_memory().available / 2 ** 30
    except Exception:  # noqa: BLE001
        avail = 0.0
    cores = torch.get_num_threads()
    return cores >= 32 and avail >= 110.0, f"{cores} cores, {avail:.0f} GB free"


def run_full_c2(dtypes=(torch.float16,), verbose=False, dev=None):
    """c2 ITSELF (BASELINE.json configs[1], the benched shape): the full 1,524,623,082-parameter topology on one 14-frame 512x320 clip
    (latent 40x64), the oracle's weights through the HIP path -- loss, the gradient of every trainable tensor, prediction, the
    parameters after the AdamW step.  (Rounds 2-4 compared loss and prediction only, inside bench.py's cpu_baseline leg.)"""
    from oracle.unet import SVD_CONFIG
    dev = dev or torch.device("cuda")
    t0 = time.time()
    ref = oracle_step(SVD_CONFIG, 1, 14, 40, 64, seed=0, lr=1e-4, cross_dim=1024, with_pred_after=False)
    t_or = time.time() - t0
    res = {}
    for dt in dtypes:
        key = f"c2 full topology 14x40x64 {str(dt).split('.')[-1]}"
        try:
            res[key] = compare(ref, product_step(ref, SVD_CONFIG, dt, dev, 1e-4))
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc()
            res[key] = {"error": repr(e)[:400]}
        if verbose:
            print(key, res[key], f"oracle {t_or:.1f}s total {time.time() - t0:.1f}s", flush=True)
        torch.cuda.empty_cache() if dev.type == "cuda" else None
    return res


def trajectory_vs_oracle(cfg, geom, dtype=torch.float16, steps=3, lr=1e-4, seed=11, dev=None):
    """`steps` consecutive optimizer steps on one batch, oracle (autograd + torch.optim.AdamW, fp32) against the product (Trainer.step).
    A first-step loss / prediction cannot see a wrong GRADIENT (four commits of round 4 carried one); the second step's
    loss is computed on weights that the first step's gradients moved, and the per-tensor UPDATE p_final - p_0 is compared directly:
    its cosine against the oracle's update drops to ~0 for a tensor whose gradient was zeroed, mis-scaled per element or mis-routed.
    (Seed 11 = run_levels' seed: at the 64x40 level the oracle's three steps are the ones test_c2_level_blocks_match_oracle[L0] read step 1 of.)"""
    dev = dev or torch.device("cuda")
    B, T, h, w = geom
    ref = oracle_steps_shared(cfg, B, T, h, w, seed, lr, cfg["cross_attention_dim"], steps=steps)
    sd0, batch = ref["sd0"], ref["batch"]
    unet_in, ts, ehs, ids, noisy = ref["inputs"]
    ref_losses, ref_upd = ref["traj"]["losses"], ref["traj"]["update"]
    p0 = {n: sd0[n] for n in ref_upd}

    m = UNetSpatioTemporalConditionModel(**cfg)
    m.load_state_dict(sd0, strict=True)
    m.to(dev)
    tr = Trainer(m, dtype=dtype, lr=lr)
    b = {k: v.to(dev) for k, v in dict(unet_in=unet_in, timesteps=ts, ehs=ehs, added_time_ids=ids, noisy_latents=noisy,
                                       target=batch["latents"], sigmas=batch["sigmas"]).items()}
    losses = []
    for _ in range(steps):
        tr.step(b)
        losses.append(float(tr.last_loss()))
    upd = {n: p.detach().float().cpu() - p0[n] for n, p in m.named_parameters() if p.requires_grad}
    cos = {n: cosine(upd[n], u) for n, u in ref_upd.items() if float(u.abs().max()) > 0.0}
    size = {n: float(upd[n].double().norm() / (u.double().norm() + 1e-30)) for n, u in ref_upd.items() if float(u.abs().max()) > 0.0}
    worst = min(cos, key=cos.get)
    return dict(losses_ref=ref_losses, losses=losses, loss_rel=[abs(a - r) / abs(r) for a, r in zip(losses, ref_losses)],
                update_cos_min=cos[worst], update_cos_worst=worst, update_norm_ratio_min=min(size.values()),
                update_norm_ratio_max=max(size.values()), n_tensors=len(cos), opt_steps=float(tr.opt_state[0]), lr=lr)


def autograd_route(dev=None, dtype=torch.float16, cfg=None, shape=(1, 4, 16, 16), lr=1e-4):
    """The minimal-change route INTEGRATION.md shows first: the host script keeps `loss.backward()` and its own
    `torch.optim.AdamW`; `unet(...).sample` goes through `_UNetFn`, whose backward runs on autograd's worker thread (with that
    thread's current stream) -- SURVEY.md 8(b).  Compared with the oracle's step on the same weights."""
    dev = dev or torch.device("cuda")
    cfg = cfg or TINY_CONFIG
    B, T, h, w = shape
    ref = oracle_step(cfg, B, T, h, w, seed=9, lr=lr, cross_dim=cfg["cross_attention_dim"])
    m = UNetSpatioTemporalConditionModel(**cfg)
    m.load_state_dict(ref["sd0"], strict=True)
    m.to(dev)
    for n, p in m.named_parameters():                          # train_svd.py:761-766
        p.requires_grad_("temporal_transformer_block" in n)
    m.prepare(dtype)
    opt = torch.optim.AdamW([p for p in m.parameters() if p.requires_grad], lr=lr, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    unet_in, ts, ehs, ids, noisy = (t.to(dev) for t in ref["inputs"])
    b = ref["batch"]
    sig = b["sigmas"].to(dev)[:, None, None, None, None]
    scale = 1024.0 if dtype == torch.float16 else 1.0          # a GradScaler-style constant loss scale on the host side
    with torch.no_grad():
        pred0 = m(unet_in, ts, ehs, ids).sample.float().cpu()
    pred = m(unet_in, ts, ehs, ids).sample                     # train_svd.py:1021
    loss = edm_loss(pred, noisy, b["latents"].to(dev), sig)    # :1025-1036 (the oracle's statement-level restatement)
    opt.zero_grad(set_to_none=False)
    (loss * scale).backward()                                  # :1044
    torch.cuda.synchronize() if dev.type == "cuda" else None
    grads = {n: (p.grad.detach().float().cpu() / scale) for n, p in m.named_parameters() if p.requires_grad}
    for p in m.parameters():
        if p.requires_grad:
            p.grad.div_(scale)
    opt.step()                                                 # :1047
    m.refresh_trainable()
    params = {n: p.detach().float().cpu() for n, p in m.named_parameters() if p.requires_grad}
    with torch.no_grad():
        pred1 = m(unet_in, ts, ehs, ids).sample.float().cpu()
    got = dict(loss=float(loss.detach()), grads=grads, params_after=params, pred=pred0, pred_after=pred1, state=[])
    return compare(ref, got)
