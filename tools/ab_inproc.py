"""Same-box, same-process A/B of developer knobs inside the real train step.

    python tools/ab_inproc.py [bench flags: --dtype bf16 --lora-rank 64 --frames 25 ...] -- CFG [CFG ...]

A CFG is a comma-separated list of settings applied on top of the defaults, `base` for none:
    batch_small=0              Runtime attribute (ops.Runtime) -- int / bool / float literal
    SVDX_X=1                   environment variable (none selects a kernel since round 6; kept for one-off experiments)
    tuned                      the in-situ GEMM tuner's table (Trainer.tune_gemms, run once, staged candidates included)
The model is built ONCE; every CFG gets its own hipGraph of the step (captured after two eager steps under that setting), then
the graphs are replayed alternately, `--reps` times `--steps` steps each: box-to-box spread (+-5 %) and the cost of a fresh
process per measurement (~40 s of a 90-minute GPU budget) both drop out.  Prints one line per (rep, CFG) and a summary."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_val(v):
    for cast in (int, float):
        try:
            return cast(v)
        except ValueError:
            pass
    return v


def main():
    argv = sys.argv[1:]
    cfgs = ["base"]
    if "--" in argv:
        i = argv.index("--")
        argv, cfgs = argv[:i], argv[i + 1:]
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--lora-rank", type=int, default=0)
    ap.add_argument("--tune-rounds", type=int, default=1)
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)

    import bench
    from svd_xtend_amd import ops
    from svd_xtend_amd.train import GraphedStep, Trainer
    from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    with torch.device(dev):
        model = UNetSpatioTemporalConditionModel()
    bench.init_weights_(model, seed=1234)
    if args.lora_rank:
        from svd_xtend_amd.lora import LoraConfig
        for p in model.parameters():
            p.requires_grad_(False)
        with torch.device(dev):
            model.add_adapter(LoraConfig(r=args.lora_rank, lora_alpha=args.lora_rank, init_lora_weights="gaussian"))
    tr = Trainer(model, dtype=dt, lr=1e-5)
    batch = bench.make_batch(1, args.frames, args.height // 8, args.width // 8, model.config.cross_attention_dim, seed=123, dev=dev)
    rt = tr.rt
    defaults_env = dict(os.environ)
    tuner = None

    def apply(cfg):
        """-> undo()"""
        nonlocal tuner
        saved_attr, saved_env = {}, {}
        rt.tuner = None
        for kv in ([] if cfg == "base" else cfg.split(",")):
            if kv == "tuned":
                if tuner is None:
                    os.environ["SVDX_STAGED"] = "1"
                    t0 = time.time()
                    n = tr.tune_gemms(batch, rounds=args.tune_rounds, max_steps=400)
                    tuner = rt.tuner
                    print(f"# tuned in {n} sweeps, {time.time() - t0:.1f} s", flush=True)
                rt.tuner = tuner
                continue
            k, v = kv.split("=", 1)
            if k.isupper():
                saved_env[k] = os.environ.get(k)
                os.environ[k] = v
            else:
                if not hasattr(rt, k):
                    raise SystemExit(f"Runtime has no attribute {k!r}")
                saved_attr[k] = getattr(rt, k)
                setattr(rt, k, parse_val(v))
        for f in (getattr(ops, n, None) for n in dir(ops)):          # memoised choices must not leak across settings
            if hasattr(f, "cache_clear"):
                f.cache_clear()

        def undo():
            for k, v in saved_attr.items():
                setattr(rt, k, v)
            for k, v in saved_env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            rt.tuner = None
        return undo

    graphs = {}
    for cfg in cfgs:
        undo = apply(cfg)
        for _ in range(2):
            tr.step(batch)
        torch.cuda.synchronize()
        graphs[cfg] = GraphedStep(tr, batch)
        graphs[cfg]()
        torch.cuda.synchronize()
        undo()
        # NB: no torch kernel may be launched between replays here (round 4 found that an out-of-place ATen op after a replay leaves
        # every later replay of the graph with a non-finite loss on this runtime -- profiles/r4_graph_replay_prime.txt); the loss slot is
        # read with a plain D2H copy
        print(f"# captured {cfg}: loss {float(tr.loss_slot.cpu()):.6f}", flush=True)

    res = {c: [] for c in cfgs}
    for c in cfgs:                       # settle clocks: the first replays after capture run warm-up fast
        for _ in range(10):
            graphs[c]()
    torch.cuda.synchronize()
    for rep in range(args.reps):
        for c in cfgs:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                graphs[c]()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            res[c].append(ms)
            print(f"[{c}] rep {rep}: {ms:.3f} ms/step", flush=True)
    st = tr.opt_state.cpu().tolist()
    lf = float(tr.loss_slot.cpu())
    print(f"# after the timed replays: loss {lf:.6f}, optimizer steps {st[0]:.0f}, loss scale {st[1]:g}" + ("" if lf == lf and abs(lf) < 1e30 else "   <-- NON-FINITE: the timings above are void"))
    base = min(res[cfgs[0]])
    print("# summary (min / median over reps; delta of medians against the first CFG)")
    med0 = sorted(res[cfgs[0]])[len(res[cfgs[0]]) // 2]
    for c in cfgs:
        v = sorted(res[c])
        print(f"  {c:48s} min {v[0]:.3f}  med {v[len(v) // 2]:.3f}  delta {v[len(v) // 2] - med0:+.3f} ms")
    if args.out:
        out = {"cfgs": res}
        if tuner is not None:
            out["tuned"] = [[repr(kk), repr(tuner.table.get(kk)),
                             [[list(c) if isinstance(c, tuple) else c, (st[0] / st[1]) if st[1] else None] for c, st in zip(tuner.cands[kk], tuner.stats[kk])]]
                            for kk in tuner.cands]
        with open(args.out, "w") as f:
            json.dump(out, f)
    del base


if __name__ == "__main__":
    main()
