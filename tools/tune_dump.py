"""In-situ timing of every (split, tile) candidate of every NT GEMM problem of the benched step (ops.GemmTuner: each candidate is timed with
events INSIDE real forward + backward sweeps -- cold caches, real neighbours), printed per problem, fastest first.

    python tools/tune_dump.py [--rounds 2] [--only 32,34]      (--only: print problems where one of these tile variants is a candidate)
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--only", default="")
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--latent", default="40x64", help="latent height x width (c2: 40x64, c4: 72x128)")
    ap.add_argument("--diff", action="store_true", help="print only the problems where the cost model's choice is not the fastest, with what the table would win")
    args = ap.parse_args()
    import bench
    from svd_xtend_amd.train import Trainer
    from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel
    dev = torch.device("cuda", 0)
    with torch.device(dev):
        model = UNetSpatioTemporalConditionModel()
    bench.init_weights_(model, seed=1234)
    tr = Trainer(model, dtype=torch.float16, lr=1e-5)
    lh, lw = (int(v) for v in args.latent.split("x"))
    batch = bench.make_batch(1, args.frames, lh, lw, model.config.cross_attention_dim, seed=123, dev=dev)
    for _ in range(2):
        tr.step(batch)
    n = tr.tune_gemms(batch, rounds=args.rounds, max_steps=600)
    t = tr.rt.tuner
    only = {int(v) for v in args.only.split(",") if v}
    print(f"# {n} sweeps, {len(t.cands)} problems")
    tot_best = tot_model = 0.0
    from svd_xtend_amd import ops
    for key, cands in sorted(t.cands.items(), key=lambda kv: -min((s[0] / s[1]) * s[1] for s in t.stats[kv[0]] if s[1])):
        st = t.stats[key]
        rows = sorted(((s[0] / s[1] * 1e3, c, s[1]) for s, c in zip(st, cands) if s[1]), key=lambda r: r[0])
        if not rows:
            continue
        calls = rows[0][2] // max(1, args.rounds)
        if only and not any((c[1] if isinstance(c, tuple) else c) in only for _, c, _ in rows):
            continue
        if args.diff:
            choice = None
            if key[0] == "nt" and len(key) == 11:          # no second operand pair
                g = key[6]
                t.active, saved = False, tr.rt.tuner
                tr.rt.tuner = None
                choice = ops.choose_cfg(tr.rt, key[1], key[2], key[3], key[5], g[3] if isinstance(g, tuple) else 0, False)
                tr.rt.tuner = saved
            us_model = next((us for us, c, _ in rows if c == choice), None)
            if us_model is None or us_model <= rows[0][0] * 1.02:
                continue
            tot_model += (us_model - rows[0][0]) * calls
            print(f"{str(key)[:100]:100s} x{calls:3d} | model {choice}: {us_model:.1f}  best {rows[0][1]}: {rows[0][0]:.1f}  -> {(us_model - rows[0][0]) * calls / 1e3:.3f} ms / sweep")
            continue
        print(f"{str(key)[:110]:110s} x{calls:3d} | " + "  ".join(f"{c}: {us:.1f}" for us, c, _ in rows[:7]))
    if args.diff:
        print(f"# cost model's choices behind the fastest measured candidates by {tot_model / 1e3:.3f} ms per forward + backward sweep (problems > 2 % apart)")


if __name__ == "__main__":
    main()
