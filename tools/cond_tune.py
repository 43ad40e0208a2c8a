"""In-situ timing of every (split, tile) candidate of every GEMM problem of the frozen conditioners (VAE encode of T + 1 frames, CLIP embed):
ops.GemmTuner inside real passes, as tools/tune_dump.py does for the UNet step.  Prints, per problem, the candidates fastest first with the cost
model's choice marked, and the sum over problems of (model's choice - fastest) -- what a table for these shapes would be worth.

    python tools/cond_tune.py [--rounds 2] [--json out.json]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    import bench
    from svd_xtend_amd import ops
    from svd_xtend_amd.clip import CLIPVisionModelWithProjection, encode_image
    from svd_xtend_amd.vae import AutoencoderKLTemporalDecoder
    dev = torch.device("cuda", 0)
    dt = torch.float16
    with torch.device(dev):
        vae, enc = AutoencoderKLTemporalDecoder(), CLIPVisionModelWithProjection()
    bench.init_weights_(vae, seed=4321)
    bench.init_weights_(enc, seed=4322)
    for m in (vae, enc):
        m.requires_grad_(False)
        m.prepare(dt)
    pix = (torch.rand(1, 14, 3, 320, 512, device=dev) * 2 - 1)
    frames = torch.cat([pix, pix[:, 0:1]], dim=1).reshape(15, 3, 320, 512)
    out = {}
    with torch.no_grad():
        for name, model, run in (("vae", vae, lambda: vae.encode(frames).latent_dist), ("clip", enc, lambda: encode_image(pix[:, 0], enc))):
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            model.rt.tuner = t = ops.GemmTuner(args.rounds)
            n = 0
            while n < 400:
                run()
                n += 1
                if t.end_step():
                    break
            if t.active:
                t.freeze()
            print(f"# {name}: {n} passes, {len(t.cands)} problems")
            tot_model = tot_best = 0.0
            rows_out = []
            for key, cands in t.cands.items():
                st = t.stats[key]
                rows = sorted(((s[0] / s[1] * 1e3, c, s[1]) for s, c in zip(st, cands) if s[1]), key=lambda r: r[0])
                if not rows:
                    continue
                calls = max(1, rows[0][2] // max(1, args.rounds))
                model.rt.tuner = None
                choice = None
                if key[0] == "nt":
                    M, N, Kd, ldc = key[1], key[2], key[3], key[5]
                    g = key[6]
                    choice = ops.choose_cfg(model.rt, M, N, Kd, ldc, g[3] if isinstance(g, tuple) else 0, False)
                model.rt.tuner = t
                us_model = next((us for us, c, _ in rows if c == choice), None)
                if us_model is not None:
                    tot_model += us_model * calls
                    tot_best += rows[0][0] * calls
                rows_out.append((rows[0][0] * calls, key, calls, choice, us_model, rows))
            for _, key, calls, choice, us_model, rows in sorted(rows_out, key=lambda r: -r[0]):
                print(f"{str(key)[:100]:100s} x{calls:3d} | model {choice}: {us_model if us_model is None else round(us_model, 1)} | " +
                      "  ".join(f"{c}: {us:.1f}" for us, c, _ in rows[:6]))
            print(f"# {name}: cost model's choices {tot_model / 1e3:.3f} ms per pass, fastest candidates {tot_best / 1e3:.3f} ms")
            out[name] = {str(k): v for k, v in t.table.items()}
            # what the frozen table gives end to end
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for label, tuner in (("tuned", t), ("model", None), ("tuned", t), ("model", None)):
                model.rt.tuner = tuner
                run()
                e0.record()
                for _ in range(5):
                    run()
                e1.record()
                torch.cuda.synchronize()
                print(f"# {name} [{label}]: {e0.elapsed_time(e1) / 5:.3f} ms per pass")
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
