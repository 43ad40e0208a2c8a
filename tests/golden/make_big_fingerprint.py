"""Fingerprints of the cached big oracle references (tests/golden/_big/*.pt: git-ignored, 145 MB each) -- loss, the fingerprint of the seeded
weights, the L2 norm of the prediction and of every gradient -- committed as tests/golden/big_ref_fingerprints.json, so that a cached
reference (or a recomputed one) can be held to what this repository recorded: `e2e_checks.oracle_step_cached` checks it on load.

    SVDX_SAVE_BIG_REF=1 python tests/golden/make_big_refs.py L0      # ~17 min on 8 host threads, ~20 GB
    python tests/golden/make_big_fingerprint.py
"""
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
BIG = os.path.join(HERE, "_big")
OUT = os.path.join(HERE, "big_ref_fingerprints.json")


def fingerprint(ref) -> dict:
    return dict(loss=float(ref["loss"]), sd0_fingerprint=float(ref["sd0_fingerprint"]), lr=float(ref["lr"]),
                pred_l2=float(ref["pred"].double().norm()), n_grads=len(ref["grads"]),
                grad_l2={k: float(v.double().norm()) for k, v in sorted(ref["grads"].items())})


def main():
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for f in sorted(os.listdir(BIG)):
        if f.endswith(".pt"):
            out[f[:-3]] = fingerprint(torch.load(os.path.join(BIG, f), weights_only=False))
            print(f, "loss", out[f[:-3]]["loss"], "gradients", out[f[:-3]]["n_grads"])
    json.dump(out, open(OUT, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
