"""Launches per C-ABI entry (and per call site) of ONE steady-state optimizer step, counted on the CPU emulation of the kernels
(tests/emul.py) with the tiny topology -- same block structure and call graph as the full model, so a launch that should not be there
shows up without a GPU.  (It found config 5 re-casting its frozen feed-forward weights after every step: 128 launches.)

    python tools/launch_audit.py [--lora-rank 64] [--sites small_linear,outer_acc]   # TEST-SIDE tool: imports tests/emul.py
"""
import argparse
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lora-rank", type=int, default=0)
    ap.add_argument("--sites", default="", help="comma-separated entries whose call sites are listed")
    args = ap.parse_args()
    import emul
    from svd_xtend_amd import kernels as K
    from svd_xtend_amd.train import Trainer
    be = emul.EmuBackend()
    cnt, sites = collections.Counter(), collections.Counter()
    on, nested = [False], [0]
    want = set(filter(None, args.sites.split(",")))

    def wrap(name, f):
        def w(*a, **kw):
            if on[0] and nested[0] == 0:          # the emulation of a *_batch entry loops over the single entries: count the entry once
                cnt[name] += 1
                if name in want:
                    st = traceback.extract_stack(limit=5)[:-1]
                    sites[(name,) + tuple(f"{s.name}:{s.lineno}" for s in st[-3:])] += 1
            batch = name.endswith("_batch")
            nested[0] += batch
            try:
                return f(*a, **kw)
            finally:
                nested[0] -= batch
        return w
    for name in dir(be):
        if not name.startswith("_") and callable(getattr(be, name)):
            setattr(be, name, wrap(name, getattr(be, name)))
    K._set_backend_for_tests(be)
    import e2e_checks
    orig, n = Trainer.step, [0]

    def step(self, *a, **kw):
        n[0] += 1
        on[0] = n[0] == 2                         # the second step: packing and caches are behind us
        try:
            return orig(self, *a, **kw)
        finally:
            on[0] = False
    Trainer.step = step
    e2e_checks.run_steps(dev=torch.device("cpu"), dtype=torch.float32, steps=2, lora_r=args.lora_rank)
    for k, v in cnt.most_common():
        print(f"{v:6d}  {k}")
    print(f"{sum(cnt.values()):6d}  total")
    for k, v in sites.most_common():
        print(f"{v:6d}  {' <- '.join(k)}")


if __name__ == "__main__":
    main()
