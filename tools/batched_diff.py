"""What exactly differs on the GPU between the table-driven skinny launches (Runtime.batch_small) and one launch each?
The simulator holds the two forms to bit equality; the GPU test holds them to rounding level.  This prints, per dtype / trainable set:
loss equality, and for the flat weights / Adam moments after two optimizer steps the number of differing elements, the largest
difference in units of the last place, and the parameters they sit in.
    python tools/batched_diff.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_checks  # noqa: E402


def ulps(a, b):
    ia, ib = a.view(torch.int32).long(), b.view(torch.int32).long()
    ia = torch.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = torch.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return (ia - ib).abs()


def main():
    for dt, r in ((torch.float16, 0), (torch.bfloat16, 0), (torch.bfloat16, 8)):
        a, b = e2e_checks.batched_vs_single_small_launches(dtype=dt, lora_r=r)
        print(f"== {dt} lora_r={r}: loss batched {a['loss']!r} single {b['loss']!r} equal={a['loss'] == b['loss']}; launches {len(a['launches'])} vs {len(b['launches'])}")
        for k in ("p", "m", "v"):
            x, y = a[k].float().cpu(), b[k].float().cpu()
            ne = (x != y)
            n = int(ne.sum())
            if not n:
                print(f"   {k}: bit-identical ({x.numel()} elements)")
                continue
            u = ulps(x, y)
            print(f"   {k}: {n} of {x.numel()} elements differ, max {int(u.max())} ulp, max |diff| {float((x - y).abs().max()):.3e}")
            idx = ne.nonzero().flatten()
            per = {}
            for off, numel, name in a["layout"]:
                c = int(((idx >= off) & (idx < off + numel)).sum())
                if c:
                    per[name] = (c, numel, int(u[off:off + numel].max()))
            for name, (c, numel, mu) in sorted(per.items(), key=lambda t: -t[1][0])[:12]:
                print(f"      {name}: {c} / {numel} differ, max {mu} ulp")
            if len(per) > 12:
                print(f"      ... and {len(per) - 12} more parameters")


if __name__ == "__main__":
    main()
