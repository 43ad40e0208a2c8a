"""One-shot GPU diagnostics: runs every per-kernel parity check and the end-to-end tiny train step, writes
gpurun_out/kernel_report.json and prints a summary.  Usage (on the GPU box): python tools/gpu_report.py [--quick]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402


def main():
    import kernel_checks as kc
    from svd_xtend_amd import kernels as K
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    dev = torch.device("cuda")
    print("device:", torch.cuda.get_device_name(0), flush=True)
    be = K.backend()
    t0 = time.time()
    res = kc.run_all(be, dev)
    bad = [r for r in res if not r["ok"]]
    print(f"kernel checks: {len(res)} total, {len(bad)} failed, {time.time() - t0:.1f}s", flush=True)
    by_group = {}
    for r in res:
        g = by_group.setdefault((r["group"], r["dtype"]), [0, 0, 0.0])
        g[0] += 1
        g[1] += 0 if r["ok"] else 1
        if r["err"] == r["err"] and r["err"] != float("inf"):
            g[2] = max(g[2], r["err"])
    for (g, dt), (n, nb, worst) in sorted(by_group.items()):
        print(f"  {g:22s} {dt:16s} n={n:3d} failed={nb:3d} worst_err={worst:.3e}")
    with open(os.path.join(out_dir, "kernel_report.json"), "w") as f:
        json.dump(res, f, indent=1)
    e2e = {}
    try:
        import e2e_checks
        e2e = e2e_checks.run_all(verbose=True)
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        e2e = {"error": repr(e)}
    with open(os.path.join(out_dir, "e2e_report.json"), "w") as f:
        json.dump(e2e, f, indent=1, default=str)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
