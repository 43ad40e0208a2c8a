"""How long one CPU-oracle step of the 64x40-level block takes at a given torch thread count (run on the GPU box's host):
    python tools/oracle_threads.py 32 64 128"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import time

    import torch
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import e2e_checks
    torch.set_num_threads(int(sys.argv[2]))
    cfg = e2e_checks.level_config(320, 5)
    t = time.time()
    e2e_checks.oracle_step(cfg, 1, 14, 40, 64, seed=11, lr=1e-4, cross_dim=1024)
    print(f"threads {sys.argv[2]}: {time.time() - t:.1f} s", flush=True)
else:
    for n in sys.argv[1:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", n], env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
