"""How long one CPU-oracle step takes at a given torch thread count (run on the GPU box's host; profiles/r6b_oracle_threads.txt,
r6c_oracle_threads_c1.txt):
    python tools/oracle_threads.py [--c1] 16 32 64 128        default: the 64x40-level block (35840 rows x 320 channels); --c1: the full topology at 8 x 256x192"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import time

    import torch
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import e2e_checks
    from oracle.unet import SVD_CONFIG
    what, n = sys.argv[2], int(sys.argv[3])
    torch.set_num_threads(n)
    t = time.time()
    if what == "c1":
        e2e_checks.oracle_step(SVD_CONFIG, 1, 8, 24, 32, seed=0, lr=1e-4, cross_dim=1024, with_pred_after=False)
    else:
        e2e_checks.oracle_step(e2e_checks.level_config(320, 5), 1, 14, 40, 64, seed=11, lr=1e-4, cross_dim=1024)
    print(f"{what} threads {n}: {time.time() - t:.1f} s", flush=True)
else:
    what = "c1" if "--c1" in sys.argv else "L0"
    for n in [a for a in sys.argv[1:] if a.isdigit()]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", what, n], env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
