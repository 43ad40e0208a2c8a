"""One CPU-oracle step of the FULL topology at c1' (8 frames 256x192) at a given torch thread count (run on the GPU box's host):
    python tools/oracle_threads_c1.py 16 32 64"""
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
    torch.set_num_threads(int(sys.argv[2]))
    t = time.time()
    e2e_checks.oracle_step(SVD_CONFIG, 1, 8, 24, 32, seed=0, lr=1e-4, cross_dim=1024, with_pred_after=False)
    print(f"c1' threads {sys.argv[2]}: {time.time() - t:.1f} s", flush=True)
else:
    for n in sys.argv[1:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", n], env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
