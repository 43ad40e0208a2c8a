"""CLIP ViT-H image tower alone (the `encode_image` of train_svd.py:857-876 on one 512x320 frame), random weights: ms per call.
usage: python tools/clip_bench.py [--iters 20]      (under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svd_xtend_amd.clip import CLIPVisionModelWithProjection, encode_image  # noqa: E402
from bench import init_weights_  # noqa: E402

if __name__ == "__main__":
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 20
    dev = torch.device("cuda:0")
    with torch.device(dev):
        enc = CLIPVisionModelWithProjection()
    init_weights_(enc, seed=4322)
    enc.requires_grad_(False)
    enc.prepare(torch.float16)
    pix = (torch.rand(1, 3, 320, 512, device=dev) * 2 - 1)
    for _ in range(3):
        encode_image(pix, enc)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        encode_image(pix, enc)
    torch.cuda.synchronize()
    print(f"encode_image: {(time.perf_counter() - t) / iters * 1e3:.3f} ms per call (eager launches, {iters} calls)")
