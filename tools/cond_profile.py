"""The frozen conditioners of one clip (VAE encode of T + 1 frames, CLIP embed of the first) as bench.py's real_loop runs them, a few times in a
row: for `rocprofv3 --kernel-trace --stats -- python tools/cond_profile.py`, and (--table) the per-problem GEMM table of one pass."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from svd_xtend_amd import kernels as K
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
    k = K.backend()
    with torch.no_grad():
        for _ in range(2):
            vae.encode(frames).latent_dist
            encode_image(pix[:, 0], enc)
        torch.cuda.synchronize()
        if "--table" in sys.argv:
            k.launch_log = log = []
            vae.encode(frames).latent_dist
            k.launch_log = None
            from collections import Counter
            c = Counter()
            for name, a, extra in log:
                key = (name,) + (tuple(a[3:6]) if name.startswith("svdx_gemm") and name not in ("svdx_gemm_finalize", "svdx_gemm_finalize_gn") else
                                 tuple(x for x in a if isinstance(x, int) and x < (1 << 32))[:4])
                c[key] += 1
            for key, n in sorted(c.items(), key=lambda kv: -kv[1]):
                print(n, key)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            vae.encode(frames).latent_dist
        e1.record()
        torch.cuda.synchronize()
        print(f"vae.encode x15 frames: {e0.elapsed_time(e1) / 5:.2f} ms")
        e0.record()
        for _ in range(5):
            encode_image(pix[:, 0], enc)
        e1.record()
        torch.cuda.synchronize()
        print(f"clip embed: {e0.elapsed_time(e1) / 5:.2f} ms")


if __name__ == "__main__":
    main()
