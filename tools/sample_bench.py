"""Wall time of the validation sampler at the reference's settings (train_svd.py:1124-1133: 512x320, 14 frames, 25 Euler steps with
classifier-free guidance, decode_chunk_size 8) on random-init full-size modules (no checkpoints offline): CLIP ViT-H/14 embedding +
VAE encode of the conditioning frame, 25 UNet forwards at batch 2, temporal VAE decoder.  One JSON line.
    python tools/sample_bench.py [--height 320 --width 512 --frames 14 --steps 25 --dtype fp16]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svd_xtend_amd.clip import CLIPVisionModelWithProjection  # noqa: E402
from svd_xtend_amd.pipeline import StableVideoDiffusionPipeline  # noqa: E402
from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel  # noqa: E402
from svd_xtend_amd.vae import AutoencoderKLTemporalDecoder  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--dtype", default="fp16")
    a = ap.parse_args()
    dt = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda")
    torch.manual_seed(0)
    t0 = time.time()
    with torch.device("cpu"):
        unet = UNetSpatioTemporalConditionModel(num_frames=a.frames)
        vae = AutoencoderKLTemporalDecoder()
        clip = CLIPVisionModelWithProjection()
    for m in (unet, vae, clip):
        m.requires_grad_(False)
        m.to(dev)
        m.prepare(dt)
    build_s = time.time() - t0
    pipe = StableVideoDiffusionPipeline(vae, clip, unet)
    img = torch.rand(1, 3, a.height, a.width)
    kw = dict(height=a.height, width=a.width, num_frames=a.frames, decode_chunk_size=8, motion_bucket_id=127, fps=7,
              noise_aug_strength=0.02, output_type="pt")
    pipe(img, num_inference_steps=2, **kw)                      # warm-up: allocator, lazy module state
    torch.cuda.synchronize()
    t0 = time.time()
    frames = pipe(img, num_inference_steps=a.steps, generator=torch.Generator().manual_seed(1), **kw).frames
    torch.cuda.synchronize()
    total = time.time() - t0
    # the parts, timed separately
    def timed(fn):
        torch.cuda.synchronize()
        t = time.time()
        fn()
        torch.cuda.synchronize()
        return time.time() - t
    lat = torch.randn(2, a.frames, 8, a.height // 8, a.width // 8, device=dev)
    emb, ids = torch.zeros(2, 1, 1024, device=dev), torch.tensor([[6.0, 127.0, 0.02]] * 2, device=dev)
    unet_s = timed(lambda: [unet(lat, torch.tensor(1.0), emb, ids) for _ in range(5)]) / 5
    z = torch.randn(a.frames, 4, a.height // 8, a.width // 8, device=dev)
    dec_s = timed(lambda: [vae.decode(z[i:i + 8], num_frames=z[i:i + 8].shape[0]) for i in range(0, a.frames, 8)])
    print(json.dumps({"what": "validation sampler (train_svd.py:1124-1133 settings)", "height": a.height, "width": a.width, "frames": a.frames,
                      "steps": a.steps, "dtype": a.dtype, "seconds_total": total, "unet_forward_batch2_ms": unet_s * 1e3,
                      "vae_decode_all_frames_ms": dec_s * 1e3, "finite": bool(torch.isfinite(frames).all()),
                      "frames_shape": list(frames.shape), "model_build_s": build_s, "data": "random-init weights, random image"}))


if __name__ == "__main__":
    main()
