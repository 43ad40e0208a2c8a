"""Golden vectors for oracle/clip_image.py from the reference's OWN code, run in this container.

/root/reference/train_svd.py cannot be imported (its module level imports diffusers, accelerate, cv2 ..., none installed), but the
anti-aliased resize it feeds CLIP with (`_resize_with_antialiasing`, `_compute_padding`, `_filter2d`, `_gaussian`,
`_gaussian_blur2d`, train_svd.py:140-248) is plain torch, and so is the noise-level sampler `rand_log_normal` (:63-66).  This script parses the file, executes ONLY those six function
definitions in a namespace holding `torch`, runs them on seeded inputs and stores inputs and outputs.  Nothing of the reference's
source is written to the repo.  Usage (needs /root/reference, i.e. this container):  python tests/golden/make_golden_resize.py
"""
import ast
import os

import torch
from safetensors.torch import save_file

REF = "/root/reference/train_svd.py"
WANT = {"_resize_with_antialiasing", "_compute_padding", "_filter2d", "_gaussian", "_gaussian_blur2d", "rand_log_normal"}
# (name, batch, height, width, target size): the c2 frame, an up-scaling case (sigma clamps to 0.001, 3 taps), odd sizes / batch 2,
# and a strongly anisotropic one (different taps per axis)
CASES = [("c2_frame_320x512_to_224", 1, 320, 512, (224, 224)), ("upscale_64x96_to_80x120", 1, 64, 96, (80, 120)),
         ("odd_101x75_to_32x24", 2, 101, 75, (32, 24)), ("aniso_400x90_to_50x60", 1, 400, 90, (50, 60))]


SIGMA_CASES = [(-3.0, 0.5), (0.7, 1.6), (0.0, 1.0)]      # cond_sigmas (:954), sigmas (:964), the defaults


def golden_input(seed, b, h, w):
    """Frames in [-1, 1] on the CPU generator (deterministic across runs; the test re-creates them instead of storing 3 MB)."""
    return torch.rand(b, 3, h, w, generator=torch.Generator().manual_seed(seed)) * 2 - 1


def reference_functions():
    tree = ast.parse(open(REF).read())
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT]
    assert {d.name for d in defs} == WANT
    ns = {"torch": torch}
    exec(compile(ast.Module(body=defs, type_ignores=[]), REF, "exec"), ns)
    return ns


def main():
    fn = reference_functions()["_resize_with_antialiasing"]
    out = {}
    for i, (name, b, h, w, size) in enumerate(CASES):
        x = golden_input(100 + i, b, h, w)
        y = fn(x, size)                                                  # interpolation="bicubic", align_corners=True (defaults)
        out[f"{name}.out"] = y.contiguous()
        out[f"{name}.meta"] = torch.tensor([100 + i, b, h, w, size[0], size[1]])     # the input is regenerated from its seed
    # sigma sampling of the training loop (train_svd.py:954, :964) on the global CPU generator
    rln = reference_functions()["rand_log_normal"]
    for j, (loc, scale) in enumerate(SIGMA_CASES):
        torch.manual_seed(200 + j)
        out[f"rand_log_normal.{j}"] = rln(shape=[8], loc=loc, scale=scale)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize_antialias.safetensors")
    save_file(out, path)
    print("wrote", path, {k: tuple(v.shape) for k, v in out.items() if k.endswith(".out")})


if __name__ == "__main__":
    main()
