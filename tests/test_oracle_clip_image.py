"""oracle/clip_image.py (groundwork for SURVEY.md 8(f) rank 2) against golden outputs of the reference's own resize functions
(tests/golden/make_golden_resize.py) and against transformers' CLIP vision tower, which the reference itself calls."""
import os
import sys

import torch
from safetensors.torch import load_file

from oracle.clip_image import CLIP_MEAN, CLIP_STD, blur_taps, clip_pixel_values, encode_image, resize_with_antialiasing

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def test_resize_matches_reference_outputs():
    from make_golden_resize import CASES, golden_input
    g = load_file(os.path.join(HERE, "golden", "resize_antialias.safetensors"))
    assert len(CASES) == 4
    for name, *_ in CASES:
        seed, b, h, w, s0, s1 = g[f"{name}.meta"].tolist()
        y = resize_with_antialiasing(golden_input(seed, b, h, w), (s0, s1))
        ref = g[f"{name}.out"]
        assert y.shape == ref.shape
        assert float((y - ref).abs().max()) <= 1e-6, name            # same torch ops in the same order: bit-equal in practice


def test_rand_log_normal_matches_reference_outputs():
    """oracle.step.rand_log_normal (the sigma sampler of train_svd.py:63-66, used at :954 / :964) against the reference's function."""
    from make_golden_resize import SIGMA_CASES
    from oracle.step import rand_log_normal
    g = load_file(os.path.join(HERE, "golden", "resize_antialias.safetensors"))
    for j, (loc, scale) in enumerate(SIGMA_CASES):
        torch.manual_seed(200 + j)
        assert torch.equal(rand_log_normal([8], loc=loc, scale=scale), g[f"rand_log_normal.{j}"]), (loc, scale)


def test_blur_parameters():
    assert blur_taps(320 / 224) == (3, (320 / 224 - 1) / 2)           # c2 height: sigma 0.214, minimum window
    assert blur_taps(512 / 224) == (3, (512 / 224 - 1) / 2)           # c2 width: sigma 0.643 -> int(2.57) = 2 -> max(.., 3)
    assert blur_taps(1024 / 224)[0] == 7 and blur_taps(0.5) == (3, 0.001)      # c4 width; up-scaling clamps sigma
    assert blur_taps(4.0) == (7, 1.5)                                 # int(6.0) = 6 -> made odd


def test_clip_input_normalisation_and_tower():
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.manual_seed(0)
    frames = torch.rand(2, 3, 96, 160) * 2 - 1
    px = clip_pixel_values(frames)
    assert px.shape == (2, 3, 224, 224)
    # the reference hands the resized [0, 1] image to CLIPImageProcessor with only do_normalize on (train_svd.py:864-871)
    fe = CLIPImageProcessor()
    assert tuple(fe.image_mean) == CLIP_MEAN and tuple(fe.image_std) == CLIP_STD
    x01 = (resize_with_antialiasing(frames, (224, 224)) + 1.0) / 2.0
    ref = fe(images=x01, do_normalize=True, do_center_crop=False, do_resize=False, do_rescale=False, return_tensors="pt").pixel_values
    assert float((px - ref).abs().max()) <= 1e-5
    cfg = CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, image_size=224,
                           patch_size=14, projection_dim=32)
    tower = CLIPVisionModelWithProjection(cfg).eval()
    with torch.no_grad():
        e = encode_image(frames, tower)
    assert e.shape == (2, 32) and torch.isfinite(e).all()
