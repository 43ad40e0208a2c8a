"""SURVEY.md 8(f) rank 4, second half: the validation sampler (/root/reference/train_svd.py:1093-1150, infer_svd.ipynb cell 3) --
svd_xtend_amd/pipeline.py against oracle/sampler.py (diffusers' StableVideoDiffusionPipeline + EulerDiscreteScheduler restated;
parity unpinned, see the oracle's header).  CPU tests drive the host orchestration over the fp32 emulation of the C-ABI; `-m gpu`
tests run the kernels."""
import math

import pytest
import torch

from oracle import sampler as O
from oracle.unet import TINY_CONFIG, UNetSpatioTemporalConditionOracle, scaled_init_
from svd_xtend_amd.pipeline import EulerDiscreteScheduler, StableVideoDiffusionPipeline, tensor2vid
from svd_xtend_amd.unet import UNetSpatioTemporalConditionModel

from test_clip import SMALL as CLIP_SMALL, make_pair as make_clip_pair
from test_vae import SMALL as VAE_SMALL, make_pair as make_vae_pair

gpu = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def test_scheduler_is_svds_euler_configuration():
    sch = EulerDiscreteScheduler()
    sch.set_timesteps(25)
    sig = O.karras_sigmas(25)
    assert torch.equal(sch.sigmas, sig) and sig.shape == (26,) and float(sig[-1]) == 0.0
    assert abs(float(sig[0]) - 700.0) < 1e-3 and abs(float(sig[24]) - 0.002) < 1e-7
    assert torch.allclose(sch.timesteps, 0.25 * sig[:-1].log())
    assert abs(sch.init_noise_sigma - math.sqrt(700.0 ** 2 + 1)) < 1e-3                    # "leading" spacing
    g = torch.Generator().manual_seed(0)
    x, v = torch.randn(2, 3, 4, 5, 5, generator=g), torch.randn(2, 3, 4, 5, 5, generator=g)
    for i in (0, 1):
        s = float(sig[i])
        assert torch.allclose(sch.scale_model_input(x), x / (s * s + 1) ** 0.5)
        got = sch.step(v, sch.timesteps[i], x).prev_sample
        assert torch.allclose(got, O.euler_step_v(x, v, s, float(sig[i + 1])), rtol=1e-6, atol=1e-6)
    # the last step lands on sigma = 0: the sample becomes the predicted clean latent
    sch.set_timesteps(2)
    sch._step_index = 1
    out = sch.step(v, None, x)
    assert torch.allclose(out.prev_sample, out.pred_original_sample, atol=1e-6)
    with pytest.raises(NotImplementedError):
        EulerDiscreteScheduler(prediction_type="epsilon")


def make_models(seed, dev="cpu"):
    orc_unet = UNetSpatioTemporalConditionOracle(**TINY_CONFIG).eval()
    scaled_init_(orc_unet, seed=seed)
    unet = UNetSpatioTemporalConditionModel(**TINY_CONFIG)
    unet.load_state_dict(orc_unet.state_dict(), strict=True)
    unet.requires_grad_(False)
    orc_vae, vae = make_vae_pair(VAE_SMALL, seed + 1, dev)
    ref_clip, clip = make_clip_pair(CLIP_SMALL, seed + 2, dev)
    return (orc_unet, orc_vae, ref_clip), (unet.to(dev), vae, clip)


def test_pipeline_matches_the_oracle_sampler_on_the_emulated_kernels(emu_backend):
    (ou, ov, oc), (unet, vae, clip) = make_models(21)
    for m in (unet, vae, clip):
        m.prepare(torch.float32)
    pipe = StableVideoDiffusionPipeline(vae, clip, unet)
    img = torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(22))
    kw = dict(num_frames=3, num_inference_steps=3, decode_chunk_size=2, motion_bucket_id=127, fps=7, noise_aug_strength=0.02)
    want = O.svd_sample(img, ou, ov, oc, generator=torch.Generator().manual_seed(23), **kw)
    got = pipe(img, height=128, width=128, generator=torch.Generator().manual_seed(23), output_type="pt", **kw).frames
    assert got.shape == (1, 3, 3, 128, 128)                                                # [b, f, 3, H, W]
    want01 = (want.permute(0, 2, 1, 3, 4) / 2 + 0.5).clamp(0, 1)
    assert rel(got, want01) <= 1e-4, rel(got, want01)
    # latents before decoding, without guidance (one UNet row per clip), given start noise
    lat0 = torch.randn(1, 3, 4, 16, 16, generator=torch.Generator().manual_seed(24))
    kw2 = dict(kw, max_guidance_scale=1.0)
    want_l = O.svd_sample(img, ou, ov, oc, generator=torch.Generator().manual_seed(25), latents=lat0, output_latents=True, **kw2)
    got_l = pipe(img, height=128, width=128, generator=torch.Generator().manual_seed(25), latents=lat0, output_type="latent", **kw2).frames
    assert rel(got_l, want_l) <= 1e-4, rel(got_l, want_l)
    np_frames = pipe(img, height=128, width=128, generator=torch.Generator().manual_seed(23), output_type="np", **kw).frames
    assert np_frames.shape == (1, 3, 128, 128, 3) and 0.0 <= np_frames.min() and np_frames.max() <= 1.0
    with pytest.raises(ValueError):
        pipe(img, height=60, width=128)


def test_tensor2vid_and_image_inputs():
    v = torch.linspace(-1.5, 1.5, 2 * 3 * 2 * 4 * 4).reshape(2, 3, 2, 4, 4)
    pt = tensor2vid(v, "pt")
    assert pt.shape == (2, 2, 3, 4, 4) and float(pt.min()) == 0.0 and float(pt.max()) == 1.0
    pil = tensor2vid(v, "pil")
    assert len(pil) == 2 and len(pil[0]) == 2 and pil[0][0].size == (4, 4)
    from svd_xtend_amd.pipeline import _to_unit_tensor
    from PIL import Image
    im = Image.fromarray((torch.rand(20, 30, 3) * 255).byte().numpy())
    t = _to_unit_tensor(im, 16, 24)                                                        # PIL: resized like load_image(...).resize(...)
    assert t.shape == (1, 3, 16, 24) and 0 <= float(t.min()) and float(t.max()) <= 1
    with pytest.raises(ValueError):
        _to_unit_tensor(torch.rand(3, 8, 8), 16, 16)


@gpu
def test_pipeline_matches_the_oracle_sampler_fp16():
    """Three Euler steps with guidance + the temporal decoder through the HIP kernels (fp16) against the fp32 CPU oracle."""
    dev = torch.device("cuda")
    (ou, ov, oc), (unet, vae, clip) = make_models(21, dev)
    for m in (unet, vae, clip):
        m.prepare(torch.float16)
    pipe = StableVideoDiffusionPipeline(vae, clip, unet)
    img = torch.rand(1, 3, 128, 128, generator=torch.Generator().manual_seed(22))
    kw = dict(num_frames=3, num_inference_steps=3, decode_chunk_size=2, motion_bucket_id=127, fps=7, noise_aug_strength=0.02)
    lat0 = torch.randn(1, 3, 4, 16, 16, generator=torch.Generator().manual_seed(24))
    want_l = O.svd_sample(img, ou, ov, oc, generator=torch.Generator().manual_seed(25), latents=lat0, output_latents=True, **kw)
    got_l = pipe(img, height=128, width=128, generator=torch.Generator().manual_seed(25), latents=lat0, output_type="latent", **kw).frames.cpu()
    want = O.svd_sample(img, ou, ov, oc, generator=torch.Generator().manual_seed(25), latents=lat0, **kw)
    got = pipe(img, height=128, width=128, generator=torch.Generator().manual_seed(25), latents=lat0, output_type="pt", **kw).frames.cpu()
    want01 = (want.permute(0, 2, 1, 3, 4) / 2 + 0.5).clamp(0, 1)
    print("sampler fp16: latents rel-L2", rel(got_l, want_l), " frames rel-L2", rel(got, want01))
    assert rel(got_l, want_l) <= 2e-2 and rel(got, want01) <= 2e-2
