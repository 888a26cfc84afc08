"""Host-side image / mask pre-processing (`stablediffusion_amd/image_processor.py`) and its use by the pipeline's
img2img and inpaint branches with PIL inputs, as the RunPod handler would pass them
(`/root/reference/models/stable_diffusion.py:96-101`, `/root/reference/pipelines/sd_unified_pipeline.py:238,
:270-285`).  diffusers is absent: properties and hand-computed values, not a diffusers comparison
(parity unpinned).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from doubles import OracleUNet, OracleVAE  # noqa: E402
from stablediffusion_amd import config, schedulers, weights  # noqa: E402
from stablediffusion_amd.image_processor import VaeImageProcessor  # noqa: E402
from stablediffusion_amd.pipeline import SDModelWrapper, StableDiffusionUnifiedPipeline  # noqa: E402


def _pil(h, w, seed=0):
    rng = np.random.default_rng(seed)
    return Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8))


def test_image_preprocess_pil_numpy_tensor():
    ip = VaeImageProcessor(vae_scale_factor=8)
    im = _pil(64, 72)
    t = ip.preprocess(im)
    assert t.shape == (1, 3, 64, 72) and t.dtype == torch.float32
    want = torch.from_numpy(np.asarray(im).astype(np.float32) / 255.0).permute(2, 0, 1)[None] * 2 - 1
    assert torch.allclose(t, want)                                   # same size: no resampling, /255, 2x-1
    assert ip.preprocess(_pil(70, 75)).shape == (1, 3, 64, 72)       # rounded down to a multiple of 8
    assert ip.preprocess(im, height=32, width=40).shape == (1, 3, 32, 40)
    assert ip.preprocess([im, im]).shape == (2, 3, 64, 72)
    arr = np.asarray(im).astype(np.float32) / 255.0
    assert torch.allclose(ip.preprocess(arr), want)                  # HWC numpy in [0, 1]
    assert torch.allclose(ip.preprocess(want), want)                 # already in [-1, 1]: passes through
    lat = torch.randn(2, 4, 8, 8)
    assert torch.equal(ip.preprocess(lat), lat)                      # latents are not touched
    back = ip.postprocess(t, output_type="pil")[0]
    assert np.abs(np.asarray(back).astype(int) - np.asarray(im).astype(int)).max() <= 1


def test_mask_preprocess_binarises_to_one_channel():
    mp = VaeImageProcessor(vae_scale_factor=8, do_normalize=False, do_binarize=True, do_convert_grayscale=True)
    m = np.zeros((64, 64, 3), dtype=np.uint8)
    m[:, 32:] = 200
    m[:, :32] = 90                                                   # 90 / 255 < 0.5 -> 0
    t = mp.preprocess(Image.fromarray(m))
    assert t.shape == (1, 1, 64, 64) and set(t.unique().tolist()) == {0.0, 1.0}
    assert t[0, 0, :, 32:].min() == 1 and t[0, 0, :, :32].max() == 0
    tt = mp.preprocess(torch.tensor(m[..., 0] / 255.0, dtype=torch.float32)[None])      # [1, H, W] tensor mask
    assert tt.shape == (1, 1, 64, 64) and torch.equal(tt, t)


def test_get_crop_region_pads_and_matches_aspect_ratio():
    m = np.zeros((128, 128), dtype=np.uint8)
    m[40:60, 50:90] = 255                                            # 40 wide, 20 high
    x1, y1, x2, y2 = VaeImageProcessor.get_crop_region(Image.fromarray(m), 64, 64, pad=4)
    assert x1 <= 46 and x2 >= 94 and y1 <= 36 and y2 >= 64           # contains the padded box
    assert (x2 - x1) == (y2 - y1)                                    # grown to 1 : 1
    assert 0 <= x1 < x2 <= 128 and 0 <= y1 < y2 <= 128
    x1, y1, x2, y2 = VaeImageProcessor.get_crop_region(Image.fromarray(m), 128, 64, pad=0)
    assert abs((x2 - x1) / (y2 - y1) - 2.0) < 0.15


@pytest.fixture(scope="module")
def model():
    ucfg, vcfg = config.tiny_unet(), config.tiny_vae()
    uw = weights.synth_state_dict(weights.unet_manifest(ucfg), seed=4, perturb=0.1)
    vw = weights.synth_state_dict(weights.vae_manifest(vcfg), seed=5, perturb=0.1)
    m = SDModelWrapper(base=OracleUNet(ucfg, uw), vae=OracleVAE(vcfg, vw), scheduler=schedulers.DDIMScheduler(), device="cpu")
    g = torch.Generator().manual_seed(2)
    return m, torch.randn(1, 77, ucfg.cross_attention_dim, generator=g), torch.randn(1, 77, ucfg.cross_attention_dim, generator=g)


def test_pipeline_img2img_takes_a_pil_image(model):
    m, pos, neg = model
    assert hasattr(m, "image_processor") and hasattr(m, "mask_processor")
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu")
    im = _pil(64, 64, seed=3)
    a = pipe(m, prompt_embeds=pos, negative_prompt_embeds=neg, image=im, strength=0.5, num_inference_steps=4, seed=7)
    t = VaeImageProcessor(vae_scale_factor=8).preprocess(im)
    b = pipe(m, prompt_embeds=pos, negative_prompt_embeds=neg, image=t, strength=0.5, num_inference_steps=4, seed=7)
    assert a.shape == (1, 3, 64, 64) and torch.equal(a, b)           # PIL path == pre-processed tensor path


def test_pipeline_inpaint_takes_pil_image_and_mask_with_padding_crop(model):
    m, pos, neg = model
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu", output_type="latents")
    im = _pil(96, 96, seed=4)
    mk = np.zeros((96, 96), dtype=np.uint8)
    mk[30:60, 40:70] = 255
    mask = Image.fromarray(mk)
    out = pipe(m, prompt_embeds=pos, negative_prompt_embeds=neg, image=im, mask_image=mask, num_inference_steps=3,
               seed=1, height=64, width=64)
    assert out.shape == (1, 4, 8, 8) and torch.isfinite(out).all()   # resized to 64 x 64 like the reference
    crop = pipe(m, prompt_embeds=pos, negative_prompt_embeds=neg, image=im, mask_image=mask, num_inference_steps=3,
                seed=1, height=64, width=64, padding_mask_crop=8)
    assert crop.shape == (1, 4, 8, 8) and torch.isfinite(crop).all() and not torch.equal(crop, out)
