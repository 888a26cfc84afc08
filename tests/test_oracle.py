"""The oracle against the committed golden vectors, and the host-side (product) schedulers and
pipeline logic against the oracle's independent restatement.  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import rel_l2
from oracle import pipeline_ref, schedulers_ref, unet_ref, vae_ref
from stablediffusion_amd import schedulers
from stablediffusion_amd.pipeline import SDModelWrapper, StableDiffusionUnifiedPipeline
from doubles import OracleUNet, OracleVAE

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(min(8, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def golden():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    ucfg, vcfg, uw, vw = mg.golden_weights()
    data = np.load(os.path.join(HERE, "golden", "tiny_sd.npz"))
    assert abs(mg.checksum(uw) - float(data["unet_weight_checksum"])) < 1e-6 * float(data["unet_weight_checksum"])
    assert abs(mg.checksum(vw) - float(data["vae_weight_checksum"])) < 1e-6 * float(data["vae_weight_checksum"])
    return ucfg, vcfg, uw, vw, {k: torch.from_numpy(np.asarray(data[k])) for k in data.files}


def test_oracle_reproduces_golden(golden):
    ucfg, vcfg, uw, vw, d = golden
    with torch.no_grad():
        y = unet_ref.unet_forward(ucfg, uw, d["unet_x"], d["unet_t"], d["unet_ehs"])
        img = vae_ref.vae_decode(vcfg, vw, d["vae_z"])
        mom = vae_ref.vae_encode_moments(vcfg, vw, d["vae_pix"])
    assert rel_l2(y, d["unet_y"]) < 1e-5
    assert rel_l2(img, d["vae_img"]) < 1e-5
    assert rel_l2(mom, d["vae_moments"]) < 1e-5


def test_timestep_sinusoid_layout():
    e = unet_ref.timestep_sinusoid(torch.tensor([0.0, 3.0]), 320)
    assert e.shape == (2, 320)
    assert torch.allclose(e[0, :160], torch.ones(160))      # flip_sin_to_cos: cos first
    assert torch.allclose(e[0, 160:], torch.zeros(160))
    assert abs(e[1, 0].item() - np.cos(3.0)) < 1e-6 and abs(e[1, 160].item() - np.sin(3.0)) < 1e-6
    assert abs(e[1, 159].item() - np.cos(3.0 * np.exp(-np.log(1e4) * 159 / 160))) < 1e-6


def test_ddim_schedule_constants():
    """DDIM leading spacing, steps_offset=1 -> 981, 961, ..., 1 for 50 steps."""
    s = schedulers.DDIMScheduler()
    s.set_timesteps(50)
    assert s.timesteps.tolist() == [(49 - k) * 20 + 1 for k in range(50)]
    s.set_timesteps(10)
    assert s.timesteps.tolist() == [(9 - k) * 100 + 1 for k in range(10)]
    ac = schedulers_ref.alphas_cumprod()
    assert abs(ac[0] - (1 - 0.00085)) < 1e-12 and abs(ac[-1] - 0.0046600) < 1e-4   # SD1.5 schedule end


@pytest.mark.parametrize("name,ref_cls,n", [("DDIM", schedulers_ref.DDIMRef, 7), ("DPM++ 2M", schedulers_ref.DPMpp2MRef, 6),
                                            ("euler", schedulers_ref.EulerRef, 5)])
def test_product_schedulers_match_oracle(name, ref_cls, n):
    kw = {"timestep_spacing": "leading"} if name == "DPM++ 2M" else {}
    prod = schedulers.REGISTRY[name](schedulers.DDIMScheduler(**kw).config)
    ref = ref_cls()
    prod.set_timesteps(n)
    ts = ref.set_timesteps(n)
    assert np.allclose(prod.timesteps.double().numpy(), ts)
    assert abs(prod.init_noise_sigma - ref.init_noise_sigma) < 1e-6
    g = torch.Generator().manual_seed(n)
    x = torch.randn(2, 4, 8, 8, generator=g).double() * ref.init_noise_sigma
    xr = x.numpy().copy()
    for t in prod.timesteps:
        eps = torch.randn(2, 4, 8, 8, generator=g).double()
        xi = prod.scale_model_input(x, t)
        assert np.allclose(xi.numpy(), ref.scale_model_input(xr, float(t)), atol=1e-6)
        x = prod.step(eps, t, x)[0]
        xr = ref.step(eps.numpy(), float(t), xr)
        assert np.allclose(x.numpy(), xr, atol=1e-5), (name, float(t))


@pytest.mark.parametrize("name,ref_cls", [("DDIM", schedulers_ref.DDIMRef), ("euler", schedulers_ref.EulerRef)])
def test_add_noise_matches_oracle(name, ref_cls):
    """img2img / inpaint noising (`sd_unified_pipeline.py:502, :841`): alpha-space for DDIM, sigma-space
    for Euler (x0 + sigma(t) noise) -- and the two really differ."""
    prod = schedulers.REGISTRY[name](schedulers.DDIMScheduler().config)
    ref = ref_cls()
    prod.set_timesteps(10)
    ref.set_timesteps(10)
    g = torch.Generator().manual_seed(4)
    x0, noise = torch.randn(2, 4, 8, 8, generator=g).double(), torch.randn(2, 4, 8, 8, generator=g).double()
    for t in prod.timesteps[[0, 3, 9]]:
        got = prod.add_noise(x0, noise, torch.as_tensor([float(t)]))
        assert np.allclose(got.numpy(), ref.add_noise(x0.numpy(), noise.numpy(), float(t)), atol=1e-6)
    if name == "euler":
        t = prod.timesteps[3]
        base = schedulers.DDIMScheduler.add_noise(prod, x0, noise, torch.as_tensor([int(t)]))
        assert not np.allclose(base.numpy(), prod.add_noise(x0, noise, torch.as_tensor([float(t)])).numpy(), atol=1e-2)


def test_ddim_affine_coefficients():
    s = schedulers.DDIMScheduler()
    s.set_timesteps(50)
    ref = schedulers_ref.DDIMRef()
    ref.set_timesteps(50)
    g = torch.Generator().manual_seed(1)
    x, e = torch.randn(4, generator=g).double(), torch.randn(4, generator=g).double()
    for t in (981, 501, 1):
        cx, ce = s.step_coefficients(t)
        assert np.allclose(cx * x.numpy() + ce * e.numpy(), ref.step(e.numpy(), t, x.numpy()), atol=1e-12)


@pytest.mark.parametrize("name,n", [("DDIM", 10), ("DPM++ 2M", 8), ("euler", 7)])
def test_fused_plan_equals_step(name, n):
    """The affine coefficients handed to sd_cfg_linear_step reproduce scheduler.step (fp32 emulation of
    the kernel: x0 = h_x x + h_eps eps; x <- c_x x + c_eps eps + c_hist hist; hist <- x0)."""
    cfg = schedulers.DDIMScheduler().config
    a, b = schedulers.REGISTRY[name](cfg), schedulers.REGISTRY[name](cfg)
    a.set_timesteps(n)
    b.set_timesteps(n)
    g = torch.Generator().manual_seed(n)
    xa = torch.randn(2, 4, 8, 8, generator=g) * float(a.init_noise_sigma)
    xb = xa.clone()
    hist = torch.zeros_like(xb)
    for t in a.timesteps.tolist():
        eps = torch.randn(2, 4, 8, 8, generator=g)
        plan = b.fused_plan(t)
        assert rel_l2(plan.in_scale * xb, b.scale_model_input(xb, t)) < 1e-6
        xa = a.step(eps, t, xa)[0]
        x0 = plan.h_x * xb + plan.h_eps * eps
        xb = plan.c_x * xb + plan.c_eps * eps + (plan.c_hist * hist if plan.use_hist else 0.0)
        hist = x0
        b.fused_commit()
        assert rel_l2(xb, xa) < 1e-5, (name, t)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("name,ref_cls", [("DPM++ 2M", schedulers_ref.DPMpp2MRef), ("euler", schedulers_ref.EulerRef)])
@pytest.mark.parametrize("mode", ["strength", "denoising_start"])
def test_mid_schedule_start_uses_the_sigma_of_its_timestep(name, ref_cls, mode, fused):
    """img2img / inpaint with strength < 1 and denoising_start enter the loop at timesteps[t_start:]
    (`sd_unified_pipeline.py:722-761`); the sigma-indexed schedulers must then step from sigma(t_start),
    not sigma[0]: add_noise at sigma(t_start) followed by the loop, product scheduler
    (generic step and fused plan) vs the oracle started at the same index."""
    n = 10
    kw = {"timestep_spacing": "leading"} if name == "DPM++ 2M" else {}
    prod = schedulers.REGISTRY[name](schedulers.DDIMScheduler(**kw).config)
    ref = ref_cls()
    prod.set_timesteps(n)
    ref.set_timesteps(n)
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu")
    from types import SimpleNamespace
    pipe.model = SimpleNamespace(scheduler=prod)
    if mode == "strength":
        ts, cnt = pipe.get_timesteps(n, 0.5)
        assert cnt == 5
    else:
        ts, cnt = pipe.get_timesteps(n, 1.0, denoising_start=0.35)
        assert 0 < cnt < n
    assert np.allclose(ts.double().numpy(), np.asarray(ref.timesteps[-cnt:], dtype=np.float64))
    ref.start_at(float(ts[0]))
    assert ref.i == n - cnt
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 8, 8, generator=g).double()
    xr = x.numpy().copy()
    hist = torch.zeros_like(x)
    for t in ts:
        eps = torch.randn(2, 4, 8, 8, generator=g).double()
        want_in = ref.scale_model_input(xr, float(t))
        xr = ref.step(eps.numpy(), float(t), xr)
        if fused:
            plan = prod.fused_plan(t)
            assert np.allclose((plan.in_scale * x).numpy(), want_in, atol=1e-5)
            x0 = plan.h_x * x + plan.h_eps * eps
            x = plan.c_x * x + plan.c_eps * eps + (plan.c_hist * hist if plan.use_hist else 0.0)
            hist = x0
            prod.fused_commit()
        else:
            assert np.allclose(prod.scale_model_input(x, t).numpy(), want_in, atol=1e-5)
            x = prod.step(eps, t, x)[0]
        assert np.allclose(x.numpy(), xr, atol=2e-5), (name, mode, float(t))


def test_unknown_timestep_is_rejected_by_sigma_schedulers():
    s = schedulers.EulerDiscreteScheduler()
    s.set_timesteps(10)
    with pytest.raises(ValueError, match="not in the schedule"):
        s.scale_model_input(torch.zeros(1), 123.0)


@pytest.mark.parametrize("name,ref_cls,n", [("euler_a", schedulers_ref.EulerAncestralRef, 6),
                                            ("DPM++ 2M Karras", schedulers_ref.DPMpp2MKarrasRef, 9),
                                            ("DPM++ 2M SDE Karras", schedulers_ref.DPMpp2MSDERef, 7),
                                            ("PNDM", schedulers_ref.PNDMRef, 8), ("uni_pc", schedulers_ref.UniPCRef, 7),
                                            ("uni_pc", schedulers_ref.UniPCRef, 2)])
def test_remaining_registry_schedulers_match_oracle(name, ref_cls, n):
    """The other five names of `models/stable_diffusion.py:199-227`: product (torch host code) vs the oracle's
    independent numpy float64 restatement, whole trajectories; the stochastic ones get the same noise."""
    kw = {"timestep_spacing": "leading"}
    prod = schedulers.REGISTRY[name](schedulers.DDIMScheduler(**kw).config)
    ref = ref_cls()
    prod.set_timesteps(n)
    ts = ref.set_timesteps(n)
    assert np.allclose(prod.timesteps.double().numpy(), np.asarray(ts, dtype=np.float64)), (prod.timesteps, ts)
    assert abs(float(prod.init_noise_sigma) - float(ref.init_noise_sigma)) < 1e-6
    stochastic = name in ("euler_a", "DPM++ 2M SDE Karras")
    g = torch.Generator().manual_seed(n)
    x = torch.randn(2, 4, 8, 8, generator=g).double() * float(ref.init_noise_sigma)
    xr = x.numpy().copy()
    for t in prod.timesteps:
        eps = torch.randn(2, 4, 8, 8, generator=g).double()
        assert np.allclose(prod.scale_model_input(x, t).numpy(), ref.scale_model_input(xr, float(t)), atol=5e-5)
        if stochastic:
            z = torch.randn(2, 4, 8, 8, generator=g).double()
            x = prod.step(eps, t, x, noise=z)[0]
            xr = ref.step(eps.numpy(), float(t), xr, z.numpy())
        else:
            x = prod.step(eps, t, x)[0]
            xr = ref.step(eps.numpy(), float(t), xr)
        assert np.allclose(x.numpy(), xr, atol=5e-5), (name, float(t), np.abs(x.numpy() - xr).max())


def test_registry_has_the_references_eight_names_and_stochastic_steps_draw_noise():
    assert sorted(schedulers.REGISTRY) == sorted(["DDIM", "euler", "euler_a", "DPM++ 2M", "DPM++ 2M Karras",
                                                   "DPM++ 2M SDE Karras", "PNDM", "uni_pc"])
    s = schedulers.REGISTRY["euler_a"](schedulers.DDIMScheduler().config)
    s.set_timesteps(4)
    x = torch.ones(1, 4, 4, 4)
    a = s.step(torch.zeros_like(x), s.timesteps[0], x, generator=torch.Generator().manual_seed(1))[0]
    s.set_timesteps(4)
    b = s.step(torch.zeros_like(x), s.timesteps[0], x, generator=torch.Generator().manual_seed(2))[0]
    assert not torch.equal(a, b)                       # sigma_up * noise really enters
    assert not getattr(s, "supports_fused", True)      # the pipeline keeps it on the host path


def test_pipeline_host_logic_matches_oracle_loop(golden):
    """StableDiffusionUnifiedPipeline (product host code) driving oracle-backed doubles must equal
    the oracle's own loop: checks CFG order, scheduler wiring, un-scaling and decode call."""
    ucfg, vcfg, uw, vw, d = golden
    model = SDModelWrapper(base=OracleUNet(ucfg, uw), vae=OracleVAE(vcfg, vw),
                           scheduler=schedulers.DDIMScheduler(), device="cpu")
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu")
    neg, pos = d["pipe_embeds2b"][:1], d["pipe_embeds2b"][1:]
    images = pipe(model, prompt_embeds=pos, negative_prompt_embeds=neg, latents=d["pipe_latents0"],
                  num_inference_steps=4, guidance_scale=5.0, height=64, width=64)
    assert rel_l2(images, d["pipe_images"]) < 1e-4
    u8 = pipeline_ref.to_uint8_hwc(images)
    assert np.abs(u8.astype(int) - d["pipe_uint8"].numpy().astype(int)).max() <= 1
    lat_pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu", output_type="latents")
    lat = lat_pipe(model, prompt_embeds=pos, negative_prompt_embeds=neg, latents=d["pipe_latents0"],
                   num_inference_steps=4, output_type="pt")       # kwarg ignored, ctor value wins (quirk kept)
    assert rel_l2(lat, d["pipe_latents"]) < 1e-4


def test_pipeline_img2img_and_errors(golden):
    ucfg, vcfg, uw, vw, d = golden
    model = SDModelWrapper(base=OracleUNet(ucfg, uw), vae=OracleVAE(vcfg, vw),
                           scheduler=schedulers.DDIMScheduler(), device="cpu")
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu", output_type="latents")
    neg, pos = d["pipe_embeds2b"][:1], d["pipe_embeds2b"][1:]
    image = torch.randn(1, 3, 64, 64, generator=torch.Generator().manual_seed(5)).clamp(-1, 1)
    out = pipe(model, prompt_embeds=pos, negative_prompt_embeds=neg, image=image, strength=0.5,
               num_inference_steps=4, seed=3)
    assert out.shape == (1, 4, 8, 8) and torch.isfinite(out).all()
    assert len(pipe.get_timesteps(4, 0.5)[0]) == 2                 # int(4*0.5) last steps
    with pytest.raises(ValueError):
        pipe(model)                                                 # neither prompt nor embeds
    with pytest.raises(ValueError):
        pipe(model, prompt="a cat")                                 # no tokenizer in this wrapper
    with pytest.raises(ValueError):
        pipe(model, prompt_embeds=pos, negative_prompt_embeds=neg, denoising_start=0.8, denoising_end=0.5)
    with pytest.raises(ValueError):
        model.set_scheduler("nope")
    model.set_scheduler("DPM++ 2M")
    assert type(model.scheduler).__name__ == "DPMSolverMultistepScheduler"
    # a wrapper built around a non-default scheduler must really switch to "euler" (r2: it used to
    # believe it already was "euler" and kept the DDIM object)
    m2 = SDModelWrapper(base=model.base, vae=model.vae, scheduler=schedulers.DDIMScheduler(), device="cpu")
    m2.set_scheduler("euler")
    assert type(m2.scheduler).__name__ == "EulerDiscreteScheduler"
    assert isinstance(model.scheduler, schedulers.DPMSolverMultistepScheduler)


def test_pipeline_inpaint(golden):
    """4-channel inpaint (sd_unified_pipeline.py:268-380, :492-506): outside the mask the result is
    exactly the original image's latents after the last step; inside it is denoised."""
    ucfg, vcfg, uw, vw, d = golden
    model = SDModelWrapper(base=OracleUNet(ucfg, uw), vae=OracleVAE(vcfg, vw),
                           scheduler=schedulers.DDIMScheduler(), device="cpu")
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu", output_type="latents")
    neg, pos = d["pipe_embeds2b"][:1], d["pipe_embeds2b"][1:]
    g = torch.Generator().manual_seed(9)
    image = torch.randn(1, 3, 64, 64, generator=g).clamp(-1, 1)
    mask = torch.zeros(1, 1, 64, 64)
    mask[:, :, :, 32:] = 1.0                                  # repaint the right half
    # like the reference (:176-177, :278-285) the inpaint branch resizes image and mask to height x width,
    # which default to sample_size * 8: the working size is passed explicitly
    out = pipe(model, prompt_embeds=pos, negative_prompt_embeds=neg, image=image, mask_image=mask,
               num_inference_steps=3, seed=1, height=64, width=64)
    assert out.shape == (1, 4, 8, 8) and torch.isfinite(out).all()
    gen = torch.Generator().manual_seed(1)
    ref_lat = model.vae.encode(image).latent_dist.sample(gen) * vcfg.scaling_factor
    assert torch.allclose(out[..., :4], ref_lat[..., :4], atol=1e-5)       # kept region == encoded original
    assert not torch.allclose(out[..., 4:], ref_lat[..., 4:], atol=1e-2)    # repainted region changed
    with pytest.raises(ValueError):
        pipe(model, prompt_embeds=pos, negative_prompt_embeds=neg, image=image, mask_image=mask, strength=0.1,
             num_inference_steps=3, height=64, width=64)      # int(3 * 0.1) = 0 steps
