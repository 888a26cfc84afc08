"""Parity at BASELINE.json's FULL sizes (C2: CFG batch 8 at 64x64 latents + the 512 px decode; C5: 96x96;
C4: SDXL at 128x128), where the tile variants, split-K choices and norm kernels are the ones the benchmark
runs -- the CPU oracle only finishes in seconds at C1 size.  The checker here is the same oracle code
(`oracle/unet_ref.py`, `oracle/vae_ref.py`) evaluated in fp32 ON THE GPU through PyTorch-ROCm (MIOpen /
rocBLAS fp32, TF32-style shortcuts disabled): an implementation independent of the engine's kernels,
itself first pinned to its own CPU evaluation at small size (`test_gpu_evaluation_of_the_oracle_...`).
Tolerance: relative L2 <= 1e-2 (BASELINE.json), measured values are ~2e-3."""
import os

os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")   # the checker's fp32 convs: skip MIOpen's exhaustive kernel search

import pytest  # noqa: E402
import torch  # noqa: E402

from conftest import rel_l2  # noqa: E402
from oracle import unet_ref, vae_ref  # noqa: E402
from stablediffusion_amd import config, weights  # noqa: E402
from stablediffusion_amd.models import HipAutoencoderKL, HipUNet2DConditionModel  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-2


@pytest.fixture(autouse=True)
def _true_fp32():
    old = (torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32)
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    yield
    torch.backends.cudnn.allow_tf32, torch.backends.cuda.matmul.allow_tf32 = old


def oracle_unet_on_gpu(cfg, sd16, x, t, ehs, added=None):
    """unet_ref.unet_forward with every tensor on cuda in fp32 (the sinusoid table is built on the host)."""
    w = {k: v.float().cuda() for k, v in sd16.items()}
    orig = unet_ref.timestep_sinusoid
    unet_ref.timestep_sinusoid = lambda tt, *a, **k: orig(tt.cpu(), *a, **k).cuda()
    try:
        with torch.no_grad():
            add = {k: v.float().cuda() for k, v in added.items()} if added else None
            kw = {"added_cond_kwargs": add} if add else {}
            return unet_ref.unet_forward(cfg, w, x.float().cuda(), t, ehs.float().cuda(), **kw)
    finally:
        unet_ref.timestep_sinusoid = orig


def test_gpu_evaluation_of_the_oracle_equals_its_cpu_evaluation(engine_lib):
    ucfg, vcfg = config.tiny_unet(), config.tiny_vae()
    usd = weights.synth_state_dict(weights.unet_manifest(ucfg), seed=3, dtype=torch.float16, perturb=0.1)
    vsd = weights.synth_state_dict(weights.vae_manifest(vcfg), seed=4, dtype=torch.float16, perturb=0.1)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 4, 16, 16, generator=g).half()
    ehs = torch.randn(2, 9, ucfg.cross_attention_dim, generator=g).half()
    with torch.no_grad():
        cpu = unet_ref.unet_forward(ucfg, {k: v.float() for k, v in usd.items()}, x.float(), torch.tensor(333.0), ehs.float())
        cpu_img = vae_ref.vae_decode(vcfg, {k: v.float() for k, v in vsd.items()}, x[:1].float())
        gpu_img = vae_ref.vae_decode(vcfg, {k: v.float().cuda() for k, v in vsd.items()}, x[:1].float().cuda())
    gpu = oracle_unet_on_gpu(ucfg, usd, x, torch.tensor(333.0), ehs)
    assert rel_l2(gpu, cpu) < 1e-4 and rel_l2(gpu_img, cpu_img) < 1e-4


@pytest.mark.parametrize("name,B,hw", [("C2 (512 px, CFG batch 8)", 8, 64), ("C5 (768 px img2img, CFG batch 8)", 8, 96)])
def test_sd15_unet_at_benchmark_size(engine_lib, name, B, hw):
    cfg = config.sd15_unet()
    sd = weights.synth_state_dict(weights.unet_manifest(cfg), seed=41, dtype=torch.float16, perturb=0.1)
    net = HipUNet2DConditionModel(cfg).load_state_dict(sd)
    g = torch.Generator().manual_seed(hw)
    x = torch.randn(B, 4, hw, hw, generator=g).half()
    ehs = torch.randn(B, 77, 768, generator=g).half()
    got = net(x.cuda(), torch.tensor(501.0), ehs.cuda())[0]
    ref = oracle_unet_on_gpu(cfg, sd, x, torch.tensor(501.0), ehs)
    assert torch.isfinite(got.float()).all()
    assert rel_l2(got, ref) < TOL, name


def test_sd15_vae_decode_at_benchmark_size(engine_lib):
    cfg = config.sd15_vae()
    sd = weights.synth_state_dict(weights.vae_manifest(cfg), seed=42, dtype=torch.float16, perturb=0.1)
    vae = HipAutoencoderKL(cfg).load_state_dict(sd)
    z = (torch.randn(4, 4, 64, 64, generator=torch.Generator().manual_seed(6)) * 1.5).half()
    got = vae.decode(z.cuda())[0]
    with torch.no_grad():
        ref = vae_ref.vae_decode(cfg, {k: v.float().cuda() for k, v in sd.items()}, z.float().cuda())
    assert got.shape == (4, 3, 512, 512) and rel_l2(got, ref) < TOL
    # and the encoder on the 512 px images (img2img / inpaint prep, C5's first step)
    img = got.clamp(-1, 1)
    enc = vae.encode_moments(img)
    with torch.no_grad():
        ref_m = vae_ref.vae_encode_moments(cfg, {k: v.float().cuda() for k, v in sd.items()}, img.float())
    assert rel_l2(enc[:, :4], ref_m[:, :4]) < TOL              # the means (log-variances are clamped to [-30, 20])


def test_sdxl_unet_at_benchmark_size(engine_lib):
    """C4: SDXL-base UNet, CFG batch 4 at 128x128 latents (1024 px), text_time conditioning."""
    cfg = config.sdxl_unet()
    sd = weights.synth_state_dict(weights.unet_manifest(cfg), seed=43, dtype=torch.float16, perturb=0.1)
    net = HipUNet2DConditionModel(cfg).load_state_dict(sd)
    g = torch.Generator().manual_seed(8)
    x = torch.randn(4, 4, 128, 128, generator=g).half()
    ehs = torch.randn(4, 77, 2048, generator=g).half()
    added = {"text_embeds": torch.randn(4, 1280, generator=g).half(),
             "time_ids": torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * 4)}
    got = net(x.cuda(), torch.tensor(741.0), ehs.cuda(),
              added_cond_kwargs={k: v.cuda() for k, v in added.items()})[0]
    ref = oracle_unet_on_gpu(cfg, sd, x, torch.tensor(741.0), ehs, added)
    assert torch.isfinite(got.float()).all()
    assert rel_l2(got, ref) < TOL


@pytest.mark.parametrize("steps", [10, 50])
def test_c2_loop_and_decode_at_benchmark_size(engine_lib, steps):
    """BASELINE.json C2 through the whole path: 4 latents at 64x64, CFG on, DDIM with 10 steps and with the
    benchmark's own 50, VAE decode to 512 px -- engine pipeline against the oracle loop (numpy float64
    scheduler, fp32 UNet / VAE evaluated on the GPU).  Also states the result in the uint8 pixel domain of
    `handler_logic.py:21-29`."""
    from oracle import pipeline_ref
    from stablediffusion_amd.pipeline import SDModelWrapper, StableDiffusionUnifiedPipeline
    from stablediffusion_amd.schedulers import DDIMScheduler
    ucfg, vcfg = config.sd15_unet(), config.sd15_vae()
    usd = weights.synth_state_dict(weights.unet_manifest(ucfg), seed=51, dtype=torch.float16)
    vsd = weights.synth_state_dict(weights.vae_manifest(vcfg), seed=52, dtype=torch.float16)
    model = SDModelWrapper(base=HipUNet2DConditionModel(ucfg).load_state_dict(usd),
                           vae=HipAutoencoderKL(vcfg).load_state_dict(vsd), scheduler=DDIMScheduler(), device="cuda")
    g = torch.Generator().manual_seed(12)
    lat0 = torch.randn(4, 4, 64, 64, generator=g).half()
    neg, pos = torch.randn(4, 77, 768, generator=g).half(), torch.randn(4, 77, 768, generator=g).half()
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cuda")
    images = pipe(model, prompt_embeds=pos.cuda(), negative_prompt_embeds=neg.cuda(), latents=lat0.cuda(),
                  num_inference_steps=steps, guidance_scale=5.0, height=512, width=512)
    # the oracle loop with its UNet / VAE calls routed to the GPU evaluation
    uw = {k: v.float().cuda() for k, v in usd.items()}
    vw = {k: v.float().cuda() for k, v in vsd.items()}
    emb2 = torch.cat([neg, pos]).float()
    orig_u, orig_v, orig_s = pipeline_ref.unet_forward, pipeline_ref.vae_decode, unet_ref.timestep_sinusoid
    pipeline_ref.unet_forward = lambda c, w, x, t, e, a=None: orig_u(c, w, x.cuda(), t, e.cuda(), a).cpu()
    pipeline_ref.vae_decode = lambda c, w, z: orig_v(c, w, z.cuda()).cpu()
    unet_ref.timestep_sinusoid = lambda tt, *a, **k: orig_s(tt.cpu(), *a, **k).cuda()
    try:
        ref_img, ref_lat = pipeline_ref.txt2img_ref(ucfg, uw, vcfg, vw, lat0.float(), emb2, steps=steps, guidance_scale=5.0)
    finally:
        pipeline_ref.unet_forward, pipeline_ref.vae_decode, unet_ref.timestep_sinusoid = orig_u, orig_v, orig_s
    assert images.shape == (4, 3, 512, 512) and torch.isfinite(images.float()).all()
    print(f"c2 loop, {steps} steps: image rel-L2", rel_l2(images, ref_img))
    assert rel_l2(images, ref_img) < TOL
    u8 = pipeline_ref.to_uint8_hwc(images.float().cpu()).astype(int)
    u8_ref = pipeline_ref.to_uint8_hwc(ref_img).astype(int)
    diff = abs(u8 - u8_ref)
    print(f"c2 loop, {steps} steps: uint8 mean |d|", diff.mean(), "within 4 levels", (diff <= 4).mean())
    assert diff.mean() <= 1.0 and (diff <= 4).mean() >= 0.995


# ------------------------------------------------------------------------------------------------
# Configs C4 and C5 at their own sizes
# ------------------------------------------------------------------------------------------------

class _oracle_on_gpu:
    """Route the oracle loop's UNet / VAE calls to the fp32 GPU evaluation (same code, cuda tensors)."""

    def __enter__(self):
        from oracle import pipeline_ref
        self.p = pipeline_ref
        self.saved = (pipeline_ref.unet_forward, pipeline_ref.vae_decode, unet_ref.timestep_sinusoid)
        ou, ov, os_ = self.saved

        def _u(c, w, x, t, e, a=None):
            a = {k: v.cuda() for k, v in a.items()} if a else None
            return ou(c, w, x.cuda(), t, e.cuda(), a).cpu()
        pipeline_ref.unet_forward = _u
        pipeline_ref.vae_decode = lambda c, w, z: ov(c, w, z.cuda()).cpu()
        unet_ref.timestep_sinusoid = lambda tt, *a, **k: os_(tt.cpu(), *a, **k).cuda()
        return pipeline_ref

    def __exit__(self, *exc):
        self.p.unet_forward, self.p.vae_decode, unet_ref.timestep_sinusoid = self.saved


@pytest.mark.parametrize("name,cfg_fn,B,hw", [("C4 SDXL VAE, 1024 px, batch 2", config.sdxl_vae, 2, 128),
                                              ("C5 SD1.5 VAE, 768 px, batch 4", config.sd15_vae, 4, 96)])
def test_vae_decode_and_encode_at_c4_c5_size(engine_lib, name, cfg_fn, B, hw):
    """`sd_unified_pipeline.py:511-523` (decode) and `:1020-1036` (encode for img2img) at the image sizes of
    BASELINE.json C4 (1024 px, batch 2) and C5 (768 px, batch 4)."""
    cfg = cfg_fn()
    sd = weights.synth_state_dict(weights.vae_manifest(cfg), seed=44, dtype=torch.float16, perturb=0.1)
    vae = HipAutoencoderKL(cfg).load_state_dict(sd)
    w32 = {k: v.float().cuda() for k, v in sd.items()}
    z = (torch.randn(B, 4, hw, hw, generator=torch.Generator().manual_seed(hw)) * 1.5).half()
    got = vae.decode(z.cuda())[0]
    with torch.no_grad():
        ref = vae_ref.vae_decode(cfg, w32, z.float().cuda())
    assert got.shape == (B, 3, 8 * hw, 8 * hw) and torch.isfinite(got.float()).all()
    print(name, "decode rel-L2", rel_l2(got, ref))
    assert rel_l2(got, ref) < TOL, name
    del ref
    img = got.clamp(-1, 1)
    enc = vae.encode_moments(img)
    with torch.no_grad():
        ref_m = vae_ref.vae_encode_moments(cfg, w32, img.float())
    assert torch.isfinite(enc.float()).all()
    print(name, "encode rel-L2", rel_l2(enc[:, :4], ref_m[:, :4]))
    assert rel_l2(enc[:, :4], ref_m[:, :4]) < TOL, name


def test_c4_sdxl_dpmpp2m_loop_at_benchmark_size(engine_lib):
    """BASELINE.json C4: SDXL-base, 1024 px (128x128 latents), 30-step DPM++ 2M, batch 2, CFG on -- the engine
    pipeline (device-fused CFG + scheduler update) against the oracle loop (numpy float64 DPM++ 2M, fp32 UNet
    on the GPU), all 30 steps; the latents are compared (the 1024 px decode has its own test above)."""
    from stablediffusion_amd.pipeline import SDModelWrapper, StableDiffusionUnifiedPipeline
    from stablediffusion_amd.schedulers import DDIMScheduler
    ucfg, vcfg = config.sdxl_unet(), config.sdxl_vae()
    usd = weights.synth_state_dict(weights.unet_manifest(ucfg), seed=61, dtype=torch.float16)
    vsd = weights.synth_state_dict(weights.vae_manifest(vcfg), seed=62, dtype=torch.float16)
    model = SDModelWrapper(base=HipUNet2DConditionModel(ucfg).load_state_dict(usd),
                           vae=HipAutoencoderKL(vcfg).load_state_dict(vsd), scheduler=DDIMScheduler(), device="cuda",
                           model_type="sdxl")
    model.set_scheduler("DPM++ 2M")
    g = torch.Generator().manual_seed(14)
    B, steps = 2, 30
    lat0 = torch.randn(B, 4, 128, 128, generator=g).half()
    neg, pos = torch.randn(B, 77, 2048, generator=g).half(), torch.randn(B, 77, 2048, generator=g).half()
    npool, pool = torch.randn(B, 1280, generator=g).half(), torch.randn(B, 1280, generator=g).half()
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cuda", output_type="latents")
    got = pipe(model, prompt_embeds=pos.cuda(), negative_prompt_embeds=neg.cuda(), pooled_prompt_embeds=pool.cuda(),
               negative_pooled_prompt_embeds=npool.cuda(), latents=lat0.cuda(), num_inference_steps=steps,
               guidance_scale=5.0, height=1024, width=1024)
    added = {"text_embeds": torch.cat([npool, pool]).float(),
             "time_ids": torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * (2 * B))}
    uw = {k: v.float().cuda() for k, v in usd.items()}
    with _oracle_on_gpu() as pref:
        ref = pref.denoise_ref(ucfg, uw, lat0.float(), torch.cat([neg, pos]).float(), steps=steps, guidance_scale=5.0,
                               scheduler="DPM++ 2M", added_cond_kwargs=added)
    assert got.shape == (B, 4, 128, 128) and torch.isfinite(got.float()).all()
    print("C4 loop, 30-step DPM++ 2M: latents rel-L2", rel_l2(got, ref))
    assert rel_l2(got, ref) < TOL


@pytest.mark.parametrize("sched", ["DDIM", "euler"])
def test_c5_img2img_with_fused_lora_at_full_width(engine_lib, sched):
    """BASELINE.json C5: full-width SD1.5 UNet with a rank-16 LoRA (alpha = rank; on to_q / to_k / to_v / to_out.0
    as the reference trainer writes it, `train_lora_pipeline.py:247-252`) fused on load, img2img at 96x96
    latents (768 px), batch 4, strength 0.6 of 10 steps -> 6 steps entered mid-schedule
    (`sd_unified_pipeline.py:236-264`, `:722-761`).  The noise the product draws from its seeded device
    generator is reproduced with the same generator and handed to the oracle."""
    from stablediffusion_amd.pipeline import SDModelWrapper, StableDiffusionUnifiedPipeline
    from stablediffusion_amd.schedulers import DDIMScheduler
    ucfg, vcfg = config.sd15_unet(), config.sd15_vae()
    base = weights.synth_state_dict(weights.unet_manifest(ucfg), seed=71, dtype=torch.float16)
    g = torch.Generator().manual_seed(5)
    lora = {}
    for k, w in base.items():
        if k.endswith(("to_q.weight", "to_k.weight", "to_v.weight", "to_out.0.weight")):
            mod = k[: -len(".weight")]
            lora[f"unet.{mod}.lora.down.weight"] = torch.randn(16, w.shape[1], generator=g) * (w.shape[1] ** -0.5)
            lora[f"unet.{mod}.lora.up.weight"] = torch.randn(w.shape[0], 16, generator=g) * 0.05
    fused = {k: v.half() for k, v in weights.fuse_lora({k: v.float() for k, v in base.items()}, lora,
                                                       adapter_weight=0.8).items()}
    k0 = "mid_block.attentions.0.transformer_blocks.0.attn1.to_q.weight"
    assert not torch.equal(fused[k0], base[k0])
    vsd = weights.synth_state_dict(weights.vae_manifest(vcfg), seed=72, dtype=torch.float16)
    model = SDModelWrapper(base=HipUNet2DConditionModel(ucfg).load_state_dict(fused),
                           vae=HipAutoencoderKL(vcfg).load_state_dict(vsd), scheduler=DDIMScheduler(), device="cuda")
    model.set_scheduler(sched)
    B, steps, strength, seed = 4, 10, 0.6, 11
    init = torch.randn(B, 4, 96, 96, generator=g).half()
    neg, pos = torch.randn(B, 77, 768, generator=g).half(), torch.randn(B, 77, 768, generator=g).half()
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cuda", output_type="latents")
    got = pipe(model, prompt_embeds=pos.cuda(), negative_prompt_embeds=neg.cuda(), image=init.cuda(), strength=strength,
               num_inference_steps=steps, guidance_scale=5.0, seed=seed)
    noise = torch.randn(init.shape, generator=torch.Generator(device="cuda").manual_seed(seed), device="cuda",
                        dtype=torch.float16).cpu()
    uw = {k: v.float().cuda() for k, v in fused.items()}
    with _oracle_on_gpu() as pref:
        ref = pref.img2img_denoise_ref(ucfg, uw, init.float(), noise.float(), torch.cat([neg, pos]).float(), steps=steps,
                                       strength=strength, guidance_scale=5.0, scheduler=sched)
    assert got.shape == (B, 4, 96, 96) and torch.isfinite(got.float()).all()
    print(f"C5 img2img + LoRA, {sched}: latents rel-L2", rel_l2(got, ref))
    assert rel_l2(got, ref) < TOL


# ---------------------------------------------------------------------------------------------------------------
# Checkpoint-like dynamic range.  Real weights cannot be had offline, so the stress is synthetic:
# weights.synth_state_dict(profile="heavy_tail") -- outlier output channels (x30 on 1 % of the rows of every conv /
# linear), biases ~ N(0, 0.5), norm gains log-uniform in [0.2, 3] -- and inputs at 4x the usual scale.  What the fp16
# inter-layer storage, the pre-scaled fp16 queries, the pkrtz probabilities and the 2^8 skip-rescale threshold of the
# attention kernel see is REPORTED (per-layer max |activation| of the fp32 oracle, printed with -s), not assumed.
# ---------------------------------------------------------------------------------------------------------------
class _ActivationLog:
    """max |output| of every conv / linear the oracle evaluates (oracle.unet_ref._conv / _lin patched, in every oracle
    module that imported them by name)."""

    def __init__(self, *mods):
        self.mods, self.rows = mods, []

    def __enter__(self):
        oc, ol = unet_ref._conv, unet_ref._lin

        def conv(x, w, p, *a, **k):
            y = oc(x, w, p, *a, **k)
            self.rows.append((p, float(y.abs().max())))
            return y

        def lin(x, w, p, *a, **k):
            y = ol(x, w, p, *a, **k)
            self.rows.append((p, float(y.abs().max())))
            return y

        self.saved = [(m, m._conv, m._lin) for m in self.mods]
        for m in self.mods:
            m._conv, m._lin = conv, lin
        return self

    def __exit__(self, *exc):
        for m, c, l in self.saved:
            m._conv, m._lin = c, l

    def report(self, title, top=8):
        rows = sorted(self.rows, key=lambda r: -r[1])
        print(f"\n[{title}] {len(rows)} conv / linear outputs; largest |activation| (fp16 max 65504):")
        for p, v in rows[:top]:
            print(f"    {v:10.1f}  {p}")
        return rows[0][1] if rows else 0.0


def oracle_unet_fp16_on_gpu(cfg, sd16, x, t, ehs):
    """The same oracle code with fp16 weights and activations through PyTorch-ROCm (rocBLAS / MIOpen fp16, fp32
    accumulation; F.group_norm / layer_norm / SDPA as torch runs them in half): what ANY fp16 evaluation of the network
    loses against fp32 on these inputs -- the yardstick for the engine's own fp16 storage under stress."""
    w = {k: v.half().cuda() for k, v in sd16.items()}
    orig = unet_ref.timestep_sinusoid
    unet_ref.timestep_sinusoid = lambda tt, *a, **k: orig(tt.cpu(), *a, **k).cuda().half()
    try:
        with torch.no_grad():
            return unet_ref.unet_forward(cfg, w, x.half().cuda(), t, ehs.half().cuda())
    finally:
        unet_ref.timestep_sinusoid = orig


@pytest.mark.parametrize("row_scale,in_scale", [(8.0, 2.0), (16.0, 4.0)])
def test_sd15_unet_heavy_tailed_weights_at_benchmark_size(engine_lib, row_scale, in_scale):
    cfg = config.sd15_unet()
    sd = weights.synth_state_dict(weights.unet_manifest(cfg), seed=51, dtype=torch.float16, profile=f"heavy_tail:{row_scale}")
    net = HipUNet2DConditionModel(cfg).load_state_dict(sd)
    g = torch.Generator().manual_seed(12)
    x = (torch.randn(8, 4, 64, 64, generator=g) * in_scale).half()
    ehs = (torch.randn(8, 77, 768, generator=g) * in_scale).half()
    got = net(x.cuda(), torch.tensor(501.0), ehs.cuda())[0]
    with _ActivationLog(unet_ref) as log:
        ref = oracle_unet_on_gpu(cfg, sd, x, torch.tensor(501.0), ehs)
    peak = log.report(f"SD1.5 UNet, heavy-tailed weights (outlier rows x{row_scale:g}), {in_scale:g}x inputs")
    ref16 = oracle_unet_fp16_on_gpu(cfg, sd, x, torch.tensor(501.0), ehs)
    assert torch.isfinite(got.float()).all()
    assert peak < 65504.0, "the stress profile must stay inside fp16's range for the comparison to mean anything"
    err, err16 = rel_l2(got, ref), rel_l2(ref16, ref)
    print(f"    rel-L2 vs the fp32 oracle: engine {err:.2e}, the oracle itself evaluated in fp16 by PyTorch-ROCm {err16:.2e} "
          f"(tolerance {TOL:.0e}, or twice what torch's fp16 loses)")
    assert err < max(TOL, 2.0 * err16)


def test_sd15_vae_heavy_tailed_weights_at_benchmark_size(engine_lib):
    cfg = config.sd15_vae()
    sd = weights.synth_state_dict(weights.vae_manifest(cfg), seed=52, dtype=torch.float16, profile="heavy_tail")
    vae = HipAutoencoderKL(cfg).load_state_dict(sd)
    z = (torch.randn(4, 4, 64, 64, generator=torch.Generator().manual_seed(7)) * 4.0).half()
    got = vae.decode(z.cuda())[0]
    with torch.no_grad(), _ActivationLog(unet_ref, vae_ref) as log:
        ref = vae_ref.vae_decode(cfg, {k: v.float().cuda() for k, v in sd.items()}, z.float().cuda())
    peak = log.report("SD1.5 VAE decoder 512 px, heavy-tailed weights, 4x latents")
    assert torch.isfinite(got.float()).all() and peak < 65504.0
    err = rel_l2(got, ref)
    print(f"    rel-L2 vs the fp32 oracle: {err:.2e} (tolerance {TOL:.0e})")
    assert err < TOL


def test_sd15_unet_at_128x128_latents(engine_lib):
    """north_star's second input shape, 4 x 128 x 128 latents (1024 px; CFG batch 8): parity, not only timing."""
    cfg = config.sd15_unet()
    sd = weights.synth_state_dict(weights.unet_manifest(cfg), seed=41, dtype=torch.float16, perturb=0.1)
    net = HipUNet2DConditionModel(cfg).load_state_dict(sd)
    g = torch.Generator().manual_seed(128)
    x = torch.randn(8, 4, 128, 128, generator=g).half()
    ehs = torch.randn(8, 77, 768, generator=g).half()
    got = net(x.cuda(), torch.tensor(261.0), ehs.cuda())[0]
    ref = oracle_unet_on_gpu(cfg, sd, x, torch.tensor(261.0), ehs)
    assert torch.isfinite(got.float()).all()
    assert rel_l2(got, ref) < TOL


def test_force_upcast_vae_encode_beyond_fp16_range(engine_lib):
    """`config.force_upcast` (sd_unified_pipeline.py:1020-1036: the reference runs SDXL's VAE in fp32 around encode because
    its activations leave fp16's range).  An SDXL-configured encoder whose first convolution is scaled until plain fp16
    storage overflows (residual stream ~ 2e5 > 65504): the engine re-runs it with its activations stored 2^-k times
    smaller (GroupNorm is scale-invariant; sd_vae_encode_range_shift) and has to match the fp32 oracle, which never
    notices the scale.  Round 2 raised EngineError here."""
    cfg = config.sdxl_vae()
    assert cfg.force_upcast
    sd = weights.synth_state_dict(weights.vae_manifest(cfg), seed=61, dtype=torch.float16, perturb=0.1)
    sd["encoder.conv_in.weight"] = (sd["encoder.conv_in.weight"].float() * 2.0e4).half()     # |conv_in(x)| ~ 1e5
    sd["encoder.conv_in.bias"] = (sd["encoder.conv_in.bias"].float() * 2.0e4).half()
    vae = HipAutoencoderKL(cfg).load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    img = (torch.rand(2, 3, 256, 256, generator=g) * 2 - 1).half()
    with torch.no_grad(), _ActivationLog(unet_ref, vae_ref) as log:
        ref = vae_ref.vae_encode_moments(cfg, {k: v.float().cuda() for k, v in sd.items()}, img.float().cuda())
    peak = log.report("SDXL-config VAE encoder, conv_in x 2e4")
    assert peak > 65504.0, "the case must really leave fp16's range"
    got = vae.encode_moments(img.cuda())
    assert vae._enc_shift > 0 and torch.isfinite(got.float()).all()
    err = rel_l2(got[:, :4], ref[:, :4])
    print(f"    activations stored 2^-{vae._enc_shift} times smaller: means rel-L2 vs the fp32 oracle {err:.2e}")
    assert err < TOL
    # an ordinary image on the same handle keeps working at the raised shift (precision unchanged: powers of two)
    plain = HipAutoencoderKL(cfg).load_state_dict(weights.synth_state_dict(weights.vae_manifest(cfg), seed=61, dtype=torch.float16, perturb=0.1))
    a = plain.encode_moments(img.cuda())
    engine_lib.sd_vae_encode_range_shift(plain._h, 8)
    b = plain.encode_moments(img.cuda())
    assert plain._enc_shift == 0 and rel_l2(b[:, :4], a[:, :4]) < 2e-3


@pytest.mark.parametrize("B,H,W", [(3, 32, 48), (1, 64, 64), (2, 64, 64), (5, 24, 40)])
def test_sd15_unet_odd_batches_and_non_square_maps(engine_lib, B, H, W):
    """Full-width SD1.5 UNet off the benchmark's shapes: the kernels with shape conditions (the fused feed-forward needs
    >= 64 blocks of 128 rows, conv_in 128-pixel tiles inside one image, the conv_out tail 8 x 16 pixel tiles, the halo
    kernels a 256-pixel patch, the tuned tile table the benchmark's M x N x K) must fall back, not mis-index: odd batch,
    48- / 40-wide maps (16-wide patches; 24 x 40: no 128-pixel tile, no 256-pixel patch at the lower levels), batch 1 and
    2 at 64 x 64 (the fused feed-forward's threshold from both sides)."""
    cfg = config.sd15_unet()
    sd = weights.synth_state_dict(weights.unet_manifest(cfg), seed=43, dtype=torch.float16, perturb=0.1)
    net = HipUNet2DConditionModel(cfg).load_state_dict(sd)
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, 4, H, W, generator=g).half()
    ehs = torch.randn(B, 77, 768, generator=g).half()
    t = torch.tensor(77.0)
    got = net(x.cuda(), t, ehs.cuda())[0]
    ref = oracle_unet_on_gpu(cfg, sd, x, t, ehs)
    assert torch.isfinite(got.float()).all()
    assert rel_l2(got, ref) < TOL, rel_l2(got, ref)
    again = net(x.cuda(), t, ehs.cuda())[0]                  # same shapes -> bitwise the same (no atomics, fixed orders)
    assert torch.equal(got, again)


@pytest.mark.parametrize("B,h,w", [(3, 24, 40), (1, 40, 56), (5, 16, 16)])
def test_sd15_vae_odd_batches_and_non_square_maps(engine_lib, B, h, w):
    """Full-width SD1.5 VAE decode + encode off the benchmark's shapes (192 x 320, 320 x 448, 128 x 128 px; odd batches):
    statistics slabs that do not divide the map, halo patches of every width, the upsample-fused convolutions and the
    3-channel conv_out on maps the tuned table has no row for."""
    cfg = config.sd15_vae()
    sd = weights.synth_state_dict(weights.vae_manifest(cfg), seed=44, dtype=torch.float16, perturb=0.1)
    vae = HipAutoencoderKL(cfg).load_state_dict(sd)
    z = (torch.randn(B, 4, h, w, generator=torch.Generator().manual_seed(h * w)) * 1.5).half()
    got = vae.decode(z.cuda())[0]
    w32 = {k: v.float().cuda() for k, v in sd.items()}
    with torch.no_grad():
        ref = vae_ref.vae_decode(cfg, w32, z.float().cuda())
    assert got.shape == (B, 3, 8 * h, 8 * w) and torch.isfinite(got.float()).all()
    assert rel_l2(got, ref) < TOL, rel_l2(got, ref)
    img = got.clamp(-1, 1)
    enc = vae.encode_moments(img)
    with torch.no_grad():
        ref_m = vae_ref.vae_encode_moments(cfg, w32, img.float())
    assert rel_l2(enc[:, :4], ref_m[:, :4]) < TOL
    assert torch.equal(got, vae.decode(z.cuda())[0])


@pytest.mark.parametrize("B,H,W", [(3, 40, 24), (1, 64, 96)])
def test_sdxl_unet_odd_batches_and_non_square_maps(engine_lib, B, H, W):
    """SDXL-base UNet (text_time conditioning, d = 64 heads, two / ten transformer layers per block) off the benchmark's
    shape: odd batch on 40 x 24 latents, batch 1 on 64 x 96 (512 x 768 px)."""
    cfg = config.sdxl_unet()
    sd = weights.synth_state_dict(weights.unet_manifest(cfg), seed=45, dtype=torch.float16, perturb=0.1)
    net = HipUNet2DConditionModel(cfg).load_state_dict(sd)
    g = torch.Generator().manual_seed(B + H)
    x = torch.randn(B, 4, H, W, generator=g).half()
    ehs = torch.randn(B, 77, 2048, generator=g).half()
    added = {"text_embeds": torch.randn(B, 1280, generator=g).half(),
             "time_ids": torch.tensor([[8.0 * H, 8.0 * W, 0, 0, 8.0 * H, 8.0 * W]] * B)}
    t = torch.tensor(333.0)
    got = net(x.cuda(), t, ehs.cuda(), added_cond_kwargs={k: v.cuda() for k, v in added.items()})[0]
    ref = oracle_unet_on_gpu(cfg, sd, x, t, ehs, added)
    assert torch.isfinite(got.float()).all()
    assert rel_l2(got, ref) < TOL, rel_l2(got, ref)
