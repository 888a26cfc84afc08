"""File-level LoRA surface of the wrapper (`/root/reference/models/stable_diffusion.py:229-335`,
`/root/reference/pipelines/train_lora_pipeline.py:496-528` file layout) and the handler's JPEG / base64 tail
(`/root/reference/runpod-worker/handler_logic.py:170-192`), on oracle-backed doubles.  CPU only."""
import base64
import io
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from doubles import OracleUNet, OracleVAE  # noqa: E402
from stablediffusion_amd import config, lora, schedulers, serving, weights  # noqa: E402
from stablediffusion_amd.pipeline import SDModelWrapper, StableDiffusionUnifiedPipeline  # noqa: E402


def _lora_sd(base, rank, seed, peft=False, alpha=None):
    g = torch.Generator().manual_seed(seed)
    down, up = (".lora_A.weight", ".lora_B.weight") if peft else (".lora.down.weight", ".lora.up.weight")
    sd = {}
    for k, w in base.items():
        if k.endswith(("to_q.weight", "to_k.weight", "to_v.weight", "to_out.0.weight")):   # train_lora_pipeline.py:247-252
            mod = "unet." + k[: -len(".weight")]
            sd[mod + down] = torch.randn(rank, w.shape[1], generator=g) * 0.05
            sd[mod + up] = torch.randn(w.shape[0], rank, generator=g) * 0.05
            if alpha is not None:
                sd[mod + ".alpha"] = torch.tensor(float(alpha))
    return sd


@pytest.fixture()
def model():
    ucfg, vcfg = config.tiny_unet(), config.tiny_vae()
    uw = weights.synth_state_dict(weights.unet_manifest(ucfg), seed=4, perturb=0.1)
    vw = weights.synth_state_dict(weights.vae_manifest(vcfg), seed=5, perturb=0.1)
    return SDModelWrapper(base=OracleUNet(ucfg, uw), vae=OracleVAE(vcfg, vw), scheduler=schedulers.DDIMScheduler(),
                          device="cpu", unet_state_dict=uw), uw


KEY = "mid_block.attentions.0.transformer_blocks.0.attn1.to_q.weight"


def test_load_lora_weights_from_the_trainers_file(model, tmp_path):
    from safetensors.torch import save_file
    m, uw = model
    a = _lora_sd(uw, 16, 1)
    os.makedirs(tmp_path / "out")
    save_file({k: v.contiguous() for k, v in a.items()}, str(tmp_path / "out" / lora.LORA_FILE))
    assert m.get_list_adapters() == {}
    m.load_lora_weights(str(tmp_path / "out"), adapter_name="style")            # folder, as the trainer writes it
    assert m.get_list_adapters() == {"base": ["style"]}
    m.apply_adapters()                                                           # (the pipeline does this before it runs)
    mod = "unet." + KEY[: -len(".weight")]
    delta = a[mod + ".lora.up.weight"] @ a[mod + ".lora.down.weight"]
    assert torch.allclose(m.base.sd[KEY], uw[KEY] + delta, atol=1e-6)            # active at weight 1
    m.set_adapters(["style"], [0.25])
    m.apply_adapters()
    assert torch.allclose(m.base.sd[KEY], uw[KEY] + 0.25 * delta, atol=1e-6)
    b = _lora_sd(uw, 8, 2, peft=True, alpha=4)                                   # second adapter: peft spelling, alpha / r = 0.5
    m.load_lora_weights(b, adapter_name="char")
    m.set_adapters(["style", "char"], [1.0, 2.0])
    m.apply_adapters()
    delta_b = b[mod + ".lora_B.weight"] @ b[mod + ".lora_A.weight"]
    assert torch.allclose(m.base.sd[KEY], uw[KEY] + delta + 2.0 * 0.5 * delta_b, atol=1e-6)
    m.delete_adapters(["style", "char"])
    m.apply_adapters()
    assert m.get_list_adapters() == {} and torch.equal(m.base.sd[KEY], uw[KEY])
    assert torch.equal(uw[KEY], weights.synth_state_dict(weights.unet_manifest(config.tiny_unet()), seed=4, perturb=0.1)[KEY])
    with pytest.raises(ValueError, match="Invalid LoRA checkpoint"):
        m.load_lora_weights({"unet.foo.weight": torch.zeros(1)})
    with pytest.raises(ValueError):
        m.load_lora_weights(str(tmp_path / "weights.bin"))                        # nothing is unpickled
    with pytest.raises(NotImplementedError):      # text-encoder layers, but this wrapper holds no text-encoder base weights
        m.load_lora_weights({"text_encoder.text_model.encoder.layers.0.self_attn.q_proj.lora_linear_layer.down.weight": torch.zeros(4, 8),
                             "text_encoder.text_model.encoder.layers.0.self_attn.q_proj.lora_linear_layer.up.weight": torch.zeros(8, 4)})
    with pytest.raises(ValueError):
        m.set_adapters(["nope"])


def test_cross_attention_kwargs_scale_reaches_the_adapters(model):
    m, uw = model
    m.load_lora_weights(_lora_sd(uw, 16, 3), adapter_name="style")
    g = torch.Generator().manual_seed(2)
    ucfg = config.tiny_unet()
    pos, neg = torch.randn(1, 77, ucfg.cross_attention_dim, generator=g), torch.randn(1, 77, ucfg.cross_attention_dim, generator=g)
    lat = torch.randn(1, 4, 8, 8, generator=g)
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu", output_type="latents")
    kw = dict(prompt_embeds=pos, negative_prompt_embeds=neg, latents=lat, num_inference_steps=2, height=64, width=64)
    full = pipe(m, **kw)
    half = pipe(m, cross_attention_kwargs={"scale": 0.5}, **kw)
    assert not torch.allclose(full, half, atol=1e-4)
    # the scale belongs to ONE call: the next call without the kwarg runs at 1.0 again (diffusers / peft
    # un-scale after the forward)
    assert torch.equal(pipe(m, **kw), full)
    m.set_adapters(["style"], [0.5])
    assert torch.allclose(pipe(m, **kw), half, atol=1e-5)        # scale 0.5 x weight 1 == scale 1 x weight 0.5
    plain = SDModelWrapper(base=OracleUNet(ucfg, uw), vae=m.vae, scheduler=schedulers.DDIMScheduler(), device="cpu")
    out = pipe(plain, cross_attention_kwargs={"scale": 0.3}, **kw)    # no adapters: the scale is a no-op
    assert torch.equal(out, pipe(plain, **kw))


def test_handler_tail_jpeg_base64(model):
    from PIL import Image
    m, _ = model
    g = torch.Generator().manual_seed(5)
    ucfg = config.tiny_unet()
    cfg = dict(prompt_embeds=torch.randn(2, 77, ucfg.cross_attention_dim, generator=g),
               negative_prompt_embeds=torch.randn(2, 77, ucfg.cross_attention_dim, generator=g),
               num_inference_steps=2, height=64, width=64, seed=3)
    resp = serving.inference_mode(m, cfg, device="cpu")
    assert list(resp) == ["images"] and len(resp["images"]) == 2
    for s in resp["images"]:
        raw = base64.b64decode(s)
        assert raw[:2] == bytes([0xFF, 0xD8])                            # JPEG SOI marker
        im = Image.open(io.BytesIO(raw))
        assert im.format == "JPEG" and im.size == (64, 64) and im.mode == "RGB"
    arr = (np.random.default_rng(0).random((32, 48, 3)) * 255).astype(np.uint8)
    one = serving.images_to_base64_jpeg([arr[:, ::-1]])           # a non-contiguous view, as np.ascontiguousarray expects
    assert Image.open(io.BytesIO(base64.b64decode(one[0]))).size == (48, 32)


def test_kohya_key_names_are_applied_or_rejected(model):
    """kohya / A1111 spelling (`lora_unet_<module with _>.lora_down.weight`, `.lora_up.weight`, `.alpha`): what the
    reference's load_loras fetches (`stable_diffusion.py:239-246`).  Applied, never silently skipped."""
    m, uw = model
    g = torch.Generator().manual_seed(9)
    mod = KEY[: -len(".weight")]
    kname = "lora_unet_" + mod.replace(".", "_")
    w = uw[KEY]
    down, up = torch.randn(4, w.shape[1], generator=g) * 0.1, torch.randn(w.shape[0], 4, generator=g) * 0.1
    m.load_lora_weights({kname + ".lora_down.weight": down, kname + ".lora_up.weight": up, kname + ".alpha": torch.tensor(2.0)},
                        adapter_name="kohya")
    m.apply_adapters()
    assert torch.allclose(m.base.sd[KEY], w + (2.0 / 4) * (up @ down), atol=1e-6)
    m.delete_adapters("kohya")
    m.load_lora_weights({kname + ".lora_down.weight": down, kname + ".lora_up.weight": up}, adapter_name="noalpha")
    m.apply_adapters()
    assert torch.allclose(m.base.sd[KEY], w + up @ down, atol=1e-6)                 # no alpha: ratio 1, NOT a zero delta
    m.delete_adapters("noalpha")
    n_before = len(m.get_list_adapters().get("base", []))
    with pytest.raises(KeyError):                                                   # a module the UNet does not have
        m.load_lora_weights({"lora_unet_no_such_module.lora_down.weight": down, "lora_unet_no_such_module.lora_up.weight": up})
    with pytest.raises(KeyError):                                                   # alpha without its matrices
        m.load_lora_weights({kname + ".alpha": torch.tensor(1.0), "lora_unet_" + mod.replace(".", "_").replace("to_q", "to_k") + ".lora_down.weight": down,
                             "lora_unet_" + mod.replace(".", "_").replace("to_q", "to_k") + ".lora_up.weight": up})
    with pytest.raises(KeyError):                                                   # half a pair
        m.load_lora_weights({kname + ".lora_down.weight": down})
    with pytest.raises(ValueError):                                                 # a spelling nobody writes
        m.load_lora_weights({"unet." + mod + ".lora_mid.weight": down})
    with pytest.raises(ValueError):                                                 # shapes that do not fit the module
        m.load_lora_weights({kname + ".lora_down.weight": down[:, :-1], kname + ".lora_up.weight": up})
    assert len(m.get_list_adapters().get("base", [])) == n_before                   # nothing half-registered
    m.apply_adapters()
    assert torch.equal(m.base.sd[KEY], w)


class _FakeTextEncoder:
    def __init__(self, sd):
        self.sd = dict(sd)
        self.loads = 0

    def load_state_dict(self, sd, strict=True):
        self.sd = dict(sd)
        self.loads += 1

    def to(self, *a, **k):
        return self


def test_text_encoder_lora_is_fused_into_the_text_encoder_weights():
    """`load_lora_into_text_encoder` (`stable_diffusion.py:275-295`): `text_encoder.` / `text_encoder_2.` layers of a
    diffusers-format file and `lora_te_` layers of a kohya file go into the CLIP weights, the same host algebra."""
    ucfg, vcfg = config.tiny_unet(), config.tiny_vae()
    uw = weights.synth_state_dict(weights.unet_manifest(ucfg), seed=4, perturb=0.1)
    vw = weights.synth_state_dict(weights.vae_manifest(vcfg), seed=5, perturb=0.1)
    g = torch.Generator().manual_seed(3)
    tkey = "text_model.encoder.layers.0.self_attn.q_proj"
    te_sd = {tkey + ".weight": torch.randn(16, 16, generator=g), tkey + ".bias": torch.zeros(16),
             "text_model.encoder.layers.0.mlp.fc1.weight": torch.randn(32, 16, generator=g)}
    enc = _FakeTextEncoder(te_sd)
    m = SDModelWrapper(base=OracleUNet(ucfg, uw), vae=OracleVAE(vcfg, vw), scheduler=schedulers.DDIMScheduler(), device="cpu",
                       unet_state_dict=uw, text_encoder=enc, text_encoder_state_dict=te_sd)
    down, up = torch.randn(4, 16, generator=g) * 0.1, torch.randn(16, 4, generator=g) * 0.1
    umod = "unet." + KEY[: -len(".weight")]
    ud, uu = torch.randn(2, uw[KEY].shape[1], generator=g) * 0.1, torch.randn(uw[KEY].shape[0], 2, generator=g) * 0.1
    m.load_lora_weights({"text_encoder." + tkey + ".lora_linear_layer.down.weight": down,
                         "text_encoder." + tkey + ".lora_linear_layer.up.weight": up,
                         umod + ".lora.down.weight": ud, umod + ".lora.up.weight": uu}, adapter_name="both")
    m.set_adapters(["both"], [0.5])
    m.apply_adapters()
    assert torch.allclose(enc.sd[tkey + ".weight"], te_sd[tkey + ".weight"] + 0.5 * (up @ down), atol=1e-6)
    assert torch.allclose(m.base.sd[KEY], uw[KEY] + 0.5 * (uu @ ud), atol=1e-6)
    assert torch.equal(enc.sd["text_model.encoder.layers.0.mlp.fc1.weight"], te_sd["text_model.encoder.layers.0.mlp.fc1.weight"])
    m.delete_adapters("both")
    kn = "lora_te_" + "text_model.encoder.layers.0.mlp.fc1".replace(".", "_")
    d2, u2 = torch.randn(4, 16, generator=g) * 0.1, torch.randn(32, 4, generator=g) * 0.1
    m.load_lora_weights({kn + ".lora_down.weight": d2, kn + ".lora_up.weight": u2, kn + ".alpha": torch.tensor(4.0)}, adapter_name="k")
    m.apply_adapters()
    assert torch.allclose(enc.sd["text_model.encoder.layers.0.mlp.fc1.weight"],
                          te_sd["text_model.encoder.layers.0.mlp.fc1.weight"] + u2 @ d2, atol=1e-6)
    assert torch.equal(enc.sd[tkey + ".weight"], te_sd[tkey + ".weight"])           # the deleted adapter is gone from CLIP too


def test_adapter_changes_rebuild_once(model):
    """The reference's load_loras is delete + N loads + set (`stable_diffusion.py:230-249`): one re-pack, at the next
    pipeline call, not N + 2; and the scheduler preset does not mistake euler_a for euler."""
    m, uw = model
    builds = []
    orig = OracleUNet.rebuild

    def counting(self, sd):
        builds.append(1)
        return orig(self, sd)

    OracleUNet.rebuild = counting
    try:
        m.delete_adapters([])
        m.load_lora_weights(_lora_sd(uw, 4, 1), adapter_name="a")
        m.load_lora_weights(_lora_sd(uw, 4, 2), adapter_name="b")
        m.set_adapters(["a", "b"], [1.0, 0.5])
        assert builds == []
        m.apply_adapters()
        m.apply_adapters()
        assert builds == [1]
    finally:
        OracleUNet.rebuild = orig
    ea = SDModelWrapper(base=m.base, vae=m.vae, scheduler=schedulers.EulerAncestralDiscreteScheduler(), device="cpu")
    assert getattr(ea, "scheduler_name", None) is None
    ea.set_scheduler("euler")
    assert type(ea.scheduler) is schedulers.EulerDiscreteScheduler
