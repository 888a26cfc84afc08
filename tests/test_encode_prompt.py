"""Host-side `encode_prompt` (/root/reference/pipelines/sd_unified_pipeline.py:532-719) with a randomly
initialised transformers CLIPTextModel and a stand-in tokenizer (no vocabulary files exist offline):
checks the SD1.5 / clip_skip / SDXL (two encoders, hidden_states[-2], pooled) selection logic and the
negative-prompt / num_images_per_prompt plumbing.  CPU only."""
from types import SimpleNamespace

import pytest
import torch

from stablediffusion_amd.pipeline import SDModelWrapper, StableDiffusionUnifiedPipeline

transformers = pytest.importorskip("transformers")


class FakeTokenizer:
    model_max_length = 77

    def __call__(self, texts, padding=None, max_length=77, truncation=True, return_tensors="pt"):
        texts = [texts] if isinstance(texts, str) else texts
        ids = torch.zeros(len(texts), max_length, dtype=torch.long)
        for i, t in enumerate(texts):
            codes = [1] + [3 + (ord(c) % 90) for c in t][: max_length - 2] + [2]
            ids[i, : len(codes)] = torch.tensor(codes)
        return SimpleNamespace(input_ids=ids)


def _clip(hidden, proj=None, seed=0):
    torch.manual_seed(seed)
    cfg = transformers.CLIPTextConfig(vocab_size=100, hidden_size=hidden, intermediate_size=2 * hidden,
                                      num_hidden_layers=3, num_attention_heads=4, max_position_embeddings=77,
                                      projection_dim=proj or hidden, bos_token_id=1, eos_token_id=2)
    cls = transformers.CLIPTextModelWithProjection if proj else transformers.CLIPTextModel
    return cls(cfg).eval()


def _model(sdxl=False):
    base = SimpleNamespace(dtype=torch.float32, config=SimpleNamespace(sample_size=8, in_channels=4), to=lambda d: None)
    vae = SimpleNamespace(config=SimpleNamespace(block_out_channels=(1, 1, 1, 1)), to=lambda d: None)
    kw = dict(base=base, vae=vae, text_encoder=_clip(32), tokenizer=FakeTokenizer(), device="cpu")
    if sdxl:
        kw.update(text_encoder_2=_clip(48, proj=40, seed=1), tokenizer_2=FakeTokenizer(), model_type="sdxl")
    return SDModelWrapper(**kw)


def test_sd15_encode_prompt_and_clip_skip():
    m = _model()
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu")
    pipe.model = m
    pe, ne, pooled, npooled = pipe.encode_prompt(["a cat", "a dog"], num_images_per_prompt=2)
    assert pe.shape == (4, 77, 32) and ne.shape == (4, 77, 32)
    ids = m.tokenizer(["a cat", "a dog"]).input_ids
    with torch.no_grad():
        out = m.text_encoder(ids, output_hidden_states=True)
    assert torch.allclose(pe[0], out[0][0]) and torch.allclose(pe[1], out[0][0]) and torch.allclose(pe[2], out[0][1])
    with torch.no_grad():
        neg = m.text_encoder(m.tokenizer(["", ""]).input_ids)[0]       # default negative prompt "" (:619)
    assert torch.allclose(ne[0], neg[0])
    pe2, *_ = pipe.encode_prompt("a cat", clip_skip=1)                  # final_layer_norm(hidden_states[-2]) (:608)
    with torch.no_grad():
        o = m.text_encoder(m.tokenizer("a cat").input_ids, output_hidden_states=True)
        want = getattr(m.text_encoder, 'text_model', m.text_encoder).final_layer_norm(o.hidden_states[-2])
    assert torch.allclose(pe2, want, atol=1e-6)
    with pytest.raises(ValueError):
        pipe.encode_prompt(["a", "b"], negative_prompt=["x"])            # batch mismatch (:635-640)


def test_sdxl_encode_prompt_two_encoders():
    m = _model(sdxl=True)
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu")
    pipe.model = m
    pe, ne, pooled, npooled = pipe.encode_prompt("a cat", negative_prompt="blurry")
    assert pe.shape == (1, 77, 32 + 48) and ne.shape == (1, 77, 80)       # concat on the last dim (:613)
    assert pooled.shape == (1, 40) and npooled.shape == (1, 40)           # text_encoder_2(...)[0] (:596)
    ids = m.tokenizer("a cat").input_ids
    with torch.no_grad():
        o1 = m.text_encoder(ids, output_hidden_states=True)
        o2 = m.text_encoder_2(ids, output_hidden_states=True)
    assert torch.allclose(pe[..., :32], o1.hidden_states[-2]) and torch.allclose(pe[..., 32:], o2.hidden_states[-2])
    assert torch.allclose(pooled, o2[0])
