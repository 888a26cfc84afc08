"""CLIP text encoder on the HIP engine against transformers' own
CLIPTextModel / CLIPTextModelWithProjection run in fp32 on the CPU from the same (fp16-rounded)
randomly initialised weights -- a third-party implementation importable in this image, so this
component's parity IS pinned (the reference pins transformers==4.39.3, requirements.txt:175; the
image has a newer release of the same model code).  Tolerance: relative L2 <= 5e-3 per output."""
import ctypes as C

import pytest
import torch

from conftest import rel_l2
from stablediffusion_amd import config
from stablediffusion_amd.models import HipCLIPTextModel

transformers = pytest.importorskip("transformers")
pytestmark = pytest.mark.gpu


def _hf(cfg: config.CLIPTextConfig, seed: int, proj: bool):
    torch.manual_seed(seed)
    hf_cfg = transformers.CLIPTextConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
        max_position_embeddings=cfg.max_position_embeddings, hidden_act=cfg.hidden_act,
        projection_dim=cfg.projection_dim or cfg.hidden_size, layer_norm_eps=cfg.layer_norm_eps,
        bos_token_id=cfg.bos_token_id, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id)
    cls = transformers.CLIPTextModelWithProjection if proj else transformers.CLIPTextModel
    m = cls(hf_cfg).eval()
    sd = {k: v.half().float() for k, v in m.state_dict().items()}       # both sides see fp16-representable weights
    # random-init LayerNorms are (1, 0): perturb them so every affine path is exercised
    g = torch.Generator().manual_seed(seed + 100)
    for k in sd:
        if "layer_norm" in k:
            sd[k] = (sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)).half().float()
        elif k.endswith(".bias"):
            sd[k] = (0.05 * torch.randn(sd[k].shape, generator=g)).half().float()
    m.load_state_dict(sd)
    return m, sd


def _ids(cfg, B, T, seed):
    g = torch.Generator().manual_seed(seed)
    hi = cfg.vocab_size - 2 if cfg.eos_token_id == 2 else cfg.eos_token_id
    ids = torch.randint(3, hi, (B, T), generator=g)
    ids[:, 0] = cfg.bos_token_id
    for b in range(B):
        e = 5 + 7 * b
        if cfg.eos_token_id == 2:
            ids[b, e] = cfg.vocab_size - 1          # legacy pooling: the largest id marks the end
        else:
            ids[b, e] = cfg.eos_token_id
        ids[b, e + 1:] = cfg.pad_token_id if cfg.eos_token_id != 2 else 0
    return ids


def _check(cfg, proj, B, seed, tol=5e-3):
    hf, sd = _hf(cfg, seed, proj)
    eng = HipCLIPTextModel(cfg).load_state_dict(sd)
    ids = _ids(cfg, B, cfg.max_position_embeddings, seed)
    with torch.no_grad():
        ref = hf(ids, output_hidden_states=True)
    out = eng(ids.cuda(), output_hidden_states=True)
    assert len(out.hidden_states) == cfg.num_hidden_layers + 1
    for i, (a, b) in enumerate(zip(out.hidden_states, ref.hidden_states)):
        assert rel_l2(a, b) < tol, f"hidden_states[{i}]: {rel_l2(a, b)}"
    assert rel_l2(out.last_hidden_state, ref.last_hidden_state) < tol
    if proj:
        assert out.keys()[0] == "text_embeds" and rel_l2(out[0], ref.text_embeds) < tol
        assert out[-1] is out.hidden_states
    else:
        assert rel_l2(out[0], ref.last_hidden_state) < tol
        assert rel_l2(out.pooler_output, ref.pooler_output) < tol
    # the clip_skip branch of encode_prompt (:608): final_layer_norm applied to an earlier hidden state
    tm = getattr(hf, "text_model", hf)
    want = tm.final_layer_norm(ref.hidden_states[-2])
    assert rel_l2(eng.text_model.final_layer_norm(out.hidden_states[-2]), want) < tol
    # without output_hidden_states the same last_hidden_state comes back (ping-pong residual buffers)
    out2 = eng(ids.cuda())
    assert out2.hidden_states is None and torch.equal(out2.last_hidden_state, out.last_hidden_state)
    return eng, hf


def test_tiny_clip_quick_gelu_all_hidden_states(engine_lib):
    cfg = config.CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                                num_attention_heads=2, eos_token_id=2)
    _check(cfg, proj=False, B=3, seed=0)


def test_tiny_clip_gelu_with_projection_and_explicit_eos(engine_lib):
    cfg = config.CLIPTextConfig(vocab_size=500, hidden_size=192, intermediate_size=320, num_hidden_layers=2,
                                num_attention_heads=3, hidden_act="gelu", projection_dim=96,
                                eos_token_id=499, bos_token_id=498, pad_token_id=1)
    _check(cfg, proj=True, B=2, seed=1)


def test_full_size_clip_l(engine_lib):
    """CLIP ViT-L/14 text tower as SD1.5 uses it (123 M parameters)."""
    _check(config.clip_l(), proj=False, B=2, seed=2)


def test_full_size_openclip_bigg_with_projection(engine_lib):
    """SDXL's text_encoder_2 (OpenCLIP bigG text tower, 32 layers x 1280, erf-gelu, text_projection)."""
    _check(config.openclip_bigg(), proj=True, B=1, seed=3)


@pytest.mark.parametrize("B,T,heads,d", [(2, 77, 12, 64), (1, 200, 2, 64), (2, 77, 4, 32), (1, 300, 2, 40)])
def test_causal_attention_op(engine_lib, B, T, heads, d):
    g = torch.Generator().manual_seed(T + d)
    q, k, v = (torch.randn(B, T, heads * d, generator=g).half() for _ in range(3))
    ref = torch.nn.functional.scaled_dot_product_attention(
        *(t.float().view(B, T, heads, d).transpose(1, 2) for t in (q, k, v)), is_causal=True)
    ref = ref.transpose(1, 2).reshape(B, T, heads * d)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    out = torch.empty_like(qd)
    P = lambda t: C.c_void_p(t.data_ptr())
    ld = heads * d
    rc = engine_lib.sd_op_attention_ex(P(qd), P(kd), P(vd), P(out), B, T, T, heads, d, ld, ld, ld, ld, 1, 0,
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, engine_lib.sd_last_error()
    torch.cuda.synchronize()
    assert rel_l2(out, ref) < 2e-3


def test_encode_prompt_through_the_engine(engine_lib):
    """encode_prompt (sd_unified_pipeline.py:532-719) with the HIP encoders in the text_encoder slots:
    SDXL selection logic (hidden_states[-2] of both, pooled = text_encoder_2(...)[0]) and SD1.5 clip_skip."""
    from types import SimpleNamespace
    from stablediffusion_amd.pipeline import SDModelWrapper, StableDiffusionUnifiedPipeline
    from test_encode_prompt import FakeTokenizer
    c1 = config.CLIPTextConfig(vocab_size=100, hidden_size=128, intermediate_size=256, num_hidden_layers=3,
                               num_attention_heads=2, eos_token_id=2)
    c2 = config.CLIPTextConfig(vocab_size=100, hidden_size=192, intermediate_size=384, num_hidden_layers=3,
                               num_attention_heads=3, hidden_act="gelu", projection_dim=64, eos_token_id=2)
    hf1, sd1 = _hf(c1, 5, False)
    hf2, sd2 = _hf(c2, 6, True)
    e1, e2 = HipCLIPTextModel(c1).load_state_dict(sd1), HipCLIPTextModel(c2).load_state_dict(sd2)
    base = SimpleNamespace(dtype=torch.float16, config=SimpleNamespace(sample_size=8, in_channels=4), to=lambda d: None)
    vae = SimpleNamespace(config=SimpleNamespace(block_out_channels=(1, 1, 1, 1)), to=lambda d: None)

    def wrap(t1, t2=None, dev="cuda"):
        kw = dict(base=base, vae=vae, text_encoder=t1, tokenizer=FakeTokenizer(), device=dev)
        if t2 is not None:
            kw.update(text_encoder_2=t2, tokenizer_2=FakeTokenizer(), model_type="sdxl")
        return SDModelWrapper(**kw)

    def run(model, dev, **kw):
        pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device=dev)
        pipe.model = model
        with torch.no_grad():
            return pipe.encode_prompt(["a cat", "two dogs"], negative_prompt="blurry", **kw)

    base.dtype = torch.float32
    want = run(wrap(hf1, hf2, "cpu"), "cpu")
    base.dtype = torch.float16
    got = run(wrap(e1, e2), "cuda")
    for g_, w_ in zip(got, want):
        assert g_.shape == w_.shape and rel_l2(g_, w_) < 5e-3
    base.dtype = torch.float32
    want = run(wrap(hf1, dev="cpu"), "cpu", clip_skip=1)
    base.dtype = torch.float16
    got = run(wrap(e1), "cuda", clip_skip=1)
    assert rel_l2(got[0], want[0]) < 5e-3 and rel_l2(got[1], want[1]) < 5e-3
