"""The C-ABI library loads and exports every symbol include/sd_engine.h declares; argument
validation works without a GPU (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from stablediffusion_amd import _lib, config
from stablediffusion_amd.models import HipAutoencoderKL, HipUNet2DConditionModel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "sd_engine.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(engine_lib):
    syms = header_symbols()
    assert len(syms) >= 25
    assert sorted(_lib.SIGNATURES) == syms
    for s in syms:
        assert hasattr(engine_lib, s), s


def test_struct_layout_matches_header():
    src = open(os.path.join(ROOT, "include", "sd_engine.h")).read()
    for struct, cls in (("sd_unet_config", _lib.SdUNetConfig), ("sd_vae_config", _lib.SdVAEConfig),
                        ("sd_clip_config", _lib.SdClipConfig), ("sd_prof_entry", _lib.SdProfEntry)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = [re.sub(r"\[.*", "", d.split()[-1]) for d in body.split(";") if d.strip()]
        assert names == [f[0] for f in cls._fields_], struct


def test_library_identity(engine_lib):
    assert engine_lib.sd_engine_arch() == b"gfx950"
    assert engine_lib.sd_engine_version() >= 1


def test_unsupported_config_is_rejected(engine_lib):
    c = _lib.SdUNetConfig()
    c.num_blocks = 1
    c.block_out_channels[0] = 100           # not a multiple of 64
    c.norm_num_groups = 32
    c.num_heads[0] = 4
    h = C.c_void_p()
    assert engine_lib.sd_unet_create(C.byref(c), C.byref(h)) == 4
    assert b"unsupported" in engine_lib.sd_last_error()


def test_clip_manifest_matches_transformers_state_dict(engine_lib):
    """The weight names / shapes sd_clip_weight_info reports are exactly the state-dict entries of
    transformers' CLIPTextModel(WithProjection) of the same config (host objects only, no GPU)."""
    transformers = pytest.importorskip("transformers")
    from stablediffusion_amd.models import HipCLIPTextModel
    for proj in (0, 48):
        cfg = config.CLIPTextConfig(vocab_size=90, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                    num_attention_heads=2, projection_dim=proj)
        eng = HipCLIPTextModel(cfg)
        got = {}
        for i in range(engine_lib.sd_clip_num_weights(eng._h)):
            key, shape, ndim = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
            assert engine_lib.sd_clip_weight_info(eng._h, i, C.byref(key), shape, C.byref(ndim)) == 0
            got[key.value.decode()] = tuple(shape[j] for j in range(ndim.value))
        hf_cfg = transformers.CLIPTextConfig(vocab_size=90, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                                             num_attention_heads=2, max_position_embeddings=77, projection_dim=proj or 64)
        cls = transformers.CLIPTextModelWithProjection if proj else transformers.CLIPTextModel
        want = {}
        for k, v in cls(hf_cfg).state_dict().items():
            if k.endswith("position_ids"):
                continue
            if not k.startswith(("text_model.", "text_projection.")):
                k = "text_model." + k            # transformers >= 5 drops the prefix for CLIPTextModel
            want[k] = tuple(v.shape)
        assert got == want
    bad = _lib.SdClipConfig(vocab_size=10, hidden_size=100, intermediate_size=128, num_layers=1, num_heads=2,
                            max_positions=77, hidden_act=0, projection_dim=0, layer_norm_eps=1e-5)
    h = C.c_void_p()
    assert engine_lib.sd_clip_create(C.byref(bad), C.byref(h)) == 4      # hidden % 64


def test_no_silent_cpu_fallback():
    """Without a HIP device the product path raises; nothing routes through the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = HipUNet2DConditionModel(config.tiny_unet())
    with pytest.raises(RuntimeError):
        net.load_state_dict({})
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 4, 16, 16), 1.0, torch.zeros(1, 77, 64))
    vae = HipAutoencoderKL(config.tiny_vae())
    with pytest.raises(RuntimeError):
        vae.decode(torch.zeros(1, 4, 8, 8))
    import stablediffusion_amd.models as m
    src = open(m.__file__).read()
    assert "oracle" not in src.replace("the CPU oracle", "")
