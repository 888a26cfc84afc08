"""Structural conformance -- the only thing the reference repo pins about the two models
(SURVEY.md §4): public parameter totals and the diffusers state-dict key names the reference's
converter writes (/root/reference/scripts/convert_from_A1111.py:240-485, :572-677)."""
import ctypes as C

from stablediffusion_amd import _lib, config, weights
from stablediffusion_amd.models import HipAutoencoderKL, HipUNet2DConditionModel


def test_parameter_counts():
    assert weights.param_count(weights.unet_manifest(config.sd15_unet())) == 859_520_964
    assert weights.param_count(weights.unet_manifest(config.sdxl_unet())) == 2_567_463_684
    assert weights.param_count(weights.vae_manifest(config.sd15_vae())) == 83_653_863
    assert weights.param_count(weights.vae_decoder_manifest(config.sd15_vae())) == 49_490_199


def test_key_names_follow_converter_layout():
    m = weights.unet_manifest(config.sd15_unet())
    for k in [
        "time_embedding.linear_1.weight", "time_embedding.linear_2.bias", "conv_in.weight",       # :283-291
        "conv_norm_out.weight", "conv_out.bias",                                                   # :312-317
        "down_blocks.0.resnets.1.time_emb_proj.weight", "down_blocks.2.resnets.0.conv_shortcut.weight",  # :206-225
        "down_blocks.1.downsamplers.0.conv.weight",                                                # :349-355
        "down_blocks.0.attentions.1.transformer_blocks.0.attn2.to_k.weight",
        "mid_block.resnets.1.norm2.bias", "mid_block.attentions.0.proj_out.weight",                # :371-385
        "up_blocks.0.upsamplers.0.conv.weight", "up_blocks.3.attentions.2.transformer_blocks.0.ff.net.0.proj.bias",
        "up_blocks.3.resnets.2.conv_shortcut.bias",
    ]:
        assert k in m, k
    assert "down_blocks.3.attentions.0.norm.weight" not in m           # DownBlock2D has no attention
    assert "down_blocks.3.downsamplers.0.conv.weight" not in m
    assert m["up_blocks.2.resnets.0.conv1.weight"] == (640, 1920, 3, 3)  # 1280 hidden + 640 skip
    assert m["down_blocks.0.attentions.0.proj_in.weight"] == (320, 320, 1, 1)
    v = weights.vae_manifest(config.sd15_vae())
    for k in ["post_quant_conv.weight", "quant_conv.bias", "decoder.conv_in.weight",               # :583-600
              "decoder.mid_block.attentions.0.to_q.bias", "decoder.mid_block.attentions.0.to_out.0.weight",  # :530-557
              "decoder.mid_block.attentions.0.group_norm.weight",
              "decoder.up_blocks.2.resnets.0.conv_shortcut.weight",                                 # :522
              "decoder.up_blocks.0.upsamplers.0.conv.weight", "encoder.down_blocks.2.downsamplers.0.conv.bias",
              "encoder.conv_out.weight", "decoder.conv_norm_out.bias"]:
        assert k in v, k
    assert v["encoder.conv_out.weight"] == (8, 512, 3, 3)
    assert v["decoder.up_blocks.3.resnets.0.conv1.weight"] == (128, 256, 3, 3)
    sx = weights.unet_manifest(config.sdxl_unet())
    assert sx["add_embedding.linear_1.weight"] == (1280, 2816)
    assert sx["down_blocks.1.attentions.0.proj_in.weight"] == (640, 640)                            # linear proj
    assert "down_blocks.2.attentions.1.transformer_blocks.9.attn1.to_q.weight" in sx


def _engine_manifest(handle, lib, prefix):
    out = []
    for i in range(getattr(lib, f"sd_{prefix}_num_weights")(handle)):
        key, shape, ndim = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
        assert getattr(lib, f"sd_{prefix}_weight_info")(handle, i, C.byref(key), shape, C.byref(ndim)) == 0
        out.append((key.value.decode(), tuple(shape[j] for j in range(ndim.value))))
    return out


def test_engine_manifest_matches_python(engine_lib):
    """The C++ engine declares exactly the same keys / shapes / order (no GPU needed for this)."""
    for cfg in (config.sd15_unet(), config.sdxl_unet(), config.tiny_unet(), config.tiny_unet(True, True)):
        net = HipUNet2DConditionModel(cfg)
        assert _engine_manifest(net._h, engine_lib, "unet") == list(weights.unet_manifest(cfg).items())
    for cfg in (config.sd15_vae(), config.tiny_vae()):
        vae = HipAutoencoderKL(cfg)
        assert _engine_manifest(vae._h, engine_lib, "vae") == list(weights.vae_manifest(cfg).items())


def test_lora_fuse():
    import torch
    cfg = config.tiny_unet()
    sd = weights.synth_state_dict(weights.unet_manifest(cfg), seed=3)
    mod = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q"
    g = torch.Generator().manual_seed(0)
    down, up = torch.randn(16, 64, generator=g), torch.randn(64, 16, generator=g)
    fused = weights.fuse_lora(sd, {f"unet.{mod}.lora.down.weight": down, f"unet.{mod}.lora.up.weight": up}, 0.5)
    assert torch.allclose(fused[mod + ".weight"], sd[mod + ".weight"] + 0.5 * up @ down, atol=1e-5)
    fused2 = weights.fuse_lora(sd, {f"unet.{mod}.lora_A.weight": down, f"unet.{mod}.lora_B.weight": up}, 0.5)
    assert torch.equal(fused[mod + ".weight"], fused2[mod + ".weight"])
    x = torch.randn(5, 64, generator=g)     # y = Wx + s*B(Ax) (stable_diffusion.py:252-295 runtime path)
    assert torch.allclose(x @ fused[mod + ".weight"].t(), x @ sd[mod + ".weight"].t() + 0.5 * (x @ down.t()) @ up.t(),
                          atol=1e-4)
