"""Checkpoint ingest: diffusers-folder reader and the A1111/LDM key maps of
/root/reference/scripts/convert_from_A1111.py:240-485 (UNet), :572-677 (VAE), exercised on synthetic
files written here (no real checkpoint exists offline).  The LDM-side names in this test are written
out independently of the loader's prefix tables."""
import json
import re

import pytest
import torch

from stablediffusion_amd import checkpoints as ck
from stablediffusion_amd import config, weights


def diffusers_to_ldm_unet(sd, cfg):
    """Independent inverse map: diffusers UNet names -> LDM names (by formula, not by table)."""
    lpb = cfg.layers_per_block
    res = {"norm1": "in_layers.0", "conv1": "in_layers.2", "norm2": "out_layers.0", "conv2": "out_layers.3",
           "time_emb_proj": "emb_layers.1", "conv_shortcut": "skip_connection"}
    out = {}
    for k, v in sd.items():
        m = re.match(r"(down|up)_blocks\.(\d+)\.(resnets|attentions)\.(\d+)\.(.*)", k)
        if k.startswith("time_embedding.linear_"):
            nk = f"time_embed.{0 if k.split('.')[1] == 'linear_1' else 2}." + k.split(".")[-1]
        elif k.startswith("add_embedding.linear_"):
            nk = f"label_emb.0.{0 if k.split('.')[1] == 'linear_1' else 2}." + k.split(".")[-1]
        elif k.startswith("conv_in."):
            nk = "input_blocks.0.0." + k.split(".")[-1]
        elif k.startswith("conv_norm_out."):
            nk = "out.0." + k.split(".")[-1]
        elif k.startswith("conv_out."):
            nk = "out.2." + k.split(".")[-1]
        elif k.startswith("mid_block."):
            kind, idx, rest = re.match(r"mid_block\.(resnets|attentions)\.(\d+)\.(.*)", k).groups()
            pos = {("resnets", "0"): 0, ("attentions", "0"): 1, ("resnets", "1"): 2}[(kind, idx)]
            if kind == "resnets":
                head, tail = rest.split(".", 1)
                rest = res[head] + "." + tail
            nk = f"middle_block.{pos}.{rest}"
        elif m:
            side, b, kind, l, rest = m.groups()
            b, l = int(b), int(l)
            if kind == "resnets":
                head, tail = rest.split(".", 1)
                rest = res[head] + "." + tail
            sub = 0 if kind == "resnets" else 1
            if side == "down":
                nk = f"input_blocks.{1 + b * (lpb + 1) + l}.{sub}.{rest}"
            else:
                nk = f"output_blocks.{b * (lpb + 1) + l}.{sub}.{rest}"
        elif ".downsamplers.0.conv." in k:
            b = int(k.split(".")[1])
            nk = f"input_blocks.{1 + b * (lpb + 1) + lpb}.0.op." + k.split(".")[-1]
        elif ".upsamplers.0.conv." in k:
            b = int(k.split(".")[1])
            sub = 2 if cfg.up_block_types[b] == "CrossAttnUpBlock2D" else 1
            nk = f"output_blocks.{b * (lpb + 1) + lpb}.{sub}.conv." + k.split(".")[-1]
        else:
            raise AssertionError(k)
        out[ck.UNET_PREFIX + nk] = v
    return out


def diffusers_to_ldm_vae(sd, cfg):
    nb = len(cfg.block_out_channels)
    attn = {"group_norm": "norm", "to_q": "q", "to_k": "k", "to_v": "v", "to_out.0": "proj_out"}
    out = {}
    for k, v in sd.items():
        nk = k
        nk = re.sub(r"^(encoder|decoder)\.conv_norm_out\.", r"\1.norm_out.", nk)
        nk = re.sub(r"^(encoder|decoder)\.mid_block\.resnets\.(\d)\.", lambda m: f"{m.group(1)}.mid.block_{int(m.group(2)) + 1}.", nk)
        nk = re.sub(r"^(encoder|decoder)\.mid_block\.attentions\.0\.", r"\1.mid.attn_1.", nk)
        nk = re.sub(r"^encoder\.down_blocks\.(\d+)\.resnets\.(\d+)\.", r"encoder.down.\1.block.\2.", nk)
        nk = re.sub(r"^encoder\.down_blocks\.(\d+)\.downsamplers\.0\.conv\.", r"encoder.down.\1.downsample.conv.", nk)
        nk = re.sub(r"^decoder\.up_blocks\.(\d+)\.resnets\.(\d+)\.", lambda m: f"decoder.up.{nb - 1 - int(m.group(1))}.block.{m.group(2)}.", nk)
        nk = re.sub(r"^decoder\.up_blocks\.(\d+)\.upsamplers\.0\.conv\.", lambda m: f"decoder.up.{nb - 1 - int(m.group(1))}.upsample.conv.", nk)
        nk = nk.replace("conv_shortcut", "nin_shortcut")
        if ".attn_1." in nk:
            for d, l in attn.items():
                if f".attn_1.{d}." in nk:
                    nk = nk.replace(f".attn_1.{d}.", f".attn_1.{l}.")
                    if nk.endswith(".weight") and d != "group_norm":
                        v = v.reshape(v.shape[0], v.shape[1], 1, 1)       # LDM stores 1x1 convs
                    break
        out[ck.VAE_PREFIX + nk] = v
    return out


@pytest.mark.parametrize("ucfg", [config.tiny_unet(), config.tiny_unet(linear=True, sdxl_cond=True)])
def test_ldm_single_file_roundtrip(tmp_path, ucfg):
    from safetensors.torch import save_file
    vcfg = config.tiny_vae()
    usd = weights.synth_state_dict(weights.unet_manifest(ucfg), seed=1, dtype=torch.float16)
    vsd = weights.synth_state_dict(weights.vae_manifest(vcfg), seed=2, dtype=torch.float16)
    ldm = {**diffusers_to_ldm_unet(usd, ucfg), **diffusers_to_ldm_vae(vsd, vcfg),
           "cond_stage_model.transformer.text_model.embeddings.position_ids": torch.zeros(1, 77)}
    assert "model.diffusion_model.input_blocks.1.0.in_layers.2.weight" in ldm        # converter :206-225 names
    assert "model.diffusion_model.middle_block.1.proj_in.weight" in ldm
    assert "first_stage_model.decoder.up.3.block.0.norm1.weight" in ldm              # reverse index :644-660
    assert "first_stage_model.encoder.mid.attn_1.q.weight" in ldm
    path = str(tmp_path / "model.safetensors")
    save_file({k: v.contiguous() for k, v in ldm.items()}, path)
    u2, v2 = ck.load_ldm_single_file(path, ucfg, vcfg)
    assert list(sorted(u2)) == list(sorted(usd)) and list(sorted(v2)) == list(sorted(vsd))
    for k in usd:
        assert torch.equal(u2[k], usd[k]), k
    for k in vsd:
        assert u2 is not None and v2[k].shape == vsd[k].shape and torch.equal(v2[k], vsd[k]), k
    with pytest.raises(ValueError):
        ck.load_ldm_single_file(str(tmp_path / "model.ckpt"), ucfg, vcfg)
    bad = dict(ldm)
    bad["model.diffusion_model.input_blocks.99.0.in_layers.0.weight"] = torch.zeros(1)
    with pytest.raises(KeyError):
        ck.ldm_to_diffusers_unet(bad, ucfg)


def test_diffusers_folder_reader(tmp_path):
    from safetensors.torch import save_file
    ucfg, vcfg = config.sdxl_unet(), config.sd15_vae()
    tiny_u, tiny_v = config.tiny_unet(), config.tiny_vae()
    for sub, cfg, man, fname in (("unet", tiny_u, weights.unet_manifest(tiny_u), "diffusion_pytorch_model.fp16.safetensors"),
                                 ("vae", tiny_v, weights.vae_manifest(tiny_v), "diffusion_pytorch_model.safetensors")):
        (tmp_path / sub).mkdir()
        d = cfg.to_dict()
        if sub == "unet":
            d["attention_head_dim"] = list(d["attention_head_dim"])
        json.dump(d, open(tmp_path / sub / "config.json", "w"))
        save_file(weights.synth_state_dict(man, seed=3, dtype=torch.float16), str(tmp_path / sub / fname))
    u, usd, v, vsd = ck.load_diffusers_folder(str(tmp_path))
    assert u == tiny_u and v == tiny_v
    assert list(usd) and set(usd) == set(weights.unet_manifest(tiny_u)) and set(vsd) == set(weights.vae_manifest(tiny_v))
    # the hub configs' field spellings (scalar attention_head_dim = 8 heads, SDXL tuples)
    sd15_json = {"in_channels": 4, "out_channels": 4, "down_block_types": list(config.sd15_unet().down_block_types),
                 "up_block_types": list(config.sd15_unet().up_block_types), "block_out_channels": [320, 640, 1280, 1280],
                 "layers_per_block": 2, "cross_attention_dim": 768, "attention_head_dim": 8, "sample_size": 64}
    assert ck.unet_config_from_json(sd15_json) == config.sd15_unet()
    sdxl_json = dict(ucfg.to_dict(), attention_head_dim=[5, 10, 20], transformer_layers_per_block=[1, 2, 10])
    assert ck.unet_config_from_json(sdxl_json) == ucfg
    assert ck.vae_config_from_json({"block_out_channels": [128, 256, 512, 512]}) == vcfg
