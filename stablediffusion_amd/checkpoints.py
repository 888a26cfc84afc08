"""Checkpoint ingest for the engine: diffusers folders (safetensors) and A1111 / LDM single files.

The reference obtains its weights through `from_pretrained` on hub names
(`/root/reference/models/stable_diffusion.py:110-152`) and, for single-file checkpoints, through
`/root/reference/scripts/convert_from_A1111.py` (UNet key map `:240-485`, VAE key map `:572-677`,
resnet member renames `:206-225`, VAE attention renames `:530-557`).  This module restates those
two key maps as data-driven renames so a checkpoint can be handed to
`HipUNet2DConditionModel.load_state_dict` / `HipAutoencoderKL.load_state_dict` without diffusers
installed.  Only tensor *names* move; values are untouched except the VAE attention 1x1-conv
weights, which are viewed as linear weights (`[C,C,1,1] -> [C,C]`, converter `:560-569`).

SURVEY.md §8(f) rank 2.  Files are read with `safetensors` only (no pickle is ever loaded).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Tuple

import torch

from .config import UNetConfig, VAEConfig

UNET_PREFIX = "model.diffusion_model."
VAE_PREFIX = "first_stage_model."

_RESNET_RENAMES = (  # convert_from_A1111.py:206-225
    ("in_layers.0", "norm1"), ("in_layers.2", "conv1"), ("out_layers.0", "norm2"), ("out_layers.3", "conv2"),
    ("emb_layers.1", "time_emb_proj"), ("skip_connection", "conv_shortcut"))


def load_safetensors(path: str, device: str = "cpu") -> Dict[str, torch.Tensor]:
    from safetensors.torch import load_file
    return load_file(path, device=device)


# ---------------------------------------------------------------------------------------------
# diffusers folder layout: <root>/{unet,vae}/{config.json, diffusion_pytorch_model[.fp16].safetensors}
# ---------------------------------------------------------------------------------------------
def _as_tuple(v, n):
    return tuple(v) if isinstance(v, (list, tuple)) else (v,) * n


def unet_config_from_json(d: dict) -> UNetConfig:
    nb = len(d["block_out_channels"])
    heads = d.get("num_attention_heads") or d["attention_head_dim"]   # diffusers quirk: head_dim == heads
    return UNetConfig(
        sample_size=d.get("sample_size", 64), in_channels=d["in_channels"], out_channels=d["out_channels"],
        down_block_types=tuple(d["down_block_types"]), up_block_types=tuple(d["up_block_types"]),
        block_out_channels=tuple(d["block_out_channels"]), layers_per_block=d["layers_per_block"],
        cross_attention_dim=d["cross_attention_dim"], attention_head_dim=_as_tuple(heads, nb),
        transformer_layers_per_block=_as_tuple(d.get("transformer_layers_per_block", 1), nb),
        use_linear_projection=bool(d.get("use_linear_projection", False)),
        norm_num_groups=d.get("norm_num_groups", 32), norm_eps=d.get("norm_eps", 1e-5),
        flip_sin_to_cos=d.get("flip_sin_to_cos", True), freq_shift=d.get("freq_shift", 0),
        addition_embed_type=d.get("addition_embed_type"),
        addition_time_embed_dim=d.get("addition_time_embed_dim"),
        projection_class_embeddings_input_dim=d.get("projection_class_embeddings_input_dim"))


def vae_config_from_json(d: dict) -> VAEConfig:
    return VAEConfig(
        in_channels=d.get("in_channels", 3), out_channels=d.get("out_channels", 3),
        latent_channels=d.get("latent_channels", 4), block_out_channels=tuple(d["block_out_channels"]),
        layers_per_block=d.get("layers_per_block", 2), norm_num_groups=d.get("norm_num_groups", 32),
        scaling_factor=d.get("scaling_factor", 0.18215), force_upcast=bool(d.get("force_upcast", False)),
        latents_mean=tuple(d["latents_mean"]) if d.get("latents_mean") else None,
        latents_std=tuple(d["latents_std"]) if d.get("latents_std") else None,
        sample_size=d.get("sample_size", 512))


def _find_weights(folder: str) -> str:
    for name in ("diffusion_pytorch_model.fp16.safetensors", "diffusion_pytorch_model.safetensors"):
        p = os.path.join(folder, name)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"no diffusion_pytorch_model[.fp16].safetensors under {folder}")


def load_diffusers_folder(root: str) -> Tuple[UNetConfig, Dict[str, torch.Tensor], VAEConfig, Dict[str, torch.Tensor]]:
    """Reads the `unet/` and `vae/` sub-folders of a diffusers checkpoint directory."""
    with open(os.path.join(root, "unet", "config.json")) as f:
        ucfg = unet_config_from_json(json.load(f))
    with open(os.path.join(root, "vae", "config.json")) as f:
        vcfg = vae_config_from_json(json.load(f))
    usd = load_safetensors(_find_weights(os.path.join(root, "unet")))
    vsd = load_safetensors(_find_weights(os.path.join(root, "vae")))
    return ucfg, usd, vcfg, vsd


# ---------------------------------------------------------------------------------------------
# A1111 / LDM single file -> diffusers names
# ---------------------------------------------------------------------------------------------
def _rename_resnet(rest: str) -> str:
    for old, new in _RESNET_RENAMES:
        if rest.startswith(old):
            return new + rest[len(old):]
    return rest


def ldm_unet_key_map(cfg: UNetConfig) -> Dict[str, str]:
    """Prefix map {ldm module prefix -> diffusers module prefix} for the UNet
    (convert_from_A1111.py:283-441).  Resnet members are renamed separately."""
    m: Dict[str, str] = {
        "time_embed.0": "time_embedding.linear_1", "time_embed.2": "time_embedding.linear_2",
        "input_blocks.0.0": "conv_in", "out.0": "conv_norm_out", "out.2": "conv_out",
        "middle_block.0": "mid_block.resnets.0", "middle_block.1": "mid_block.attentions.0",
        "middle_block.2": "mid_block.resnets.1",
    }
    if cfg.addition_embed_type == "text_time":
        m["label_emb.0.0"] = "add_embedding.linear_1"
        m["label_emb.0.2"] = "add_embedding.linear_2"
    lpb = cfg.layers_per_block
    nb = len(cfg.block_out_channels)
    i = 1
    for b, btype in enumerate(cfg.down_block_types):
        for l in range(lpb):
            m[f"input_blocks.{i}.0"] = f"down_blocks.{b}.resnets.{l}"
            if btype == "CrossAttnDownBlock2D":
                m[f"input_blocks.{i}.1"] = f"down_blocks.{b}.attentions.{l}"
            i += 1
        if b != nb - 1:
            m[f"input_blocks.{i}.0.op"] = f"down_blocks.{b}.downsamplers.0.conv"
            i += 1
    i = 0
    for b, btype in enumerate(cfg.up_block_types):
        for l in range(lpb + 1):
            m[f"output_blocks.{i}.0"] = f"up_blocks.{b}.resnets.{l}"
            sub = 1
            if btype == "CrossAttnUpBlock2D":
                m[f"output_blocks.{i}.1"] = f"up_blocks.{b}.attentions.{l}"
                sub = 2
            if l == lpb and b != nb - 1:
                m[f"output_blocks.{i}.{sub}.conv"] = f"up_blocks.{b}.upsamplers.0.conv"
            i += 1
    return m


def _apply_prefix_map(key: str, pmap: Dict[str, str], resnet_targets) -> Optional[str]:
    # longest prefix wins ("input_blocks.3.0.op" before "input_blocks.3.0")
    best = None
    for old in pmap:
        if (key == old or key.startswith(old + ".")) and (best is None or len(old) > len(best)):
            best = old
    if best is None:
        return None
    new = pmap[best]
    rest = key[len(best) + 1:]
    if ".resnets." in new and new in resnet_targets:
        rest = _rename_resnet(rest)
    return new + ("." + rest if rest else "")


def ldm_to_diffusers_unet(sd: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    pmap = ldm_unet_key_map(cfg)
    resnets = {v for v in pmap.values() if ".resnets." in v}
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if not k.startswith(UNET_PREFIX):
            continue
        nk = _apply_prefix_map(k[len(UNET_PREFIX):], pmap, resnets)
        if nk is None:
            raise KeyError(f"unmapped UNet key: {k}")
        out[nk] = v
    return out


def ldm_vae_key_map(cfg: VAEConfig) -> Dict[str, str]:
    """convert_from_A1111.py:583-677 (decoder up blocks are stored in reverse order, :644-660)."""
    nb = len(cfg.block_out_channels)
    lpb = cfg.layers_per_block
    m: Dict[str, str] = {"quant_conv": "quant_conv", "post_quant_conv": "post_quant_conv"}
    for side in ("encoder", "decoder"):
        m[f"{side}.conv_in"] = f"{side}.conv_in"
        m[f"{side}.conv_out"] = f"{side}.conv_out"
        m[f"{side}.norm_out"] = f"{side}.conv_norm_out"
        m[f"{side}.mid.block_1"] = f"{side}.mid_block.resnets.0"
        m[f"{side}.mid.attn_1"] = f"{side}.mid_block.attentions.0"
        m[f"{side}.mid.block_2"] = f"{side}.mid_block.resnets.1"
    for b in range(nb):
        for l in range(lpb):
            m[f"encoder.down.{b}.block.{l}"] = f"encoder.down_blocks.{b}.resnets.{l}"
        if b != nb - 1:
            m[f"encoder.down.{b}.downsample.conv"] = f"encoder.down_blocks.{b}.downsamplers.0.conv"
        for l in range(lpb + 1):
            m[f"decoder.up.{nb - 1 - b}.block.{l}"] = f"decoder.up_blocks.{b}.resnets.{l}"
        if b != nb - 1:
            m[f"decoder.up.{nb - 1 - b}.upsample.conv"] = f"decoder.up_blocks.{b}.upsamplers.0.conv"
    return m


_VAE_ATTN_RENAMES = (("norm", "group_norm"), ("q", "to_q"), ("k", "to_k"), ("v", "to_v"), ("proj_out", "to_out.0"))


def ldm_to_diffusers_vae(sd: Dict[str, torch.Tensor], cfg: VAEConfig) -> Dict[str, torch.Tensor]:
    pmap = ldm_vae_key_map(cfg)
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if not k.startswith(VAE_PREFIX):
            continue
        key = k[len(VAE_PREFIX):]
        best = None
        for old in pmap:
            if (key == old or key.startswith(old + ".")) and (best is None or len(old) > len(best)):
                best = old
        if best is None:
            raise KeyError(f"unmapped VAE key: {k}")
        new, rest = pmap[best], key[len(best) + 1:]
        if ".resnets." in new:
            rest = "conv_shortcut" + rest[len("nin_shortcut"):] if rest.startswith("nin_shortcut") else rest
        elif ".attentions." in new:
            for old_n, new_n in _VAE_ATTN_RENAMES:
                if rest == old_n + ".weight" or rest == old_n + ".bias":
                    rest = new_n + rest[len(old_n):]
                    break
            if rest.endswith(".weight") and v.ndim == 4:       # 1x1 conv -> linear (converter :560-569)
                v = v.reshape(v.shape[0], v.shape[1])
        out[new + "." + rest] = v
    return out


def load_ldm_single_file(path: str, unet_cfg: UNetConfig, vae_cfg: VAEConfig):
    """A1111-style `.safetensors` single file -> (unet_sd, vae_sd) in diffusers naming.
    (`.ckpt` pickles are refused on purpose: nothing here unpickles.)"""
    if not path.endswith(".safetensors"):
        raise ValueError("only .safetensors single files are read (pickled .ckpt files are not loaded)")
    sd = load_safetensors(path)
    return ldm_to_diffusers_unet(sd, unet_cfg), ldm_to_diffusers_vae(sd, vae_cfg)
