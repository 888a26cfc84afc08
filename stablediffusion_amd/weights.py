"""State-dict manifests (diffusers key names + shapes), seeded synthetic weights, LoRA fusing.

Key names follow the layout `/root/reference/scripts/convert_from_A1111.py` writes:
UNet `:283-317` (time_embedding / conv_in / conv_norm_out / conv_out), `:340-369` (down blocks),
`:67-69,371-385` (mid block), `:387-441` (up blocks), resnet member names `:206-225`;
VAE `:583-600` (conv_in/out, quant convs), `:644-660` (decoder up blocks, reverse index),
`:530-557` (mid attention to_q/to_k/to_v/to_out.0), `:522` (nin_shortcut -> conv_shortcut).
These names are the only thing the reference repo pins about the two models.
"""
from __future__ import annotations

from collections import OrderedDict
import math
from typing import Dict, Tuple

import torch

from .config import UNetConfig, VAEConfig

Manifest = "OrderedDict[str, Tuple[int, ...]]"


# ----------------------------------------------------------------------------------------------
# UNet2DConditionModel
# ----------------------------------------------------------------------------------------------
def _resnet(m, p, cin, cout, temb):
    m[f"{p}.norm1.weight"] = (cin,)
    m[f"{p}.norm1.bias"] = (cin,)
    m[f"{p}.conv1.weight"] = (cout, cin, 3, 3)
    m[f"{p}.conv1.bias"] = (cout,)
    if temb:
        m[f"{p}.time_emb_proj.weight"] = (cout, temb)
        m[f"{p}.time_emb_proj.bias"] = (cout,)
    m[f"{p}.norm2.weight"] = (cout,)
    m[f"{p}.norm2.bias"] = (cout,)
    m[f"{p}.conv2.weight"] = (cout, cout, 3, 3)
    m[f"{p}.conv2.bias"] = (cout,)
    if cin != cout:
        m[f"{p}.conv_shortcut.weight"] = (cout, cin, 1, 1)
        m[f"{p}.conv_shortcut.bias"] = (cout,)


def _transformer(m, p, c, depth, ctx, linear):
    m[f"{p}.norm.weight"] = (c,)
    m[f"{p}.norm.bias"] = (c,)
    m[f"{p}.proj_in.weight"] = (c, c) if linear else (c, c, 1, 1)
    m[f"{p}.proj_in.bias"] = (c,)
    for d in range(depth):
        b = f"{p}.transformer_blocks.{d}"
        m[f"{b}.norm1.weight"] = (c,)
        m[f"{b}.norm1.bias"] = (c,)
        m[f"{b}.attn1.to_q.weight"] = (c, c)
        m[f"{b}.attn1.to_k.weight"] = (c, c)
        m[f"{b}.attn1.to_v.weight"] = (c, c)
        m[f"{b}.attn1.to_out.0.weight"] = (c, c)
        m[f"{b}.attn1.to_out.0.bias"] = (c,)
        m[f"{b}.norm2.weight"] = (c,)
        m[f"{b}.norm2.bias"] = (c,)
        m[f"{b}.attn2.to_q.weight"] = (c, c)
        m[f"{b}.attn2.to_k.weight"] = (c, ctx)
        m[f"{b}.attn2.to_v.weight"] = (c, ctx)
        m[f"{b}.attn2.to_out.0.weight"] = (c, c)
        m[f"{b}.attn2.to_out.0.bias"] = (c,)
        m[f"{b}.norm3.weight"] = (c,)
        m[f"{b}.norm3.bias"] = (c,)
        m[f"{b}.ff.net.0.proj.weight"] = (8 * c, c)
        m[f"{b}.ff.net.0.proj.bias"] = (8 * c,)
        m[f"{b}.ff.net.2.weight"] = (c, 4 * c)
        m[f"{b}.ff.net.2.bias"] = (c,)
    m[f"{p}.proj_out.weight"] = (c, c) if linear else (c, c, 1, 1)
    m[f"{p}.proj_out.bias"] = (c,)


def unet_manifest(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    m: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = cfg.block_out_channels
    temb = cfg.time_embed_dim
    ctx = cfg.cross_attention_dim
    lin = cfg.use_linear_projection
    m["conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
    m["conv_in.bias"] = (boc[0],)
    m["time_embedding.linear_1.weight"] = (temb, boc[0])
    m["time_embedding.linear_1.bias"] = (temb,)
    m["time_embedding.linear_2.weight"] = (temb, temb)
    m["time_embedding.linear_2.bias"] = (temb,)
    if cfg.addition_embed_type == "text_time":
        pin = cfg.projection_class_embeddings_input_dim
        m["add_embedding.linear_1.weight"] = (temb, pin)
        m["add_embedding.linear_1.bias"] = (temb,)
        m["add_embedding.linear_2.weight"] = (temb, temb)
        m["add_embedding.linear_2.bias"] = (temb,)
    # down blocks
    out_ch = boc[0]
    nblk = len(boc)
    for i, btype in enumerate(cfg.down_block_types):
        in_ch, out_ch = out_ch, boc[i]
        for j in range(cfg.layers_per_block):
            _resnet(m, f"down_blocks.{i}.resnets.{j}", in_ch if j == 0 else out_ch, out_ch, temb)
            if btype == "CrossAttnDownBlock2D":
                _transformer(m, f"down_blocks.{i}.attentions.{j}", out_ch,
                             cfg.transformer_layers_per_block[i], ctx, lin)
        if i != nblk - 1:
            m[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (out_ch, out_ch, 3, 3)
            m[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (out_ch,)
    # mid block
    mid = boc[-1]
    _resnet(m, "mid_block.resnets.0", mid, mid, temb)
    _transformer(m, "mid_block.attentions.0", mid, cfg.transformer_layers_per_block[-1], ctx, lin)
    _resnet(m, "mid_block.resnets.1", mid, mid, temb)
    # up blocks
    rev = list(reversed(boc))
    rev_depth = list(reversed(cfg.transformer_layers_per_block))
    out_ch = rev[0]
    for i, btype in enumerate(cfg.up_block_types):
        prev_out = out_ch
        out_ch = rev[i]
        in_ch = rev[min(i + 1, nblk - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = in_ch if j == cfg.layers_per_block else out_ch
            rin = prev_out if j == 0 else out_ch
            _resnet(m, f"up_blocks.{i}.resnets.{j}", rin + skip, out_ch, temb)
            if btype == "CrossAttnUpBlock2D":
                _transformer(m, f"up_blocks.{i}.attentions.{j}", out_ch, rev_depth[i], ctx, lin)
        if i != nblk - 1:
            m[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (out_ch, out_ch, 3, 3)
            m[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (out_ch,)
    m["conv_norm_out.weight"] = (boc[0],)
    m["conv_norm_out.bias"] = (boc[0],)
    m["conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3)
    m["conv_out.bias"] = (cfg.out_channels,)
    return m


# ----------------------------------------------------------------------------------------------
# AutoencoderKL
# ----------------------------------------------------------------------------------------------
def _vae_attn(m, p, c):
    m[f"{p}.group_norm.weight"] = (c,)
    m[f"{p}.group_norm.bias"] = (c,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        m[f"{p}.{n}.weight"] = (c, c)
        m[f"{p}.{n}.bias"] = (c,)


def vae_decoder_manifest(cfg: VAEConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    m: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = cfg.block_out_channels
    lc = cfg.latent_channels
    m["post_quant_conv.weight"] = (lc, lc, 1, 1)
    m["post_quant_conv.bias"] = (lc,)
    top = boc[-1]
    m["decoder.conv_in.weight"] = (top, lc, 3, 3)
    m["decoder.conv_in.bias"] = (top,)
    _resnet(m, "decoder.mid_block.resnets.0", top, top, 0)
    _vae_attn(m, "decoder.mid_block.attentions.0", top)
    _resnet(m, "decoder.mid_block.resnets.1", top, top, 0)
    rev = list(reversed(boc))
    out_ch = rev[0]
    for i in range(len(boc)):
        prev = out_ch
        out_ch = rev[i]
        for j in range(cfg.layers_per_block + 1):
            _resnet(m, f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else out_ch, out_ch, 0)
        if i != len(boc) - 1:
            m[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (out_ch, out_ch, 3, 3)
            m[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (out_ch,)
    m["decoder.conv_norm_out.weight"] = (boc[0],)
    m["decoder.conv_norm_out.bias"] = (boc[0],)
    m["decoder.conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3)
    m["decoder.conv_out.bias"] = (cfg.out_channels,)
    return m


def vae_encoder_manifest(cfg: VAEConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    m: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    boc = cfg.block_out_channels
    lc = cfg.latent_channels
    m["encoder.conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
    m["encoder.conv_in.bias"] = (boc[0],)
    out_ch = boc[0]
    for i in range(len(boc)):
        in_ch, out_ch = out_ch, boc[i]
        for j in range(cfg.layers_per_block):
            _resnet(m, f"encoder.down_blocks.{i}.resnets.{j}", in_ch if j == 0 else out_ch, out_ch, 0)
        if i != len(boc) - 1:
            m[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (out_ch, out_ch, 3, 3)
            m[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (out_ch,)
    top = boc[-1]
    _resnet(m, "encoder.mid_block.resnets.0", top, top, 0)
    _vae_attn(m, "encoder.mid_block.attentions.0", top)
    _resnet(m, "encoder.mid_block.resnets.1", top, top, 0)
    m["encoder.conv_norm_out.weight"] = (top,)
    m["encoder.conv_norm_out.bias"] = (top,)
    m["encoder.conv_out.weight"] = (2 * lc, top, 3, 3)
    m["encoder.conv_out.bias"] = (2 * lc,)
    m["quant_conv.weight"] = (2 * lc, 2 * lc, 1, 1)
    m["quant_conv.bias"] = (2 * lc,)
    return m


def vae_manifest(cfg: VAEConfig):
    m = vae_encoder_manifest(cfg)
    m.update(vae_decoder_manifest(cfg))
    return m


def param_count(manifest) -> int:
    n = 0
    for shape in manifest.values():
        k = 1
        for s in shape:
            k *= s
        n += k
    return n


# ----------------------------------------------------------------------------------------------
# Seeded synthetic weights (N(0, 1/fan_in) conv/linear, norm affine (1,0), bias 0)
# ----------------------------------------------------------------------------------------------
def synth_state_dict(manifest, seed: int = 0, dtype=torch.float32, perturb: float = 0.0,
                     gain: float = 1.0, profile: str = "") -> Dict[str, torch.Tensor]:
    """Deterministic synthetic weights for a manifest.

    perturb=0 gives exactly the BASELINE.md §4 recipe.  perturb>0 additionally draws norm
    affines as (1 + p*N, p*N) and biases as p*N so that parity tests exercise every bias /
    affine path instead of multiplying by one and adding zero.

    profile="heavy_tail": checkpoint-like dynamic range for the fp16 stress tests, since real weights
    cannot be had offline -- 1 % of the output channels of every conv / linear behind a norm are outliers (rows x 30), biases ~ N(0, 0.5), norm gains log-uniform in [0.2, 3] with shifts ~ N(0, 0.5): activations are no longer O(1)
    everywhere (group statistics dominated by single channels, residual streams in the hundreds, saturating softmax rows).
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    heavy = profile.startswith("heavy_tail")
    if profile and not heavy:
        raise ValueError(f"unknown weight profile {profile!r}")
    outlier = float(profile.split(":")[1]) if ":" in profile else 30.0        # "heavy_tail:<row scale>"
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in manifest.items():
        is_norm = (".norm" in key or "group_norm" in key or "conv_norm_out" in key
                   or key.startswith("norm"))
        if len(shape) == 1:
            if key.endswith(".weight") and is_norm:
                t = torch.ones(shape)
                if heavy:
                    t = torch.exp(torch.rand(shape, generator=g) * (math.log(3.0) - math.log(0.2)) + math.log(0.2))
                elif perturb:
                    t = t + perturb * torch.randn(shape, generator=g)
            else:
                t = torch.zeros(shape)
                if heavy:
                    t = 0.5 * torch.randn(shape, generator=g)
                elif perturb:
                    t = perturb * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g) * (gain / fan_in ** 0.5)
            # (not on the layers that map a residual stream onto itself without a norm in front -- shortcuts, up / down
            # samplers: outliers there compound multiplicatively, 30^7 over the up path, which no trained network does)
            if heavy and shape[0] >= 100 and not any(t_ in key for t_ in ("conv_shortcut", "upsamplers", "downsamplers")):
                rows = torch.randperm(shape[0], generator=g)[: max(1, shape[0] // 100)]
                t[rows] *= outlier
        sd[key] = t.to(dtype)
    return sd


# ----------------------------------------------------------------------------------------------
# LoRA fusing (BASELINE.json config 5; reference runtime path: stable_diffusion.py:252-309,
# producer train_lora_pipeline.py:247-252,496-528).  W' = W + scale * (alpha/r) * up @ down.
# ----------------------------------------------------------------------------------------------
def fuse_lora(unet_sd: Dict[str, torch.Tensor], lora_sd: Dict[str, torch.Tensor],
              adapter_weight: float = 1.0, alpha_over_r: float = 1.0) -> Dict[str, torch.Tensor]:
    """Fold LoRA A/B pairs into the UNet's Linear weights once on the host.

    Accepts the two key spellings diffusers has written for UNet LoRA files:
      unet.<module>.lora.down.weight / .lora.up.weight          (diffusers "old" format)
      unet.<module>.lora_A.weight    / .lora_B.weight           (peft format)
    `<module>` is the diffusers module path, e.g. down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q
    The trainer in the reference uses r=16, lora_alpha=r (train_lora_pipeline.py:247-252) so the
    scale is adapter_weight * 1.
    """
    out = dict(unet_sd)
    pairs: Dict[str, Dict[str, torch.Tensor]] = {}
    for k, v in lora_sd.items():
        kk = k[5:] if k.startswith("unet.") else k
        for down_tag, up_tag in ((".lora.down.weight", ".lora.up.weight"),
                                 (".lora_A.weight", ".lora_B.weight")):
            if kk.endswith(down_tag):
                pairs.setdefault(kk[: -len(down_tag)], {})["down"] = v
            elif kk.endswith(up_tag):
                pairs.setdefault(kk[: -len(up_tag)], {})["up"] = v
    for mod, p in pairs.items():
        if "down" not in p or "up" not in p:
            raise KeyError(f"LoRA pair incomplete for {mod}")
        wkey = mod + ".weight"
        if wkey not in out:
            raise KeyError(f"LoRA targets unknown module {mod}")
        w = out[wkey]
        delta = (p["up"].float() @ p["down"].float()) * (adapter_weight * alpha_over_r)
        out[wkey] = (w.float() + delta.reshape(w.shape)).to(w.dtype)
    return out
