"""ORACLE -- test infrastructure only.  fp32 CPU restatement of `UNet2DConditionModel.forward`.

PARITY UNPINNED: the arithmetic restated here lives in the un-vendored third-party package
diffusers==0.27.2 (`/root/reference/requirements.txt:37`), which is not installed in the build
container and cannot be fetched; the reference holds no tests, golden tensors or hashes for it
(SURVEY.md §4, §8c).  The restatement follows the published diffusers 0.27.2 module structure
(`models/unets/unet_2d_condition.py`, `models/resnet.py`, `models/transformers/transformer_2d.py`,
`models/attention.py`, `models/attention_processor.py::AttnProcessor2_0`, `models/embeddings.py`,
`models/upsampling.py`, `models/downsampling.py`) and is anchored on what the reference *does*
pin: its call site `/root/reference/pipelines/sd_unified_pipeline.py:475-482`, the hyper-parameter
derivation `/root/reference/scripts/convert_from_A1111.py:97-203`, the state-dict key names that
converter writes (`:240-485`), and the public parameter total 859 520 964 (tests/test_manifest.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Weights come as a dict keyed by diffusers state-dict names (see stablediffusion_amd/weights.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F


def timestep_sinusoid(t: torch.Tensor, dim: int, flip_sin_to_cos: bool = True,
                      freq_shift: float = 0.0, max_period: float = 10000.0) -> torch.Tensor:
    """diffusers `get_timestep_embedding` (models/embeddings.py): f_i = exp(-ln(1e4) * i / (half - shift))."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    ang = t.float()[:, None] * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def _lin(x, w, p, bias=True):
    return F.linear(x, w[p + ".weight"], w[p + ".bias"] if bias else None)


def _conv(x, w, p, stride=1, padding=1):
    return F.conv2d(x, w[p + ".weight"], w[p + ".bias"], stride=stride, padding=padding)


def resnet_block(x, temb, w, p, groups=32, eps=1e-5):
    """ResnetBlock2D.forward, time_embedding_norm='default', output_scale_factor=1, dropout=0."""
    h = F.group_norm(x, groups, w[p + ".norm1.weight"], w[p + ".norm1.bias"], eps)
    h = F.silu(h)
    h = _conv(h, w, p + ".conv1")
    if temb is not None:
        h = h + _lin(F.silu(temb), w, p + ".time_emb_proj")[:, :, None, None]
    h = F.group_norm(h, groups, w[p + ".norm2.weight"], w[p + ".norm2.bias"], eps)
    h = F.silu(h)
    h = _conv(h, w, p + ".conv2")
    if (p + ".conv_shortcut.weight") in w:
        x = _conv(x, w, p + ".conv_shortcut", padding=0)
    return x + h


def attention(x, ctx, w, p, heads, qkv_bias=False):
    """Attention.forward with AttnProcessor2_0: softmax(q k^T / sqrt(d)) v, no mask, no dropout."""
    B, T, C = x.shape
    q = _lin(x, w, p + ".to_q", qkv_bias)
    k = _lin(ctx, w, p + ".to_k", qkv_bias)
    v = _lin(ctx, w, p + ".to_v", qkv_bias)
    d = C // heads
    q = q.view(B, -1, heads, d).transpose(1, 2)
    k = k.view(B, -1, heads, d).transpose(1, 2)
    v = v.view(B, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)  # scale = d ** -0.5
    o = o.transpose(1, 2).reshape(B, T, C)
    return _lin(o, w, p + ".to_out.0")


def basic_transformer_block(x, ctx, w, p, heads):
    h = F.layer_norm(x, (x.shape[-1],), w[p + ".norm1.weight"], w[p + ".norm1.bias"], 1e-5)
    x = x + attention(h, h, w, p + ".attn1", heads)
    h = F.layer_norm(x, (x.shape[-1],), w[p + ".norm2.weight"], w[p + ".norm2.bias"], 1e-5)
    x = x + attention(h, ctx, w, p + ".attn2", heads)
    h = F.layer_norm(x, (x.shape[-1],), w[p + ".norm3.weight"], w[p + ".norm3.bias"], 1e-5)
    proj = _lin(h, w, p + ".ff.net.0.proj")
    hidden, gate = proj.chunk(2, dim=-1)          # GEGLU: hidden * gelu(gate), exact-erf GELU
    h = hidden * F.gelu(gate)
    x = x + _lin(h, w, p + ".ff.net.2")
    return x


def transformer_2d(x, ctx, w, p, heads, depth, linear, groups=32):
    """Transformer2DModel.forward (continuous input)."""
    B, C, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, w[p + ".norm.weight"], w[p + ".norm.bias"], 1e-6)
    if not linear:
        h = _conv(h, w, p + ".proj_in", padding=0)
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    else:
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = _lin(h, w, p + ".proj_in")
    for d in range(depth):
        h = basic_transformer_block(h, ctx, w, f"{p}.transformer_blocks.{d}", heads)
    if not linear:
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = _conv(h, w, p + ".proj_out", padding=0)
    else:
        h = _lin(h, w, p + ".proj_out")
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return h + res


def unet_forward(cfg, w: Dict[str, torch.Tensor], sample: torch.Tensor, timestep,
                 encoder_hidden_states: torch.Tensor,
                 added_cond_kwargs: Optional[dict] = None) -> torch.Tensor:
    """Restates UNet2DConditionModel.forward for the call at sd_unified_pipeline.py:475-482.

    sample [B,4,h,w], timestep scalar or [B], encoder_hidden_states [B,L,D] -> [B,4,h,w] (fp32).
    """
    B = sample.shape[0]
    t = torch.as_tensor(timestep)
    if t.ndim == 0:
        t = t[None]
    t = t.expand(B)
    boc = cfg.block_out_channels
    g = cfg.norm_num_groups
    eps = cfg.norm_eps
    lin = cfg.use_linear_projection
    t_emb = timestep_sinusoid(t, boc[0], cfg.flip_sin_to_cos, cfg.freq_shift).to(sample.dtype)
    emb = _lin(F.silu(_lin(t_emb, w, "time_embedding.linear_1")), w, "time_embedding.linear_2")
    if cfg.addition_embed_type == "text_time":
        text_embeds = added_cond_kwargs["text_embeds"]
        time_ids = added_cond_kwargs["time_ids"]
        te = timestep_sinusoid(time_ids.flatten(), cfg.addition_time_embed_dim,
                               cfg.flip_sin_to_cos, cfg.freq_shift)
        te = te.reshape(B, -1).to(sample.dtype)
        add = torch.cat([text_embeds.to(sample.dtype), te], dim=-1)
        emb = emb + _lin(F.silu(_lin(add, w, "add_embedding.linear_1")), w, "add_embedding.linear_2")

    ctx = encoder_hidden_states
    x = _conv(sample, w, "conv_in")
    skips = [x]
    nblk = len(boc)
    for i, btype in enumerate(cfg.down_block_types):
        for j in range(cfg.layers_per_block):
            x = resnet_block(x, emb, w, f"down_blocks.{i}.resnets.{j}", g, eps)
            if btype == "CrossAttnDownBlock2D":
                x = transformer_2d(x, ctx, w, f"down_blocks.{i}.attentions.{j}",
                                   cfg.attention_head_dim[i], cfg.transformer_layers_per_block[i], lin, g)
            skips.append(x)
        if i != nblk - 1:
            x = _conv(x, w, f"down_blocks.{i}.downsamplers.0.conv", stride=2, padding=1)
            skips.append(x)

    x = resnet_block(x, emb, w, "mid_block.resnets.0", g, eps)
    x = transformer_2d(x, ctx, w, "mid_block.attentions.0", cfg.attention_head_dim[-1],
                       cfg.transformer_layers_per_block[-1], lin, g)
    x = resnet_block(x, emb, w, "mid_block.resnets.1", g, eps)

    rev_heads = list(reversed(cfg.attention_head_dim))
    rev_depth = list(reversed(cfg.transformer_layers_per_block))
    for i, btype in enumerate(cfg.up_block_types):
        for j in range(cfg.layers_per_block + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(x, emb, w, f"up_blocks.{i}.resnets.{j}", g, eps)
            if btype == "CrossAttnUpBlock2D":
                x = transformer_2d(x, ctx, w, f"up_blocks.{i}.attentions.{j}",
                                   rev_heads[i], rev_depth[i], lin, g)
        if i != nblk - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(x, w, f"up_blocks.{i}.upsamplers.0.conv")

    x = F.group_norm(x, g, w["conv_norm_out.weight"], w["conv_norm_out.bias"], eps)
    x = F.silu(x)
    x = _conv(x, w, "conv_out")
    return x
