"""ORACLE -- test infrastructure only.  fp32 CPU restatement of `AutoencoderKL.decode` / `.encode`.

PARITY UNPINNED (same reason as oracle/unet_ref.py): diffusers==0.27.2 is un-vendored and absent;
the reference has no fixtures for this path.  Restates diffusers 0.27.2
`models/autoencoders/autoencoder_kl.py` + `models/autoencoders/vae.py` (Encoder / Decoder,
UNetMidBlock2D, UpDecoderBlock2D, DownEncoderBlock2D) for the call sites
`/root/reference/pipelines/sd_unified_pipeline.py:523` (decode) and `:1027-1032` (encode), with the
config / key layout the reference's converter states (`convert_from_A1111.py:490-511`, `:572-677`).
All VAE norms use eps=1e-6; the mid attention is one head over all channels with biased q/k/v.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .unet_ref import resnet_block, _conv, _lin

VAE_EPS = 1e-6


def vae_attention(x, w, p, groups=32):
    """Attention block of UNetMidBlock2D in the VAE (heads = 1, residual_connection, group_norm)."""
    B, C, H, W = x.shape
    res = x
    h = F.group_norm(x, groups, w[p + ".group_norm.weight"], w[p + ".group_norm.bias"], VAE_EPS)
    h = h.view(B, C, H * W).transpose(1, 2)
    q = _lin(h, w, p + ".to_q")
    k = _lin(h, w, p + ".to_k")
    v = _lin(h, w, p + ".to_v")
    o = F.scaled_dot_product_attention(q[:, None], k[:, None], v[:, None])[:, 0]
    o = _lin(o, w, p + ".to_out.0")
    o = o.transpose(1, 2).reshape(B, C, H, W)
    return o + res


def vae_decode(cfg, w, z: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.decode(z)[0]: z [B,4,h,w] (already divided by scaling_factor) -> [B,3,8h,8w]."""
    g = cfg.norm_num_groups
    boc = cfg.block_out_channels
    x = _conv(z, w, "post_quant_conv", padding=0)
    x = _conv(x, w, "decoder.conv_in")
    x = resnet_block(x, None, w, "decoder.mid_block.resnets.0", g, VAE_EPS)
    x = vae_attention(x, w, "decoder.mid_block.attentions.0", g)
    x = resnet_block(x, None, w, "decoder.mid_block.resnets.1", g, VAE_EPS)
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block + 1):
            x = resnet_block(x, None, w, f"decoder.up_blocks.{i}.resnets.{j}", g, VAE_EPS)
        if i != len(boc) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(x, w, f"decoder.up_blocks.{i}.upsamplers.0.conv")
    x = F.group_norm(x, g, w["decoder.conv_norm_out.weight"], w["decoder.conv_norm_out.bias"], VAE_EPS)
    x = F.silu(x)
    return _conv(x, w, "decoder.conv_out")


def vae_encode_moments(cfg, w, img: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.encode(x) up to the moments tensor [B, 2*latent, h/8, w/8] (mean, logvar).

    Downsample2D in the encoder pads (0,1,0,1) then conv stride 2 padding 0 (SURVEY.md §7 hard part 1).
    """
    g = cfg.norm_num_groups
    boc = cfg.block_out_channels
    x = _conv(img, w, "encoder.conv_in")
    for i in range(len(boc)):
        for j in range(cfg.layers_per_block):
            x = resnet_block(x, None, w, f"encoder.down_blocks.{i}.resnets.{j}", g, VAE_EPS)
        if i != len(boc) - 1:
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0.0)
            x = _conv(x, w, f"encoder.down_blocks.{i}.downsamplers.0.conv", stride=2, padding=0)
    x = resnet_block(x, None, w, "encoder.mid_block.resnets.0", g, VAE_EPS)
    x = vae_attention(x, w, "encoder.mid_block.attentions.0", g)
    x = resnet_block(x, None, w, "encoder.mid_block.resnets.1", g, VAE_EPS)
    x = F.group_norm(x, g, w["encoder.conv_norm_out.weight"], w["encoder.conv_norm_out.bias"], VAE_EPS)
    x = F.silu(x)
    x = _conv(x, w, "encoder.conv_out")
    return _conv(x, w, "quant_conv", padding=0)


def diag_gaussian_sample(moments: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """DiagonalGaussianDistribution.sample with externally supplied N(0,1) noise."""
    mean, logvar = moments.chunk(2, dim=1)
    logvar = logvar.clamp(-30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise
