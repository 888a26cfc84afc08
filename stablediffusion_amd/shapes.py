"""Enumerates the implicit-GEMM launches of one UNet forward / VAE decode (geometry + algorithmic
FLOPs), mirroring the graphs in csrc/unet.cpp and csrc/vae.cpp.  Used by tools/tune_igemm.py, by
DESIGN.md's FLOP accounting and by tests that cross-check SURVEY.md §8(d)'s per-unit figures."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

from .config import UNetConfig, VAEConfig


@dataclass(frozen=True)
class ConvShape:
    N: int
    H: int
    W: int
    Cin: int
    Cout: int
    ks: int = 1
    stride: int = 1
    up: int = 0
    geglu: int = 0
    tag: str = ""

    @property
    def out_hw(self):
        ih, iw = self.H << self.up, self.W << self.up
        if self.stride == 1:
            return ih, iw
        pad = 1 if self.ks == 3 else 0
        return (ih + 2 * pad - self.ks) // self.stride + 1, (iw + 2 * pad - self.ks) // self.stride + 1

    @property
    def M(self):
        oh, ow = self.out_hw
        return self.N * oh * ow

    @property
    def K(self):
        return self.ks * self.ks * self.Cin

    @property
    def flops(self):
        return 2.0 * self.M * self.Cout * self.K

    def key(self):
        return (self.N, self.H, self.W, self.Cin, self.Cout, self.ks, self.stride, self.up, self.geglu)


def _resnet(out, N, h, w, cin, cout):
    out.append(ConvShape(N, h, w, cin, cout, 3, tag="res.conv1"))
    out.append(ConvShape(N, h, w, cout, cout, 3, tag="res.conv2"))
    if cin != cout:
        out.append(ConvShape(N, h, w, cin, cout, 1, tag="res.shortcut"))


def _xformer(out, N, h, w, C, depth, L, ctx):
    out.append(ConvShape(N, h, w, C, C, 1, tag="proj_in"))
    for _ in range(depth):
        out.append(ConvShape(N, h, w, C, 3 * C, 1, tag="attn1.qkv"))
        out.append(ConvShape(N, h, w, C, C, 1, tag="attn1.out"))
        out.append(ConvShape(N, h, w, C, C, 1, tag="attn2.q"))
        out.append(ConvShape(N, L, 1, ctx, 2 * C, 1, tag="attn2.kv"))
        out.append(ConvShape(N, h, w, C, C, 1, tag="attn2.out"))
        out.append(ConvShape(N, h, w, C, 8 * C, 1, geglu=1, tag="ff.geglu"))
        out.append(ConvShape(N, h, w, 4 * C, C, 1, tag="ff.out"))
    out.append(ConvShape(N, h, w, C, C, 1, tag="proj_out"))


def unet_convs(cfg: UNetConfig, B: int, H: int, W: int, L: int = 77) -> List[ConvShape]:
    out: List[ConvShape] = []
    boc = cfg.block_out_channels
    nb = len(boc)
    ctx = cfg.cross_attention_dim
    h, w = H, W
    out.append(ConvShape(B, h, w, 64, boc[0], 1, tag="conv_in(im2col K=64)"))
    ch = boc[0]
    for i, bt in enumerate(cfg.down_block_types):
        cin, ch = ch, boc[i]
        for j in range(cfg.layers_per_block):
            _resnet(out, B, h, w, cin if j == 0 else ch, ch)
            if bt == "CrossAttnDownBlock2D":
                _xformer(out, B, h, w, ch, cfg.transformer_layers_per_block[i], L, ctx)
        if i != nb - 1:
            out.append(ConvShape(B, h, w, ch, ch, 3, stride=2, tag="downsample"))
            h, w = h // 2, w // 2
    _resnet(out, B, h, w, ch, ch)
    _xformer(out, B, h, w, ch, cfg.transformer_layers_per_block[-1], L, ctx)
    _resnet(out, B, h, w, ch, ch)
    rev = list(reversed(boc))
    rdepth = list(reversed(cfg.transformer_layers_per_block))
    ch = rev[0]
    for i, bt in enumerate(cfg.up_block_types):
        prev, ch = ch, rev[i]
        cin = rev[min(i + 1, nb - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = cin if j == cfg.layers_per_block else ch
            rin = prev if j == 0 else ch
            _resnet(out, B, h, w, rin + skip, ch)
            if bt == "CrossAttnUpBlock2D":
                _xformer(out, B, h, w, ch, rdepth[i], L, ctx)
        if i != nb - 1:
            out.append(ConvShape(B, h, w, ch, ch, 3, up=1, tag="upsample"))
            h, w = h * 2, w * 2
    out.append(ConvShape(B, h, w, boc[0], cfg.out_channels, 3, tag="conv_out"))
    return out


def unet_attention_flops(cfg: UNetConfig, B: int, H: int, W: int, L: int = 77) -> float:
    """4 * B * T * Tk * C per attention core (QK^T + PV), self and cross."""
    boc = cfg.block_out_channels
    nb = len(boc)
    total = 0.0
    h, w = H, W

    def add(C, depth):
        nonlocal total
        T = h * w
        total += depth * (4.0 * B * T * T * C + 4.0 * B * T * L * C)

    for i, bt in enumerate(cfg.down_block_types):
        if bt == "CrossAttnDownBlock2D":
            for _ in range(cfg.layers_per_block):
                add(boc[i], cfg.transformer_layers_per_block[i])
        if i != nb - 1:
            h, w = h // 2, w // 2
    add(boc[-1], cfg.transformer_layers_per_block[-1])
    rev = list(reversed(boc))
    rdepth = list(reversed(cfg.transformer_layers_per_block))
    for i, bt in enumerate(cfg.up_block_types):
        if bt == "CrossAttnUpBlock2D":
            for _ in range(cfg.layers_per_block + 1):
                add(rev[i], rdepth[i])
        if i != nb - 1:
            h, w = h * 2, w * 2
    return total


def vae_decoder_convs(cfg: VAEConfig, B: int, h: int, w: int) -> List[ConvShape]:
    out: List[ConvShape] = []
    boc = cfg.block_out_channels
    top = boc[-1]
    out.append(ConvShape(B, h, w, 64, top, 1, tag="conv_in(im2col K=64)"))
    _resnet(out, B, h, w, top, top)
    out.append(ConvShape(B, h, w, top, 3 * top, 1, tag="mid.qkv"))
    out.append(ConvShape(B, h, w, top, top, 1, tag="mid.attn_out"))
    _resnet(out, B, h, w, top, top)
    ch = top
    rev = list(reversed(boc))
    for i in range(len(boc)):
        prev, ch = ch, rev[i]
        for j in range(cfg.layers_per_block + 1):
            _resnet(out, B, h, w, prev if j == 0 else ch, ch)
        if i != len(boc) - 1:
            out.append(ConvShape(B, h, w, ch, ch, 3, up=1, tag="upsample"))
            h, w = h * 2, w * 2
    out.append(ConvShape(B, h, w, boc[0], cfg.out_channels, 3, tag="conv_out"))
    return out
