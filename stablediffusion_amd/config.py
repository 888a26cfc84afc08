"""Model hyper-parameters for the denoise hot path, in diffusers field names.

The reference never spells these out itself: it loads them from hub checkpoints
(`/root/reference/models/stable_diffusion.py:110-123`).  The only in-repo statement of how they
are derived is the A1111 converter, `/root/reference/scripts/convert_from_A1111.py:97-203` (UNet)
and `:490-511` (VAE); the field names below are the ones that function writes (`:175-189`).
The pipeline reads a handful of them back through `model.base.config.*` / `model.vae.config.*`
(`/root/reference/pipelines/sd_unified_pipeline.py:176,220,418,513-521,1020`).
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import Optional, Tuple


@dataclass(frozen=True)
class UNetConfig:
    sample_size: int = 64
    in_channels: int = 4
    out_channels: int = 4
    down_block_types: Tuple[str, ...] = (
        "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D")
    up_block_types: Tuple[str, ...] = (
        "UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D")
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    # diffusers quirk kept on purpose: for SD1.5 `attention_head_dim=8` means *8 heads*
    # (SURVEY.md §7 hard part 1); a tuple gives heads per down block (SDXL: 5, 10, 20).
    attention_head_dim: Tuple[int, ...] = (8, 8, 8, 8)
    transformer_layers_per_block: Tuple[int, ...] = (1, 1, 1, 1)
    use_linear_projection: bool = False
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    flip_sin_to_cos: bool = True
    freq_shift: int = 0
    addition_embed_type: Optional[str] = None
    addition_time_embed_dim: Optional[int] = None
    projection_class_embeddings_input_dim: Optional[int] = None

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    def heads_for_block(self, i: int) -> int:
        return self.attention_head_dim[i]

    def to_dict(self):
        return asdict(self)


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215
    force_upcast: bool = False
    latents_mean: Optional[Tuple[float, ...]] = None
    latents_std: Optional[Tuple[float, ...]] = None
    sample_size: int = 512

    def to_dict(self):
        return asdict(self)


@dataclass(frozen=True)
class CLIPTextConfig:
    """transformers CLIPTextConfig fields the engine needs (the reference loads the encoders at
    /root/reference/models/stable_diffusion.py:124-152).  Defaults: CLIP ViT-L/14 text tower (SD1.5,
    and `text_encoder` of SDXL)."""
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    projection_dim: int = 0            # > 0: CLIPTextModelWithProjection
    layer_norm_eps: float = 1e-5
    eos_token_id: int = 2              # 2 = legacy configs: pooled token is ids.argmax (transformers modeling_clip)
    bos_token_id: int = 1
    pad_token_id: int = 1

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_hf(cls, hf, with_projection: bool = False):
        return cls(vocab_size=hf.vocab_size, hidden_size=hf.hidden_size, intermediate_size=hf.intermediate_size,
                   num_hidden_layers=hf.num_hidden_layers, num_attention_heads=hf.num_attention_heads,
                   max_position_embeddings=hf.max_position_embeddings, hidden_act=hf.hidden_act,
                   projection_dim=hf.projection_dim if with_projection else 0, layer_norm_eps=hf.layer_norm_eps,
                   eos_token_id=hf.eos_token_id, bos_token_id=hf.bos_token_id, pad_token_id=hf.pad_token_id)


def clip_l() -> CLIPTextConfig:
    return CLIPTextConfig()


def openclip_bigg() -> CLIPTextConfig:
    """SDXL `text_encoder_2` (OpenCLIP ViT-bigG/14 text tower, with projection)."""
    return CLIPTextConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=20,
                          hidden_act="gelu", projection_dim=1280)


def sd15_unet() -> UNetConfig:
    """SD1.5: 859 520 964 parameters (SURVEY.md §8c cross-check)."""
    return UNetConfig()


def sd15_vae() -> VAEConfig:
    return VAEConfig()


def sdxl_unet() -> UNetConfig:
    """SDXL-base (BASELINE.json config 4)."""
    return UNetConfig(
        sample_size=128,
        down_block_types=("DownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D"),
        up_block_types=("CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "UpBlock2D"),
        block_out_channels=(320, 640, 1280),
        cross_attention_dim=2048,
        attention_head_dim=(5, 10, 20),
        transformer_layers_per_block=(1, 2, 10),
        use_linear_projection=True,
        addition_embed_type="text_time",
        addition_time_embed_dim=256,
        projection_class_embeddings_input_dim=2816,
    )


def sdxl_vae() -> VAEConfig:
    return VAEConfig(scaling_factor=0.13025, force_upcast=True, sample_size=1024)


def tiny_unet(linear: bool = False, sdxl_cond: bool = False) -> UNetConfig:
    """Width-reduced UNet with the SD1.5 topology; used for golden fixtures and fast parity tests."""
    kw = dict(
        sample_size=16,
        block_out_channels=(64, 128, 256, 256),
        cross_attention_dim=64,
        attention_head_dim=(2, 4, 8, 8),
        use_linear_projection=linear,
    )
    if sdxl_cond:
        kw.update(addition_embed_type="text_time", addition_time_embed_dim=32,
                  projection_class_embeddings_input_dim=6 * 32 + 64)
    return UNetConfig(**kw)


def tiny_vae() -> VAEConfig:
    return VAEConfig(block_out_channels=(64, 64, 128, 128), sample_size=64)


PRESETS = {
    "sd15": (sd15_unet, sd15_vae),
    "sdxl": (sdxl_unet, sdxl_vae),
    "tiny": (tiny_unet, tiny_vae),
}
