"""Duck-typed stand-ins for the two model objects the reference keeps in `SDModelWrapper.base`
and `SDModelWrapper.vae` (`/root/reference/models/stable_diffusion.py:110-123`), backed by the
gfx950 engine through the C-ABI (`include/sd_engine.h`).

Surface kept:
  base(sample, t, ehs, cross_attention_kwargs=, added_cond_kwargs=, return_dict=False) -> (tensor,)
        sd_unified_pipeline.py:475-482
  base.config.{sample_size,in_channels,addition_time_embed_dim}   :176, :220, :418
  base.add_embedding.linear_1.in_features                          :419
  base.dtype, base.to(device)                                      :680, stable_diffusion.py:189
  vae.decode(z, return_dict=False) -> (tensor,)                    :523
  vae.encode(x).latent_dist.{sample(generator), mode()}            :98-106, :1027-1032
  vae.config.{scaling_factor,latents_mean,latents_std,force_upcast,latent_channels,block_out_channels}
  vae.to(device | dtype=)                                          stable_diffusion.py:188, :1022,1036
PyTorch is plumbing here (device buffers + the current HIP stream); all arithmetic is in the .so.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Dict, Optional

import torch

from . import _lib
from .config import UNetConfig, VAEConfig


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _as_f16(x: torch.Tensor, device) -> torch.Tensor:
    return x.to(device=device, dtype=torch.float16).contiguous()


class _Config(SimpleNamespace):
    def get(self, k, d=None):
        return getattr(self, k, d)


def _load_weights(lib, handle, prefix: str, state_dict: Dict[str, torch.Tensor], strict: bool = True):
    num = getattr(lib, f"sd_{prefix}_num_weights")(handle)
    info = getattr(lib, f"sd_{prefix}_weight_info")
    setw = getattr(lib, f"sd_{prefix}_set_weight")
    expected = []
    for i in range(num):
        key = C.c_char_p()
        shape = (C.c_int64 * 4)()
        ndim = C.c_int()
        _lib.check(info(handle, i, C.byref(key), shape, C.byref(ndim)), "weight_info")
        expected.append((key.value.decode(), tuple(shape[j] for j in range(ndim.value))))
    missing = [k for k, _ in expected if k not in state_dict]
    if missing:
        raise KeyError(f"state_dict is missing {len(missing)} keys, e.g. {missing[:3]}")
    if strict:
        names = {k for k, _ in expected}
        extra = [k for k in state_dict if k not in names]
        if extra:
            raise KeyError(f"state_dict has {len(extra)} unexpected keys, e.g. {extra[:3]}")
    for key, shape in expected:
        t = state_dict[key]
        if tuple(t.shape) != shape:
            raise ValueError(f"{key}: expected shape {shape}, got {tuple(t.shape)}")
        if t.dtype == torch.float16:
            code = _lib.SD_DTYPE_F16
        else:
            t = t.float()
            code = _lib.SD_DTYPE_F32
        t = t.contiguous()
        shp = (C.c_int64 * len(shape))(*shape)
        _lib.check(setw(handle, key.encode(), C.c_void_p(t.data_ptr()), shp, len(shape), code),
                   f"set_weight({key})")
    return expected


class HipUNet2DConditionModel:
    """gfx950 engine behind the `UNet2DConditionModel` call surface the reference uses."""

    def __init__(self, config: UNetConfig, device: str = "cuda"):
        self._lib = _lib.load()
        self.cfg = config
        self.device = torch.device(device)
        self.dtype = torch.float16
        nb = len(config.block_out_channels)
        c = _lib.SdUNetConfig()
        c.in_channels = config.in_channels
        c.out_channels = config.out_channels
        c.num_blocks = nb
        for i in range(nb):
            c.block_out_channels[i] = config.block_out_channels[i]
            c.down_block_has_attn[i] = int(config.down_block_types[i] == "CrossAttnDownBlock2D")
            c.up_block_has_attn[i] = int(config.up_block_types[i] == "CrossAttnUpBlock2D")
            c.num_heads[i] = config.attention_head_dim[i]
            c.transformer_layers[i] = config.transformer_layers_per_block[i]
        c.layers_per_block = config.layers_per_block
        c.cross_attention_dim = config.cross_attention_dim
        c.use_linear_projection = int(config.use_linear_projection)
        c.norm_num_groups = config.norm_num_groups
        c.norm_eps = config.norm_eps
        c.flip_sin_to_cos = int(config.flip_sin_to_cos)
        c.freq_shift = float(config.freq_shift)
        c.addition_time_embed_dim = config.addition_time_embed_dim or 0
        c.projection_class_embeddings_input_dim = config.projection_class_embeddings_input_dim or 0
        self._h = C.c_void_p()
        _lib.check(self._lib.sd_unet_create(C.byref(c), C.byref(self._h)), "sd_unet_create")
        self.config = _Config(**config.to_dict())
        if config.addition_embed_type == "text_time":
            self.add_embedding = SimpleNamespace(
                linear_1=SimpleNamespace(in_features=config.projection_class_embeddings_input_dim))
        self._finalized = False

    # -- weights -------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        _lib.require_gpu()
        with torch.cuda.device(self.device):
            _load_weights(self._lib, self._h, "unet", state_dict, strict)
            _lib.check(self._lib.sd_unet_finalize(self._h), "sd_unet_finalize")
        self._finalized = True
        return self

    def rebuild(self, state_dict: Dict[str, torch.Tensor]):
        """A fresh engine of the same configuration on the same device with other weights (LoRA re-fuse:
        the packed weights are immutable once finalized)."""
        return type(self)(self.cfg, self.device).load_state_dict(state_dict)

    def rebuild_factory(self):
        """`rebuild` without a reference to this engine: the LoRA re-fuse drops the old engine (and its device memory)
        BEFORE it packs the new one (peak memory was twice the UNet weights)."""
        cls, cfg, dev = type(self), self.cfg, self.device
        return lambda state_dict: cls(cfg, dev).load_state_dict(state_dict)

    def memory(self):
        w, s = C.c_int64(), C.c_int64()
        _lib.check(self._lib.sd_unet_memory(self._h, C.byref(w), C.byref(s)), "sd_unet_memory")
        return w.value, s.value

    def use_graph(self, enable: bool = True):
        """Replay the forward from a captured hipGraph (one host call per step instead of ~480)."""
        _lib.check(self._lib.sd_unet_use_graph(self._h, int(bool(enable))), "sd_unet_use_graph")
        self._graph_on = bool(enable)          # (carried over to the engine a LoRA re-fuse builds)
        return self

    def text_kv_cache(self, enable: bool = True):
        """Reuse the cross-attention K/V projections of `encoder_hidden_states` across the forwards of one
        denoise loop (same tensor every step).  Each call invalidates the cache; the pipeline calls it with
        True before its loop and with False after it."""
        _lib.check(self._lib.sd_unet_text_kv_cache(self._h, int(bool(enable))), "sd_unet_text_kv_cache")
        return self

    # -- reference surface ---------------------------------------------------------------------
    def to(self, device=None, dtype=None):
        if device is not None and torch.device(device).type != "cuda":
            raise _lib.EngineError("HipUNet2DConditionModel lives on the HIP device only")
        return self

    def eval(self):
        return self

    def __call__(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None,
                 added_cond_kwargs=None, return_dict=False, **unused):
        if not self._finalized:
            raise _lib.EngineError("weights not loaded")
        dev = self.device
        sample = _as_f16(sample, dev)
        B, _, H, W = sample.shape
        ehs = _as_f16(encoder_hidden_states, dev)
        if ehs.shape[0] != B:
            raise ValueError(f"encoder_hidden_states batch {ehs.shape[0]} != sample batch {B}")
        if ehs.shape[2] != self.cfg.cross_attention_dim:
            raise ValueError("encoder_hidden_states width != cross_attention_dim")
        if isinstance(timestep, (int, float)):
            # a fill kernel, not a pageable host-to-device copy (which synchronises) on every step
            t = torch.full((B,), float(timestep), device=dev, dtype=torch.float32)
        else:
            t = torch.as_tensor(timestep, device=dev).to(torch.float32).reshape(-1)
            t = t.expand(B).contiguous() if t.numel() == 1 else t.contiguous()
        add_text = add_ids = None
        pt = pi = None
        if self.cfg.addition_embed_type == "text_time":
            if not added_cond_kwargs or "text_embeds" not in added_cond_kwargs or "time_ids" not in added_cond_kwargs:
                raise ValueError("added_cond_kwargs needs text_embeds and time_ids for text_time conditioning")
            add_text = _as_f16(added_cond_kwargs["text_embeds"], dev)
            add_ids = added_cond_kwargs["time_ids"].to(device=dev, dtype=torch.float32).contiguous()
            pt, pi = C.c_void_p(add_text.data_ptr()), C.c_void_p(add_ids.data_ptr())
        out = torch.empty((B, self.cfg.out_channels, H, W), device=dev, dtype=torch.float16)
        with torch.cuda.device(dev):
            rc = self._lib.sd_unet_forward(self._h, C.c_void_p(sample.data_ptr()), C.c_void_p(t.data_ptr()),
                                           C.c_void_p(ehs.data_ptr()), ehs.shape[1], pt, pi,
                                           C.c_void_p(out.data_ptr()), B, H, W, C.c_void_p(_stream_ptr()))
        _lib.check(rc, "sd_unet_forward")
        if return_dict:
            return SimpleNamespace(sample=out)
        return (out,)

    forward = __call__

    # LoRA adapters arrive fused into the weights (stablediffusion_amd.weights.fuse_lora); the
    # reference's runtime adapter switches (stable_diffusion.py:252-335) are not part of the engine.
    def set_adapters(self, *a, **k):
        raise NotImplementedError("fuse LoRA weights on load (stablediffusion_amd.weights.fuse_lora)")

    delete_adapters = add_adapter = set_adapters

    def parameters(self):
        return iter(())

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.sd_unet_destroy(self._h)
                self._h = None
        except Exception:
            pass


class _LatentDist:
    """DiagonalGaussianDistribution surface used at sd_unified_pipeline.py:98-106 (host-side sampling)."""

    def __init__(self, moments: torch.Tensor):
        self.mean, logvar = moments.float().chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return (self.mean + self.std * noise).to(torch.float16)

    def mode(self):
        return self.mean.to(torch.float16)


class HipAutoencoderKL:
    """gfx950 engine behind the `AutoencoderKL` call surface the reference uses."""

    def __init__(self, config: VAEConfig, device: str = "cuda"):
        self._lib = _lib.load()
        self.cfg = config
        self.device = torch.device(device)
        self.dtype = torch.float16
        c = _lib.SdVAEConfig()
        c.in_channels = config.in_channels
        c.out_channels = config.out_channels
        c.latent_channels = config.latent_channels
        c.num_blocks = len(config.block_out_channels)
        for i, ch in enumerate(config.block_out_channels):
            c.block_out_channels[i] = ch
        c.layers_per_block = config.layers_per_block
        c.norm_num_groups = config.norm_num_groups
        self._h = C.c_void_p()
        _lib.check(self._lib.sd_vae_create(C.byref(c), C.byref(self._h)), "sd_vae_create")
        self.config = _Config(**config.to_dict())
        self._finalized = False
        self._enc_shift = 0           # sd_vae_encode_range_shift in effect (force_upcast VAEs: raised on overflow)

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        _lib.require_gpu()
        with torch.cuda.device(self.device):
            _load_weights(self._lib, self._h, "vae", state_dict, strict)
            _lib.check(self._lib.sd_vae_finalize(self._h), "sd_vae_finalize")
        self._finalized = True
        return self

    def memory(self):
        w, s = C.c_int64(), C.c_int64()
        _lib.check(self._lib.sd_vae_memory(self._h, C.byref(w), C.byref(s)), "sd_vae_memory")
        return w.value, s.value

    def to(self, device=None, dtype=None):
        # the reference flips the VAE to fp32 around encode when force_upcast is set
        # (sd_unified_pipeline.py:1020-1036); the engine keeps fp32 accumulators and statistics and fp16
        # activations, so dtype requests are accepted and ignored; a force_upcast VAE whose activations leave
        # fp16's range is re-run range-shifted by encode_moments (see there).
        if device is not None and not isinstance(device, torch.dtype) and torch.device(device).type != "cuda":
            raise _lib.EngineError("HipAutoencoderKL lives on the HIP device only")
        return self

    def eval(self):
        return self

    def decode(self, z, return_dict=False, **unused):
        if not self._finalized:
            raise _lib.EngineError("weights not loaded")
        z = _as_f16(z, self.device)
        B, _, h, w = z.shape
        f = 2 ** (len(self.cfg.block_out_channels) - 1)
        img = torch.empty((B, self.cfg.out_channels, h * f, w * f), device=self.device, dtype=torch.float16)
        with torch.cuda.device(self.device):
            rc = self._lib.sd_vae_decode(self._h, C.c_void_p(z.data_ptr()), C.c_void_p(img.data_ptr()), B, h, w,
                                         C.c_void_p(_stream_ptr()))
        _lib.check(rc, "sd_vae_decode")
        if return_dict:
            return SimpleNamespace(sample=img)
        return (img,)

    def encode_moments(self, x):
        if not self._finalized:
            raise _lib.EngineError("weights not loaded")
        x = _as_f16(x, self.device)
        B, _, H, W = x.shape
        f = 2 ** (len(self.cfg.block_out_channels) - 1)
        mom = torch.empty((B, 2 * self.cfg.latent_channels, H // f, W // f), device=self.device,
                          dtype=torch.float16)
        with torch.cuda.device(self.device):
            rc = self._lib.sd_vae_encode(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(mom.data_ptr()), B, H, W,
                                         C.c_void_p(_stream_ptr()))
        _lib.check(rc, "sd_vae_encode")
        if getattr(self.cfg, "force_upcast", False) and not bool(torch.isfinite(mom).all()):
            # The reference runs the VAE in fp32 around encode when force_upcast is set (sd_unified_pipeline.py:1020-1036:
            # the SDXL VAE's activations leave fp16's range on some images).  The engine's equivalent: the same encoder
            # with every inter-layer activation stored 2^-k times smaller (GroupNorm is scale-invariant, eps scaled
            # along: sd_vae_encode_range_shift), k raised until the moments are finite and kept for the handle's next
            # calls.  fp32 accumulators and statistics as always; nothing overflows silently.
            for shift in (4, 8, 12):
                if shift <= self._enc_shift:
                    continue
                _lib.check(self._lib.sd_vae_encode_range_shift(self._h, shift), "sd_vae_encode_range_shift")
                self._enc_shift = shift
                with torch.cuda.device(self.device):
                    rc = self._lib.sd_vae_encode(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(mom.data_ptr()), B, H, W,
                                                 C.c_void_p(_stream_ptr()))
                _lib.check(rc, "sd_vae_encode")
                if bool(torch.isfinite(mom).all()):
                    break
            else:
                raise _lib.EngineError("VAE encode left fp16's range even with its activations stored 2^-12 times smaller "
                                       "(config.force_upcast is set); no latents were produced")
        return mom

    def encode(self, x, return_dict=True):
        return SimpleNamespace(latent_dist=_LatentDist(self.encode_moments(x)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.sd_vae_destroy(self._h)
                self._h = None
        except Exception:
            pass


class _ClipOutput(tuple):
    """Indexable like transformers' ModelOutput (positional order of the non-None fields) with the
    same attribute names: encode_prompt uses `out[0]`, `out[-1][-(k+1)]` and `out.hidden_states[-2]`
    (/root/reference/pipelines/sd_unified_pipeline.py:596-608)."""

    def __new__(cls, names, values):
        o = super().__new__(cls, values)
        o._names = tuple(names)
        return o

    def __getattr__(self, name):
        names = object.__getattribute__(self, "_names")
        if name in names:
            return self[names.index(name)]
        if name in ("last_hidden_state", "pooler_output", "text_embeds", "hidden_states", "attentions"):
            return None
        raise AttributeError(name)

    def keys(self):
        return list(self._names)


class HipCLIPTextModel:
    """gfx950 engine behind the call surface the reference uses on `SDModelWrapper.text_encoder` /
    `.text_encoder_2` (transformers CLIPTextModel / CLIPTextModelWithProjection): `__call__(input_ids,
    output_hidden_states=True)`, `.text_model.final_layer_norm`, `.config`, `.dtype`, `.to`.
    `projection_dim > 0` in the config selects the WithProjection flavour (`out[0]` = text_embeds)."""

    def __init__(self, config, device: str = "cuda"):
        from .config import CLIPTextConfig
        if not isinstance(config, CLIPTextConfig):
            raise TypeError("config must be a stablediffusion_amd.config.CLIPTextConfig (see CLIPTextConfig.from_hf)")
        acts = {"quick_gelu": 0, "gelu": 1}
        if config.hidden_act not in acts:
            raise _lib.EngineError(f"unsupported CLIP activation {config.hidden_act!r}")
        self._lib = _lib.load()
        self.cfg = config
        self.device = torch.device(device)
        self.dtype = torch.float16
        c = _lib.SdClipConfig()
        c.vocab_size = config.vocab_size
        c.hidden_size = config.hidden_size
        c.intermediate_size = config.intermediate_size
        c.num_layers = config.num_hidden_layers
        c.num_heads = config.num_attention_heads
        c.max_positions = config.max_position_embeddings
        c.hidden_act = acts[config.hidden_act]
        c.projection_dim = config.projection_dim
        c.layer_norm_eps = config.layer_norm_eps
        self._h = C.c_void_p()
        _lib.check(self._lib.sd_clip_create(C.byref(c), C.byref(self._h)), "sd_clip_create")
        self.config = _Config(**config.to_dict())
        self._finalized = False
        self.text_model = SimpleNamespace(final_layer_norm=self._final_layer_norm)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.sd_clip_destroy(self._h)
        except Exception:
            pass

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """transformers state-dict keys; the `text_model.` prefix is optional (transformers >= 5 drops it
        for CLIPTextModel), buffers such as `position_ids` are ignored."""
        _lib.require_gpu()
        sd = {}
        for k, v in state_dict.items():
            if k.endswith("position_ids"):
                continue
            if not k.startswith("text_model.") and not k.startswith("text_projection."):
                k = "text_model." + k
            if k.startswith("text_projection.") and self.cfg.projection_dim == 0:
                continue
            sd[k] = v
        with torch.cuda.device(self.device):
            _load_weights(self._lib, self._h, "clip", sd, strict)
            _lib.check(self._lib.sd_clip_finalize(self._h), "sd_clip_finalize")
        self._finalized = True
        return self

    def memory(self):
        w, s = C.c_int64(), C.c_int64()
        _lib.check(self._lib.sd_clip_memory(self._h, C.byref(w), C.byref(s)), "sd_clip_memory")
        return w.value, s.value

    def to(self, device=None, dtype=None):
        if device is not None and not isinstance(device, torch.dtype) and torch.device(device).type != "cuda":
            raise _lib.EngineError("HipCLIPTextModel lives on the HIP device only")
        return self

    def eval(self):
        return self

    def _final_layer_norm(self, x: torch.Tensor) -> torch.Tensor:
        x = _as_f16(x, self.device)
        y = torch.empty_like(x)
        rows = x.numel() // self.cfg.hidden_size
        with torch.cuda.device(self.device):
            rc = self._lib.sd_clip_final_layer_norm(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), rows,
                                                    C.c_void_p(_stream_ptr()))
        _lib.check(rc, "sd_clip_final_layer_norm")
        return y

    def _eos_index(self, ids: torch.Tensor) -> torch.Tensor:
        # transformers modeling_clip: legacy configs (eos_token_id == 2) pool at ids.argmax, newer ones
        # at the first eos_token_id
        if self.cfg.eos_token_id == 2:
            return ids.argmax(dim=-1).to(torch.int32)
        return (ids == self.cfg.eos_token_id).int().argmax(dim=-1).to(torch.int32)

    def __call__(self, input_ids, attention_mask=None, position_ids=None, output_attentions=None,
                 output_hidden_states=None, return_dict=True, **unused):
        if not self._finalized:
            raise _lib.EngineError("weights not loaded")
        if attention_mask is not None or position_ids is not None or output_attentions:
            raise NotImplementedError("HipCLIPTextModel: attention_mask / position_ids / output_attentions are not "
                                      "used by the reference's encode_prompt and not supported")
        ids = input_ids.to(self.device)
        B, T = ids.shape
        H, L, P = self.cfg.hidden_size, self.cfg.num_hidden_layers, self.cfg.projection_dim
        ids32 = ids.to(torch.int32).contiguous()
        eos = self._eos_index(ids).contiguous()
        f16 = dict(device=self.device, dtype=torch.float16)
        hs = torch.empty((L + 1, B, T, H), **f16) if output_hidden_states else None
        last = torch.empty((B, T, H), **f16)
        pooled = torch.empty((B, H), **f16)
        emb = torch.empty((B, P), **f16) if P > 0 else None
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        with torch.cuda.device(self.device):
            rc = self._lib.sd_clip_forward(self._h, ptr(ids32), ptr(eos), ptr(hs), ptr(last), ptr(pooled), ptr(emb), B, T,
                                           C.c_void_p(_stream_ptr()))
        _lib.check(rc, "sd_clip_forward")
        names, vals = [], []
        if P > 0:       # CLIPTextModelOutput: text_embeds, last_hidden_state, hidden_states
            names += ["text_embeds", "last_hidden_state"]; vals += [emb, last]
        else:           # BaseModelOutputWithPooling: last_hidden_state, pooler_output, hidden_states
            names += ["last_hidden_state", "pooler_output"]; vals += [last, pooled]
        if hs is not None:
            names.append("hidden_states"); vals.append(tuple(hs[i] for i in range(L + 1)))
        return _ClipOutput(names, vals)
