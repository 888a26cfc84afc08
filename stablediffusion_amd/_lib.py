"""ctypes binding of libsd_engine.so (the C-ABI in include/sd_engine.h).

There is deliberately no fallback: if the shared library is missing or a call fails the shim
raises -- the product path never routes through the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsd_engine.so")

SD_MAX_BLOCKS = 4
SD_DTYPE_F16 = 0
SD_DTYPE_F32 = 1


class SdUNetConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32),
        ("out_channels", C.c_int32),
        ("num_blocks", C.c_int32),
        ("block_out_channels", C.c_int32 * SD_MAX_BLOCKS),
        ("down_block_has_attn", C.c_int32 * SD_MAX_BLOCKS),
        ("up_block_has_attn", C.c_int32 * SD_MAX_BLOCKS),
        ("num_heads", C.c_int32 * SD_MAX_BLOCKS),
        ("transformer_layers", C.c_int32 * SD_MAX_BLOCKS),
        ("layers_per_block", C.c_int32),
        ("cross_attention_dim", C.c_int32),
        ("use_linear_projection", C.c_int32),
        ("norm_num_groups", C.c_int32),
        ("norm_eps", C.c_float),
        ("flip_sin_to_cos", C.c_int32),
        ("freq_shift", C.c_float),
        ("addition_time_embed_dim", C.c_int32),
        ("projection_class_embeddings_input_dim", C.c_int32),
    ]


class SdVAEConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32),
        ("out_channels", C.c_int32),
        ("latent_channels", C.c_int32),
        ("num_blocks", C.c_int32),
        ("block_out_channels", C.c_int32 * SD_MAX_BLOCKS),
        ("layers_per_block", C.c_int32),
        ("norm_num_groups", C.c_int32),
    ]


class SdClipConfig(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32),
        ("hidden_size", C.c_int32),
        ("intermediate_size", C.c_int32),
        ("num_layers", C.c_int32),
        ("num_heads", C.c_int32),
        ("max_positions", C.c_int32),
        ("hidden_act", C.c_int32),
        ("projection_dim", C.c_int32),
        ("layer_norm_eps", C.c_float),
    ]


class SdProfEntry(C.Structure):
    _fields_ = [("kernel", C.c_char * 64), ("flops", C.c_double), ("bytes", C.c_double),
                ("ms", C.c_double), ("launches", C.c_int64)]


# name -> (restype, argtypes); every symbol include/sd_engine.h declares
_P = C.c_void_p
_I = C.c_int
_I64 = C.c_int64
_F = C.c_float
SIGNATURES = {
    "sd_last_error": (C.c_char_p, []),
    "sd_engine_version": (_I, []),
    "sd_engine_arch": (C.c_char_p, []),
    "sd_unet_create": (_I, [C.POINTER(SdUNetConfig), C.POINTER(_P)]),
    "sd_unet_destroy": (_I, [_P]),
    "sd_unet_num_weights": (_I, [_P]),
    "sd_unet_weight_info": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_I64), C.POINTER(_I)]),
    "sd_unet_set_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(_I64), _I, _I]),
    "sd_unet_finalize": (_I, [_P]),
    "sd_unet_forward": (_I, [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P]),
    "sd_unet_use_graph": (_I, [_P, _I]),
    "sd_unet_text_kv_cache": (_I, [_P, _I]),
    "sd_unet_memory": (_I, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "sd_vae_create": (_I, [C.POINTER(SdVAEConfig), C.POINTER(_P)]),
    "sd_vae_destroy": (_I, [_P]),
    "sd_vae_num_weights": (_I, [_P]),
    "sd_vae_weight_info": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_I64), C.POINTER(_I)]),
    "sd_vae_set_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(_I64), _I, _I]),
    "sd_vae_finalize": (_I, [_P]),
    "sd_vae_decode": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "sd_vae_encode": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "sd_vae_encode_range_shift": (_I, [_P, _I]),
    "sd_vae_memory": (_I, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "sd_clip_create": (_I, [C.POINTER(SdClipConfig), C.POINTER(_P)]),
    "sd_clip_destroy": (_I, [_P]),
    "sd_clip_num_weights": (_I, [_P]),
    "sd_clip_weight_info": (_I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(_I64), C.POINTER(_I)]),
    "sd_clip_set_weight": (_I, [_P, C.c_char_p, _P, C.POINTER(_I64), _I, _I]),
    "sd_clip_finalize": (_I, [_P]),
    "sd_clip_forward": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P]),
    "sd_clip_final_layer_norm": (_I, [_P, _P, _P, _I64, _P]),
    "sd_clip_memory": (_I, [_P, C.POINTER(_I64), C.POINTER(_I64)]),
    "sd_op_attention_ex": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sd_cfg_duplicate": (_I, [_P, _P, _I64, _I, _F, _P]),
    "sd_cfg_ddim_step": (_I, [_P, _P, _I64, _F, _F, _F, _P]),
    "sd_inpaint_blend": (_I, [_P, _P, _P, _P, _F, _F, _I, _I, _I, _I, _P]),
    "sd_images_to_uint8": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "sd_cfg_linear_step": (_I, [_P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _P]),
    "sd_igemm_force": (_I, [_I, _I]),
    "sd_probe_mfma": (_I, [_I, C.POINTER(_F), _P]),
    "sd_probe_lds_dma": (_I, [_I64, _I, _I, _I, C.POINTER(_F), _P]),
    "sd_probe_copy": (_I, [_I64, _I, C.POINTER(_F), _P]),
    "sd_prof_enable": (_I, [_I]),
    "sd_prof_collect": (_I, [C.POINTER(SdProfEntry), _I, C.POINTER(_I)]),
    "sd_op_conv2d": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "sd_op_conv3x3_small_cout": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sd_op_conv2d_groupnorm": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I,
                                    C.POINTER(_I), _P]),
    "sd_op_groupnorm_conv2d": (_I, [_P, _P, _P, _I, _F, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_F),
                                    C.POINTER(_I), _P]),
    "sd_op_unet_conv_in": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_F), _P]),
    "sd_op_unet_conv_out": (_I, [_P, _P, _P, _I, _F, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, C.POINTER(_F), _P]),
    "sd_op_ffn_geglu": (_I, [_P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _I, C.POINTER(_F), C.POINTER(_I), _P]),
    "sd_bench_conv2d": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_F), _P]),
    "sd_op_groupnorm": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "sd_op_groupnorm_concat": (_I, [_P, _I, _I, _P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "sd_bench_groupnorm": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, C.POINTER(_F), _P]),
    "sd_op_timestep_sinusoid": (_I, [_P, _P, _I, _I, _I, _F, _P]),
    "sd_op_small_linear": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "sd_op_layernorm": (_I, [_P, _P, _P, _P, _I, _I, _F, _P]),
    "sd_op_attention": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
}

_lib = None


class EngineError(RuntimeError):
    pass


def load():
    """Load libsd_engine.so; raises if it has not been built (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            f"{LIB_PATH} not found: build it with `python -m stablediffusion_amd.build` "
            "(hipcc --offload-arch=gfx950). The HIP engine is mandatory; there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().sd_last_error()
        raise EngineError(f"{what}: error {rc}: {msg.decode() if msg else '?'}")


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise EngineError("no HIP device visible: the gfx950 engine cannot run (and nothing else will run in its place)")
