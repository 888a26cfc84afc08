"""LoRA adapters, host side: the file-level surface of `/root/reference/models/stable_diffusion.py:229-335`
(`load_lora_weights`, `set_adapters`, `delete_adapters`, `get_list_adapters`) for the fused-on-load engine.

The reference keeps peft adapter layers live inside the diffusers UNet and scales them at run time
(`cross_attention_kwargs["scale"]`, `sd_unified_pipeline.py:190`).  The engine has no adapter layers: the
active adapters are folded into the UNet's Linear weights on the host,
    W = W_base + lora_scale * sum_i adapter_weight_i * (alpha_i / r_i) * up_i @ down_i,
and the packed weights are rebuilt whenever the active set, its weights or the scale change (a host-side
re-pack, about a second for SD1.5; nothing changes per step).  Files are the trainer's
`pytorch_lora_weights.safetensors` (`train_lora_pipeline.py:496-528`: `unet.<module>.lora.down/up.weight`, peft's
`lora_A / lora_B` spelling also accepted, optional `<module>.alpha` scalars).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Union

import torch

from . import weights as _weights

LORA_FILE = "pytorch_lora_weights.safetensors"


def read_lora_file(path_or_dict: Union[str, Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    if isinstance(path_or_dict, dict):
        sd = dict(path_or_dict)
    else:
        path = path_or_dict
        if os.path.isdir(path):
            path = os.path.join(path, LORA_FILE)
        if not path.endswith(".safetensors"):
            raise ValueError("LoRA weights are read from .safetensors files only (nothing is unpickled)")
        from safetensors.torch import load_file
        sd = load_file(path)
    if not sd or not all("lora" in k or k.endswith(".alpha") for k in sd):
        raise ValueError("Invalid LoRA checkpoint.")             # models/stable_diffusion.py:244-246
    return sd


def split_lora(sd: Dict[str, torch.Tensor]):
    """-> (unet part with the `unet.` prefix stripped, {module: alpha}, text-encoder keys)."""
    unet, alphas, text = {}, {}, []
    for k, v in sd.items():
        if k.startswith(("text_encoder.", "text_encoder_2.")):
            text.append(k)
            continue
        kk = k[5:] if k.startswith("unet.") else k
        if kk.endswith(".alpha"):
            alphas[kk[: -len(".alpha")]] = float(v)
        else:
            unet[kk] = v
    return unet, alphas, text


class LoraAdapters:
    """Adapter registry around one base UNet state dict (host tensors, never modified)."""

    def __init__(self, base_unet_sd: Dict[str, torch.Tensor]):
        self.base = base_unet_sd
        self.adapters: Dict[str, Dict] = {}
        self.active: Dict[str, float] = {}
        self.scale = 1.0

    def load(self, path_or_dict, adapter_name: Optional[str] = None) -> str:
        sd = read_lora_file(path_or_dict)
        unet, alphas, text = split_lora(sd)
        if text:
            raise NotImplementedError("text-encoder LoRA layers are not supported by the engine's CLIP (UNet adapters only); "
                                      f"first such key: {text[0]}")
        name = adapter_name or f"default_{len(self.adapters)}"
        if name in self.adapters:
            raise ValueError(f"Adapter name {name} already in use in the Unet - please select a new adapter name.")
        _weights.fuse_lora(self.base, unet, 0.0)                   # validates the module names / pair completeness
        self.adapters[name] = {"unet": unet, "alphas": alphas}
        self.active[name] = 1.0                                    # a freshly loaded adapter is active at weight 1
        return name

    def set(self, adapter_names: Union[List[str], str], adapter_weights: Optional[List[float]] = None):
        names = [adapter_names] if isinstance(adapter_names, str) else list(adapter_names)
        ws = [1.0] * len(names) if adapter_weights is None else (
            [adapter_weights] * len(names) if isinstance(adapter_weights, (int, float)) else list(adapter_weights))
        if len(ws) != len(names):
            raise ValueError(f"Length of adapter names {len(names)} is not equal to the length of their weights {len(ws)}.")
        for n in names:
            if n not in self.adapters:
                raise ValueError(f"Adapter {n} is not loaded")
        self.active = {n: float(w if w is not None else 1.0) for n, w in zip(names, ws)}

    def delete(self, adapter_names: Union[List[str], str]):
        for n in ([adapter_names] if isinstance(adapter_names, str) else list(adapter_names)):
            self.adapters.pop(n, None)
            self.active.pop(n, None)

    def names(self) -> List[str]:
        return list(self.adapters)

    def fused(self) -> Dict[str, torch.Tensor]:
        sd = self.base
        for n, w in self.active.items():
            ad = self.adapters[n]
            by_ratio: Dict[float, Dict[str, torch.Tensor]] = {}
            for k, v in ad["unet"].items():
                mod = k.rsplit(".lora", 1)[0]
                rank = None
                if k.endswith((".lora.down.weight", ".lora_A.weight")):
                    rank = v.shape[0]
                ratio = 1.0
                if mod in ad["alphas"]:
                    r = rank if rank is not None else next(
                        t.shape[0] for kk, t in ad["unet"].items()
                        if kk.startswith(mod + ".") and kk.endswith((".lora.down.weight", ".lora_A.weight")))
                    ratio = ad["alphas"][mod] / r
                by_ratio.setdefault(ratio, {})[k] = v
            for ratio, part in by_ratio.items():
                sd = _weights.fuse_lora(sd, part, adapter_weight=w * self.scale, alpha_over_r=ratio)
        return sd
