"""LoRA adapters, host side: the file-level surface of `/root/reference/models/stable_diffusion.py:229-335`
(`load_loras`, `load_lora_weights`, `set_adapters`, `delete_adapters`, `get_list_adapters`) for the fused-on-load engine.

The reference keeps peft adapter layers live inside the diffusers UNet / CLIP modules and scales them at run time
(`cross_attention_kwargs["scale"]`, `sd_unified_pipeline.py:190`).  The engine has no adapter layers: the
active adapters are folded into the Linear / conv weights on the host,
    W = W_base + lora_scale * sum_i adapter_weight_i * (alpha_i / r_i) * up_i @ down_i,
and the packed weights are rebuilt when the active set, its weights or the scale change (a host-side re-pack, about a
second for SD1.5; nothing changes per step).

Files (`.safetensors` only, nothing is unpickled), three key spellings, all reduced to (part, module, down | up):
  * the trainer's `pytorch_lora_weights.safetensors` (`train_lora_pipeline.py:496-528`):
        unet.<module>.lora.down.weight / .lora.up.weight, text_encoder[_2].<module>.lora_linear_layer.down / up.weight
  * peft's spelling `<module>.lora_A.weight / .lora_B.weight`
  * kohya / A1111 files -- what the reference's `load_loras` fetches as `{type}_{name}.safetensors` and converts through
    diffusers' `lora_state_dict` (`stable_diffusion.py:259-263`): `lora_unet_<module with _ for .>.lora_down.weight`,
    `.lora_up.weight`, `.alpha`, and `lora_te_` / `lora_te1_` / `lora_te2_` for the text encoders.  The underscore names
    are resolved against the modules the base weights really have (the mapping is not invertible otherwise).
Every key of a file must land on a (down, up) pair of a known module or be that module's alpha: a file whose keys are
accepted but never applied is an error, not a no-op.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple, Union

import torch

LORA_FILE = "pytorch_lora_weights.safetensors"
PARTS = ("unet", "text_encoder", "text_encoder_2")

_DOWN_TAGS = (".lora.down.weight", ".lora_A.weight", ".lora_linear_layer.down.weight", ".lora_down.weight")
_UP_TAGS = (".lora.up.weight", ".lora_B.weight", ".lora_linear_layer.up.weight", ".lora_up.weight")
_KOHYA_PREFIX = {"lora_unet_": "unet", "lora_te_": "text_encoder", "lora_te1_": "text_encoder", "lora_te2_": "text_encoder_2"}


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


def _kohya_module_table(base_sd: Dict[str, torch.Tensor]) -> Dict[str, str]:
    """underscore spelling -> module path, for every weight-carrying module of a base state dict."""
    return {k[: -len(".weight")].replace(".", "_"): k[: -len(".weight")] for k in base_sd if k.endswith(".weight")}


class ParsedLora:
    """One adapter file resolved against the base weights: per part, {module: {"down", "up", "alpha"}}."""

    def __init__(self):
        self.parts: Dict[str, Dict[str, Dict[str, torch.Tensor]]] = {p: {} for p in PARTS}

    def modules(self, part: str):
        return self.parts[part]


def parse_lora(sd: Dict[str, torch.Tensor], bases: Dict[str, Optional[Dict[str, torch.Tensor]]]) -> ParsedLora:
    """Resolve every key of `sd` to (part, module, role) against `bases` = {part: base state dict or None}.
    Raises on keys that match nothing, incomplete pairs, alphas without a pair, unknown modules, shape mismatches."""
    out = ParsedLora()
    tables: Dict[str, Dict[str, str]] = {}

    def kohya(part: str, name: str) -> str:
        base = bases.get(part)
        if base is None:
            raise NotImplementedError(f"the file carries {part} LoRA layers but the wrapper holds no base weights for it "
                                      f"(build it with {part}_state_dict=...)")
        if part not in tables:
            tables[part] = _kohya_module_table(base)
        if name not in tables[part]:
            raise KeyError(f"LoRA targets unknown module {name} of {part}")
        return tables[part][name]

    for k, v in sd.items():
        part, rest, is_kohya = "unet", k, False
        for pre, prt in _KOHYA_PREFIX.items():
            if k.startswith(pre):
                part, rest, is_kohya = prt, k[len(pre):], True
                break
        else:
            for prt in ("text_encoder_2", "text_encoder", "unet"):
                if k.startswith(prt + "."):
                    part, rest = prt, k[len(prt) + 1:]
                    break
        role = None
        if rest.endswith(".alpha"):
            mod, role = rest[: -len(".alpha")], "alpha"
        else:
            for tags, r in ((_DOWN_TAGS, "down"), (_UP_TAGS, "up")):
                for t in tags:
                    if rest.endswith(t):
                        mod, role = rest[: -len(t)], r
                        break
                if role:
                    break
        if role is None:
            raise ValueError(f"LoRA key {k} is neither a down / up matrix nor an alpha of a known spelling")
        if is_kohya:
            mod = kohya(part, mod)
        if bases.get(part) is None and part != "unet":
            raise NotImplementedError(f"the file carries {part} LoRA layers but the wrapper holds no base weights for it "
                                      f"(build it with {part}_state_dict=...); first such key: {k}")
        out.parts[part].setdefault(mod, {})[role] = v
    total = 0
    for part in PARTS:
        base = bases.get(part)
        for mod, p in out.parts[part].items():
            if "down" not in p or "up" not in p:
                what = "alpha without its matrices" if "alpha" in p and len(p) == 1 else "LoRA pair incomplete"
                raise KeyError(f"{what} for {part}.{mod}")
            if base is not None:
                wkey = mod + ".weight"
                if wkey not in base:
                    raise KeyError(f"LoRA targets unknown module {mod} of {part}")
                w = base[wkey]
                r = p["down"].shape[0]
                if p["up"].shape[1] != r or p["up"].shape[0] != w.shape[0] or p["down"].numel() // r != w.numel() // w.shape[0]:
                    raise ValueError(f"LoRA shapes {tuple(p['up'].shape)} x {tuple(p['down'].shape)} do not fit "
                                     f"{part}.{wkey} {tuple(w.shape)}")
            total += 1
    if total == 0:
        raise ValueError("Invalid LoRA checkpoint: no (down, up) pair in it.")
    return out


def fuse_part(base: Dict[str, torch.Tensor], mods: Dict[str, Dict[str, torch.Tensor]], weight: float,
              out: Optional[Dict[str, torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """out[W] += weight * (alpha / r) * up @ down for every module of `mods` (out starts as a shallow copy of base)."""
    out = dict(base) if out is None else out
    for mod, p in mods.items():
        wkey = mod + ".weight"
        w = out[wkey]
        r = p["down"].shape[0]
        ratio = float(p["alpha"]) / r if "alpha" in p else 1.0
        delta = (p["up"].float().reshape(p["up"].shape[0], r) @ p["down"].float().reshape(r, -1)) * (weight * ratio)
        out[wkey] = (w.float() + delta.reshape(w.shape)).to(w.dtype)
    return out


class LoraAdapters:
    """Adapter registry around the base state dicts (host tensors, never modified) of the UNet and, when the wrapper
    holds them, the text encoders."""

    def __init__(self, base_unet_sd: Dict[str, torch.Tensor], text_encoder_sd: Optional[Dict[str, torch.Tensor]] = None,
                 text_encoder_2_sd: Optional[Dict[str, torch.Tensor]] = None):
        self.bases = {"unet": base_unet_sd, "text_encoder": text_encoder_sd, "text_encoder_2": text_encoder_2_sd}
        self.base = base_unet_sd
        self.adapters: Dict[str, ParsedLora] = {}
        self.active: Dict[str, float] = {}
        self.scale = 1.0

    def load(self, path_or_dict, adapter_name: Optional[str] = None) -> str:
        sd = read_lora_file(path_or_dict)
        name = adapter_name or f"default_{len(self.adapters)}"
        if name in self.adapters:
            raise ValueError(f"Adapter name {name} already in use in the Unet - please select a new adapter name.")
        parsed = parse_lora(sd, self.bases)                        # full validation BEFORE the adapter is registered
        self.adapters[name] = parsed
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

    def touches(self, part: str) -> bool:
        return any(self.adapters[n].modules(part) for n in self.adapters)

    def fused(self, part: str = "unet") -> Dict[str, torch.Tensor]:
        base = self.bases[part]
        if base is None:
            raise ValueError(f"no base weights for {part}")
        sd = None
        for n, w in self.active.items():
            mods = self.adapters[n].modules(part)
            if mods:
                sd = fuse_part(base, mods, w * self.scale, sd)
        return dict(base) if sd is None else sd

    def signature(self) -> Tuple:
        """What the fused weights depend on: rebuilds are skipped while it is unchanged."""
        return (tuple(sorted(self.active.items())), float(self.scale), tuple(self.adapters))
