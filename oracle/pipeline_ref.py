"""ORACLE -- test infrastructure only.  fp32 CPU restatement of the denoise loop + VAE decode.

Restates `/root/reference/pipelines/sd_unified_pipeline.py:465-523` for txt2img with CFG on:
  :467-469 duplicate latents, :472 scale_model_input, :475-482 UNet, :484-486 CFG combine,
  :489 scheduler.step, :511-523 un-scale + vae.decode.
PARITY UNPINNED (see oracle/unet_ref.py): the reference cannot be imported here
(`diffusers`, `torchvision` absent) and holds no golden outputs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import numpy as np
import torch

from .unet_ref import unet_forward
from .vae_ref import vae_decode
from .schedulers_ref import DDIMRef, DPMpp2MKarrasRef, DPMpp2MRef, EulerRef, PNDMRef, UniPCRef

# the deterministic schedulers of the registry (the stochastic ones take their noise explicitly: tests only)
SCHEDULERS = {"DDIM": DDIMRef, "DPM++ 2M": DPMpp2MRef, "euler": EulerRef, "DPM++ 2M Karras": DPMpp2MKarrasRef,
              "PNDM": PNDMRef, "uni_pc": UniPCRef}


@torch.no_grad()
def denoise_ref(unet_cfg, unet_w, latents, prompt_embeds_2b, steps, guidance_scale=5.0,
                scheduler="DDIM", added_cond_kwargs=None, return_trace=False):
    """latents [B,4,h,w] fp32 (unit-variance noise), prompt_embeds_2b = cat([neg, pos]) [2B,L,D]."""
    sch = SCHEDULERS[scheduler]()
    ts = sch.set_timesteps(steps)
    x = latents.double().numpy() * sch.init_noise_sigma
    trace = []
    for t in ts:
        xin = np.concatenate([x, x], axis=0)
        xin = sch.scale_model_input(xin, t)
        eps = unet_forward(unet_cfg, unet_w, torch.from_numpy(xin).float(), torch.tensor(float(t)),
                           prompt_embeds_2b, added_cond_kwargs).double().numpy()
        e_u, e_t = np.split(eps, 2, axis=0)
        e = guidance_scale * (e_t - e_u) + e_u
        x = sch.step(e, t, x)
        if return_trace:
            trace.append(torch.from_numpy(x).float())
    out = torch.from_numpy(x).float()
    return (out, trace) if return_trace else out


@torch.no_grad()
def img2img_denoise_ref(unet_cfg, unet_w, init_latents, noise, prompt_embeds_2b, steps, strength,
                        guidance_scale=5.0, scheduler="DDIM", added_cond_kwargs=None):
    """img2img branch `/root/reference/pipelines/sd_unified_pipeline.py:236-264`: get_timesteps (`:722-761`,
    t_start = steps - min(int(steps * strength), steps)), add_noise of the initial latents at the first kept
    timestep (`:841`), then the loop `:465-507` over timesteps[t_start:].  `noise` is handed in (the product
    draws it from a seeded device generator; the test reproduces that draw)."""
    sch = SCHEDULERS[scheduler]()
    ts = sch.set_timesteps(steps)
    t_start = max(steps - min(int(steps * strength), steps), 0)
    ts = ts[t_start:]
    if hasattr(sch, "start_at"):
        sch.start_at(ts[0])
    x = sch.add_noise(init_latents.double().numpy(), noise.double().numpy(), ts[0])
    for t in ts:
        xin = sch.scale_model_input(np.concatenate([x, x], axis=0), t)
        eps = unet_forward(unet_cfg, unet_w, torch.from_numpy(xin).float(), torch.tensor(float(t)),
                           prompt_embeds_2b, added_cond_kwargs).double().numpy()
        e_u, e_t = np.split(eps, 2, axis=0)
        x = sch.step(guidance_scale * (e_t - e_u) + e_u, t, x)
    return torch.from_numpy(x).float()


@torch.no_grad()
def txt2img_ref(unet_cfg, unet_w, vae_cfg, vae_w, latents, prompt_embeds_2b, steps,
                guidance_scale=5.0, scheduler="DDIM", added_cond_kwargs=None):
    lat = denoise_ref(unet_cfg, unet_w, latents, prompt_embeds_2b, steps, guidance_scale,
                      scheduler, added_cond_kwargs)
    if vae_cfg.latents_mean is not None and vae_cfg.latents_std is not None:
        mean = torch.tensor(vae_cfg.latents_mean).view(1, -1, 1, 1)
        std = torch.tensor(vae_cfg.latents_std).view(1, -1, 1, 1)
        z = lat * std / vae_cfg.scaling_factor + mean
    else:
        z = lat / vae_cfg.scaling_factor
    return vae_decode(vae_cfg, vae_w, z), lat


def to_uint8_hwc(images: torch.Tensor) -> np.ndarray:
    """`convert_pt_to_numpy`, /root/reference/runpod-worker/handler_logic.py:21-29 (truncating cast)."""
    out = []
    for i in range(images.shape[0]):
        img = (images[i].float() / 2 + 0.5).clamp(0, 1)
        out.append((img.permute(1, 2, 0) * 255).to(torch.uint8).numpy())
    return np.stack(out)
