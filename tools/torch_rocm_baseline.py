#!/usr/bin/env python3
"""Context number, not the product: the oracle's functional restatement of the UNet / VAE run as
PyTorch-ROCm eager fp16 on the same MI355X (what a diffusers-on-ROCm port of the reference would
execute: MIOpen convs, hipBLASLt linears, SDPA).  Prints ms per UNet forward (CFG batch 8, 64x64
latents) and per VAE decode (4 latents), next to the engine's numbers from bench.py."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import unet_ref, vae_ref  # noqa: E402
from stablediffusion_amd import config, weights  # noqa: E402

dev = "cuda"
ucfg, vcfg = config.sd15_unet(), config.sd15_vae()
uw = {k: v.to(dev) for k, v in weights.synth_state_dict(weights.unet_manifest(ucfg), 2, torch.float16).items()}
vw = {k: v.to(dev) for k, v in weights.synth_state_dict(weights.vae_manifest(vcfg), 3, torch.float16).items()}
x = torch.randn(8, 4, 64, 64, device=dev, dtype=torch.float16)
e = torch.randn(8, 77, 768, device=dev, dtype=torch.float16)
z = torch.randn(4, 4, 64, 64, device=dev, dtype=torch.float16)
orig = unet_ref.timestep_sinusoid
unet_ref.timestep_sinusoid = lambda t, *a, **k: orig(t.cpu(), *a, **k).to(dev)   # sinusoid table on host
with torch.no_grad():
    for name, fn, n in (("unet_forward", lambda: unet_ref.unet_forward(ucfg, uw, x, torch.tensor(501.0), e), 10),
                        ("vae_decode", lambda: vae_ref.vae_decode(vcfg, vw, z), 3)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        print(f"torch-rocm eager fp16 {name}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms")
