"""Generates tests/golden/*.npz from the fp32 CPU oracle (oracle/).

PARITY UNPINNED: the reference (`/root/reference`) cannot be imported in the build container
(diffusers / torchvision absent) and holds no fixtures, so these vectors are outputs
of the build's own restatement on a width-reduced config, committed so that (a) the oracle cannot
drift silently and (b) the GPU box can check the HIP engine against them without any reference code.
Weights are regenerated from their seed by stablediffusion_amd.weights.synth_state_dict and rounded
to fp16 (what the engine stores); a checksum of them is stored next to the vectors.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pipeline_ref, unet_ref, vae_ref  # noqa: E402
from stablediffusion_amd import config, weights  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
UNET_SEED, VAE_SEED, PERTURB = 101, 102, 0.1


def golden_weights():
    ucfg, vcfg = config.tiny_unet(), config.tiny_vae()
    uw = {k: v.half().float() for k, v in
          weights.synth_state_dict(weights.unet_manifest(ucfg), seed=UNET_SEED, perturb=PERTURB).items()}
    vw = {k: v.half().float() for k, v in
          weights.synth_state_dict(weights.vae_manifest(vcfg), seed=VAE_SEED, perturb=PERTURB).items()}
    return ucfg, vcfg, uw, vw


def checksum(sd):
    return float(sum(v.double().abs().sum().item() for v in sd.values()))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    ucfg, vcfg, uw, vw = golden_weights()
    g = torch.Generator().manual_seed(2024)
    x = torch.randn(2, 4, 16, 16, generator=g).half().float()
    ehs = torch.randn(2, 77, ucfg.cross_attention_dim, generator=g).half().float()
    t = torch.tensor([981.0, 41.0])
    with torch.no_grad():
        y = unet_ref.unet_forward(ucfg, uw, x, t, ehs)
        z = torch.randn(1, 4, 8, 8, generator=g).half().float()
        img = vae_ref.vae_decode(vcfg, vw, z)
        pix = torch.randn(1, 3, 32, 32, generator=g).half().float()
        mom = vae_ref.vae_encode_moments(vcfg, vw, pix)
        lat0 = torch.randn(1, 4, 8, 8, generator=g).half().float()
        pe2 = torch.randn(2, 77, ucfg.cross_attention_dim, generator=g).half().float()
        images, lat = pipeline_ref.txt2img_ref(ucfg, uw, vcfg, vw, lat0, pe2, steps=4, guidance_scale=5.0)
    np.savez_compressed(
        os.path.join(HERE, "tiny_sd.npz"),
        unet_x=x.numpy(), unet_ehs=ehs.numpy(), unet_t=t.numpy(), unet_y=y.numpy(),
        vae_z=z.numpy(), vae_img=img.numpy(), vae_pix=pix.numpy(), vae_moments=mom.numpy(),
        pipe_latents0=lat0.numpy(), pipe_embeds2b=pe2.numpy(), pipe_latents=lat.numpy(),
        pipe_images=images.numpy(), pipe_uint8=pipeline_ref.to_uint8_hwc(images),
        unet_weight_checksum=np.float64(checksum(uw)), vae_weight_checksum=np.float64(checksum(vw)),
    )
    print("wrote", os.path.join(HERE, "tiny_sd.npz"))


if __name__ == "__main__":
    main()
