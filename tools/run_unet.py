#!/usr/bin/env python3
"""Time the UNet forward alone (GPU box): python tools/run_unet.py [--preset sd15] [--batch 8] [--latent 64] [--iters 20]
Prints ms per forward (events around the whole loop, no per-launch brackets); run it under
`rocprofv3 --kernel-trace --stats -- python3 tools/run_unet.py` for per-kernel durations."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusion_amd import config, weights  # noqa: E402
from stablediffusion_amd.models import HipUNet2DConditionModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="sd15")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--latent", type=int, default=64)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
ucfg, _ = (f() for f in config.PRESETS[args.preset])
dev = "cuda"
sd = weights.synth_state_dict(weights.unet_manifest(ucfg), seed=2, dtype=torch.float16)
net = HipUNet2DConditionModel(ucfg, dev).load_state_dict(sd)
x = torch.randn(args.batch, 4, args.latent, args.latent, device=dev, dtype=torch.float16)
e = torch.randn(args.batch, 77, ucfg.cross_attention_dim, device=dev, dtype=torch.float16)
added = None
if args.preset == "sdxl":
    pdim = ucfg.projection_class_embeddings_input_dim - 6 * ucfg.addition_time_embed_dim
    added = {"text_embeds": torch.randn(args.batch, pdim, device=dev, dtype=torch.float16),
             "time_ids": torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * args.batch)}
t = torch.tensor(501.0)
for _ in range(3):
    net(x, t, e, added_cond_kwargs=added)
torch.cuda.synchronize()
res = []
for _ in range(args.rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        net(x, t, e, added_cond_kwargs=added)
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / args.iters)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("SD_"))
print(f"unet forward {args.preset} B{args.batch} {args.latent}x{args.latent}: " + " / ".join(f"{r:.3f}" for r in res) + f" ms  [{tag}]", flush=True)
