#!/usr/bin/env python3
"""Per-layer-shape time table of one UNet forward (GPU box): SD_PROF_SHAPES=1 python tools/profile_layers.py
[--preset sd15] [--batch 8] [--latent 64] [--vae].  Every launch is bracketed by HIP events on the launch
stream (sd_prof_*), rows are (tile variant / split-K, M x N x K, fused extras); `excess` = time above what
the row's FLOPs would take at 1.1 PFLOP/s (about the best this chip's GEMMs reach on random data)."""
import argparse
import ctypes as C
import os
import sys

os.environ.setdefault("SD_PROF_SHAPES", "1")
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusion_amd import _lib, config, weights  # noqa: E402
from stablediffusion_amd.models import HipAutoencoderKL, HipUNet2DConditionModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="sd15")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--latent", type=int, default=64)
ap.add_argument("--vae", action="store_true")
args = ap.parse_args()
lib = _lib.load()
ucfg, vcfg = (f() for f in config.PRESETS[args.preset])
dev = "cuda"
if args.vae:
    sd = weights.synth_state_dict(weights.vae_manifest(vcfg), seed=3, dtype=torch.float16)
    net = HipAutoencoderKL(vcfg, dev).load_state_dict(sd)
    z = torch.randn(args.batch, 4, args.latent, args.latent, device=dev, dtype=torch.float16)
    run = lambda: net.decode(z)
else:
    sd = weights.synth_state_dict(weights.unet_manifest(ucfg), seed=2, dtype=torch.float16)
    net = HipUNet2DConditionModel(ucfg, dev).load_state_dict(sd)
    x = torch.randn(args.batch, 4, args.latent, args.latent, device=dev, dtype=torch.float16)
    e = torch.randn(args.batch, 77, ucfg.cross_attention_dim, device=dev, dtype=torch.float16)
    added = None
    if args.preset == "sdxl":
        pdim = ucfg.projection_class_embeddings_input_dim - 6 * ucfg.addition_time_embed_dim
        added = {"text_embeds": torch.randn(args.batch, pdim, device=dev, dtype=torch.float16),
                 "time_ids": torch.tensor([[1024.0, 1024, 0, 0, 1024, 1024]] * args.batch)}
    run = lambda: net(x, torch.tensor(501.0), e, added_cond_kwargs=added)
for _ in range(3):
    run()
torch.cuda.synchronize()
lib.sd_prof_enable(1)
REP = 3
for _ in range(REP):
    run()
ents = (_lib.SdProfEntry * 512)()
n = C.c_int()
_lib.check(lib.sd_prof_collect(ents, 512, C.byref(n)), "sd_prof_collect")
lib.sd_prof_enable(0)
rows = [(e.kernel.decode(), e.flops / REP, e.bytes / REP, e.ms / REP, e.launches // REP) for e in ents[: n.value]]
tot = sum(r[3] for r in rows)
rows.sort(key=lambda r: -(r[3] - r[1] / 1.1e15 * 1e3))
print(f"{'row':46s} {'x':>3s} {'ms':>7s} {'us/launch':>9s} {'TF/s':>7s} {'GB/s':>7s} {'excess ms':>9s}")
for k, f, b, ms, l in rows:
    ex = ms - f / 1.1e15 * 1e3
    print(f"{k:46s} {l:3d} {ms:7.3f} {ms / max(l, 1) * 1e3:9.1f} {f / ms / 1e9 if ms else 0:7.0f} {b / ms / 1e6 if ms else 0:7.0f} {ex:9.3f}")
print(f"total {tot:.3f} ms over {sum(r[4] for r in rows)} launches; FLOPs {sum(r[1] for r in rows) / 1e12:.3f} T")
