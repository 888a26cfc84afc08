#!/usr/bin/env python3
"""Time the HIP CLIP text encoders against transformers' own modules run as PyTorch-ROCm eager fp16 on
the same GPU (random-init weights): python tools/run_clip.py [B]"""
import os
import sys
import time

import torch
import transformers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusion_amd import config  # noqa: E402
from stablediffusion_amd.models import HipCLIPTextModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, cfg, proj in (("CLIP-L (SD1.5 / SDXL text_encoder)", config.clip_l(), False),
                        ("OpenCLIP bigG (SDXL text_encoder_2)", config.openclip_bigg(), True)):
    hf_cfg = transformers.CLIPTextConfig(
        vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
        hidden_act=cfg.hidden_act, projection_dim=cfg.projection_dim or cfg.hidden_size)
    cls = transformers.CLIPTextModelWithProjection if proj else transformers.CLIPTextModel
    torch.manual_seed(0)
    hf = cls(hf_cfg).eval()
    eng = HipCLIPTextModel(cfg).load_state_dict({k: v.half() for k, v in hf.state_dict().items()})
    hf = hf.half().cuda()
    ids = torch.randint(3, 49000, (B, 77)).cuda()
    res = {}
    with torch.no_grad():
        for label, fn in (("engine", lambda: eng(ids, output_hidden_states=True)),
                          ("torch-rocm eager fp16", lambda: hf(ids, output_hidden_states=True))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            res[label] = (time.perf_counter() - t0) / 20 * 1e3
    print(f"{name}, B={B}: engine {res['engine']:.2f} ms, torch-rocm eager fp16 {res['torch-rocm eager fp16']:.2f} ms", flush=True)
