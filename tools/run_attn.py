#!/usr/bin/env python3
"""Time the attention kernel on one shape (GPU box): python tools/run_attn.py B T Tk heads d [iters] [prescaled]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusion_amd import _lib  # noqa: E402

B, T, Tk, H, d = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
presc = int(sys.argv[7]) if len(sys.argv) > 7 else 0
lib = _lib.load()
q = torch.randn(B, T, H * d, device="cuda", dtype=torch.float16)
k = torch.randn(B, Tk, H * d, device="cuda", dtype=torch.float16)
v = torch.randn(B, Tk, H * d, device="cuda", dtype=torch.float16)
o = torch.empty_like(q)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
for _ in range(3):
    lib.sd_op_attention_ex(P(q), P(k), P(v), P(o), B, T, Tk, H, d, H * d, H * d, H * d, H * d, 0, presc, st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    lib.sd_op_attention_ex(P(q), P(k), P(v), P(o), B, T, Tk, H, d, H * d, H * d, H * d, H * d, 0, presc, st)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"B{B} T{T} Tk{Tk} H{H} d{d} prescaled={presc}: {ms * 1e3:.1f} us  {4.0 * B * H * T * Tk * d / ms / 1e9:.1f} TF/s")
