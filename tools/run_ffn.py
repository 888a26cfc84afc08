#!/usr/bin/env python3
"""Fused GEGLU feed-forward vs its two-GEMM form (GPU box): python tools/run_ffn.py [M] [C] [iters]"""
import ctypes as C
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusion_amd import _lib  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
Cc = int(sys.argv[2]) if len(sys.argv) > 2 else 320
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lib = _lib.load()
g = torch.Generator().manual_seed(1)
x = torch.randn(M, Cc, generator=g).half().cuda()
w1 = (torch.randn(8 * Cc, Cc, generator=g) / Cc ** 0.5).half().cuda()
b1 = (torch.randn(8 * Cc, generator=g) * 0.2).cuda()
w2 = (torch.randn(Cc, 4 * Cc, generator=g) / (4 * Cc) ** 0.5).half().cuda()
b2 = (torch.randn(Cc, generator=g) * 0.2).cuda()
gamma = (1 + 0.2 * torch.randn(Cc, generator=g)).cuda()
beta = (0.2 * torch.randn(Cc, generator=g)).cuda()
y = torch.zeros(M, Cc, dtype=torch.float16, device="cuda")
P = lambda t: C.c_void_p(t.data_ptr())
ms = (C.c_float * 2)()
fused = C.c_int(-1)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.sd_op_ffn_geglu(P(x), P(gamma), P(beta), 1e-5, P(w1), P(b1), P(w2), P(b2), P(y), M, Cc, iters, ms, C.byref(fused), st), "ffn")
with torch.no_grad():
    xf = x.float()
    proj = F.linear(F.layer_norm(xf, (Cc,), gamma, beta, 1e-5), w1.float(), b1)
    hid, gate = proj.chunk(2, dim=-1)
    ref = xf + F.linear(hid * F.gelu(gate), w2.float(), b2)
err = (torch.linalg.vector_norm(y.float() - ref) / torch.linalg.vector_norm(ref)).item()
fl = 2.0 * M * (8 * Cc * Cc + 4 * Cc * Cc)
print(f"M={M} C={Cc}: fused={fused.value} path {ms[0] * 1e3:.1f} us ({fl / ms[0] / 1e9:.0f} TF/s) | two GEMMs {ms[1] * 1e3:.1f} us | rel-L2 {err:.2e}", flush=True)
