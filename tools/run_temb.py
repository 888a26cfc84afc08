#!/usr/bin/env python3
"""Times the time-embedding linears (GPU box): python tools/run_temb.py   [SD_NO_SKINNY_LINEAR=1 for the GEMV kernel]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusion_amd import _lib  # noqa: E402

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
tot = 0.0
for name, B, K, N in [("time_embedding.linear_1", 8, 320, 1280), ("time_embedding.linear_2", 8, 1280, 1280),
                      ("stacked time_emb_proj", 8, 1280, 20160)]:
    x = torch.randn(B, K, device="cuda")
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    b = torch.randn(N, device="cuda")
    y = torch.zeros(B, N, device="cuda")
    for _ in range(5):
        lib.sd_op_small_linear(P(x), P(w), P(b), P(y), B, K, N, 0, 1, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        lib.sd_op_small_linear(P(x), P(w), P(b), P(y), B, K, N, 0, 1, s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    tot += us
    print(f"{name:26s} [{B} x {K}] x [{K} x {N}]: {us:6.1f} us  ({N * K * 2 / us / 1e3:6.0f} GB/s of weights)")
print(f"sum {tot:.1f} us")
