#!/usr/bin/env python3
"""Time GroupNorm(+SiLU) / LayerNorm on the UNet's shapes (GPU box): python tools/run_norm.py [B]
Prints us per op (stats + apply launches together) and effective GB/s for read-once/write-once bytes."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusion_amd import _lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


# (HW, C, count per SD1.5 UNet forward)
GN = [(4096, 320, 12), (4096, 640, 2), (4096, 960, 1), (1024, 320, 1), (1024, 640, 10), (1024, 960, 1), (1024, 1280, 1),
      (1024, 1920, 1), (256, 640, 1), (256, 1280, 10), (256, 1920, 1), (256, 2560, 2), (64, 1280, 12), (64, 2560, 3)]
tot = 0.0
for HW, Cc, cnt in GN:
    x = torch.randn(B, HW, Cc, device="cuda", dtype=torch.float16)
    y = torch.empty_like(x)
    g = torch.ones(Cc, device="cuda", dtype=torch.float32)
    b = torch.zeros(Cc, device="cuda", dtype=torch.float32)
    ms = C.c_float()
    lib.sd_bench_groupnorm(P(x), P(g), P(b), P(y), B, HW, Cc, 32, 1e-5, 1, 50, C.byref(ms), st)
    us = ms.value * 1e3
    tot += us * cnt
    print(f"groupnorm HW{HW:5d} C{Cc:5d} x{cnt:2d}: {us:7.1f} us  {2 * x.numel() * 2 / us / 1e3:7.0f} GB/s (r+w once)", flush=True)
print(f"groupnorm sum per UNet forward: {tot / 1e3:.3f} ms")
tot = 0.0
for rows, Cc, cnt in [(B * 4096, 320, 15), (B * 1024, 640, 15), (B * 256, 1280, 15), (B * 64, 1280, 3)]:
    x = torch.randn(rows, Cc, device="cuda", dtype=torch.float16)
    y = torch.empty_like(x)
    g = torch.ones(Cc, device="cuda", dtype=torch.float32)
    b = torch.zeros(Cc, device="cuda", dtype=torch.float32)
    us = timeit(lambda: lib.sd_op_layernorm(P(x), P(g), P(b), P(y), rows, Cc, 1e-5, st))
    tot += us * cnt
    print(f"layernorm rows{rows:6d} C{Cc:5d} x{cnt:2d}: {us:7.1f} us  {2 * x.numel() * 2 / us / 1e3:7.0f} GB/s", flush=True)
print(f"layernorm sum per UNet forward: {tot / 1e3:.3f} ms")
