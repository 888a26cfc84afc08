#!/usr/bin/env python3
"""L2 -> LDS streaming rate of the LDS-DMA path with every CU streaming (GPU box): python tools/probe_dma.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusion_amd import _lib  # noqa: E402

lib = _lib.load()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
print("region per block | shared (all blocks the same bytes)            | own region per block")
print("                 | " + "  ".join(f"d{d:<2d}" for d in (1, 2, 4, 8, 16, 32)) + "   TB/s aggregate | " + "  ".join(f"d{d:<2d}" for d in (1, 2, 4, 8, 16, 32)))
for region in (64 << 10, 256 << 10, 1 << 20, 4 << 20, 16 << 20):
    row = []
    for shared in (1, 0):
        for depth in (1, 2, 4, 8, 16, 32):
            g = C.c_float(0)
            passes = max(1, (32 << 20) // region)
            rc = lib.sd_probe_lds_dma(C.c_int64(region), passes, depth, shared, C.byref(g), s)
            row.append(g.value / 1e3 if rc == 0 else float("nan"))
    reg = []
    for shared in (3, 2):
        g = C.c_float(0)
        rc = lib.sd_probe_lds_dma(C.c_int64(region), max(1, (32 << 20) // region), 8, shared, C.byref(g), s)
        reg.append(g.value / 1e3 if rc == 0 else float("nan"))
    print(f"{region >> 10:8d} KiB     | " + " ".join(f"{v:4.1f}" for v in row[:6]) + "                 | " + " ".join(f"{v:4.1f}" for v in row[6:])
          + f"   | into registers (8 in flight): shared {reg[0]:4.1f}  own {reg[1]:4.1f}", flush=True)
