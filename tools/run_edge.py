#!/usr/bin/env python3
"""Times the two one-launch ends of the UNet (edge.hip) through the C-ABI: tools/run_edge.py [N H W]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusion_amd import _lib  # noqa: E402

lib = _lib.load()
N, H, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 64, 64)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ms = C.c_float(0)
x = torch.randn(N, 4, H, W, device="cuda").half()
w = torch.randn(320, 4, 3, 3, device="cuda").half() * 0.1
b = torch.randn(320, device="cuda")
y = torch.zeros(N, H, W, 320, dtype=torch.float16, device="cuda")
summ = torch.zeros(N, H * W // 128, 32, 2, device="cuda")
rc = lib.sd_op_unet_conv_in(P(x), P(w), P(b), P(y), P(summ), 32, N, 4, H, W, 320, 20, C.byref(ms), s)
assert rc == 0, lib.sd_last_error()
print(f"conv_in  {N}x4x{H}x{W} -> 320: {ms.value * 1e3:.1f} us ({N * H * W * 320 * 2 / ms.value / 1e6:.0f} GB/s of output)")
xo = torch.randn(N, H, W, 320, device="cuda").half()
wo = torch.randn(4, 320, 3, 3, device="cuda").half() * 0.02
bo = torch.randn(4, device="cuda")
ga, be = torch.ones(320, device="cuda"), torch.zeros(320, device="cuda")
yo = torch.zeros(N, 4, H, W, dtype=torch.float16, device="cuda")
rc = lib.sd_op_unet_conv_out(P(xo), P(ga), P(be), 32, 1e-5, 1, P(wo), P(bo), P(yo), N, H, W, 320, 4, 20, C.byref(ms), s)
assert rc == 0, lib.sd_last_error()
print(f"norm_out + SiLU + conv_out {N}x{H}x{W}x320 -> 4: {ms.value * 1e3:.1f} us ({N * H * W * 320 * 2 / ms.value / 1e6:.0f} GB/s of input)")
