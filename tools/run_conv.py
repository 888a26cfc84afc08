#!/usr/bin/env python3
"""Run one conv shape repeatedly with a forced tile variant (for rocprofv3 --pmc sessions)."""
import argparse
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusion_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="8,64,64,320,320,3,1,0,0", help="N,H,W,Cin,Cout,ks,stride,up,geglu")
ap.add_argument("--variant", type=int, default=2)
ap.add_argument("--splits", type=int, default=1)
ap.add_argument("--iters", type=int, default=20)
a = ap.parse_args()
N, H, W, Cin, Cout, ks, stride, up, geglu = (int(v) for v in a.shape.split(","))
lib = _lib.load()
x = torch.randn(N, H, W, Cin, device="cuda", dtype=torch.float16)
w = (torch.randn(Cout, Cin, ks, ks, device="cuda") / (Cin * ks * ks) ** 0.5).half()
oh, ow = (H << up) // stride, (W << up) // stride
y = torch.empty(N, oh, ow, Cout // 2 if geglu else Cout, device="cuda", dtype=torch.float16)
lib.sd_igemm_force(a.variant, a.splits)
ms = C.c_float()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rc = lib.sd_bench_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(y.data_ptr()), N, H, W, Cin,
                         Cout, ks, stride, up, geglu, a.iters, C.byref(ms), st)
assert rc == 0, lib.sd_last_error()
flops = 2.0 * N * oh * ow * Cout * ks * ks * Cin
print(f"{ms.value * 1e3:.1f} us  {flops / ms.value / 1e9:.1f} TF/s")
