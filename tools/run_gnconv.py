#!/usr/bin/env python3
"""GroupNorm + SiLU + 3x3 conv, fused inside the convolution vs GroupNorm kernel + convolution (GPU box):
python tools/run_gnconv.py [N H W Cin Cout ...]   (default: the C2 UNet's shapes)"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusion_amd import _lib  # noqa: E402

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
shapes = [(8, 64, 64, 320, 320), (8, 64, 64, 640, 320), (8, 32, 32, 640, 640), (8, 32, 32, 1280, 640), (8, 16, 16, 1280, 1280),
          (8, 16, 16, 2560, 1280)]
if len(sys.argv) > 5:
    shapes = [tuple(int(v) for v in sys.argv[1:6])]
iters = 20
for N, H, W, Cin, Cout in shapes:
    x = torch.randn(N, H, W, Cin, device="cuda", dtype=torch.float16)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5).half()
    g = torch.ones(Cin, device="cuda"); b = torch.zeros(Cin, device="cuda")
    y = torch.empty(N, H, W, Cout, device="cuda", dtype=torch.float16)
    hn = torch.empty_like(x)
    ms = C.c_float(); fused = C.c_int()
    # fused (the unfused figures come from the separate entries below)
    _lib.check(lib.sd_op_groupnorm_conv2d(P(x), P(g), P(b), 32, 1e-5, 1, P(w), None, None, None, P(y), N, H, W, Cin, Cout, 3, iters,
                                          C.byref(ms), C.byref(fused), st), "gnconv")
    t_f = ms.value * 1e3
    _lib.check(lib.sd_bench_conv2d(P(x), P(w), P(y), N, H, W, Cin, Cout, 3, 1, 0, 0, iters, C.byref(ms), st), "conv")
    t_c = ms.value * 1e3
    _lib.check(lib.sd_bench_groupnorm(P(x), P(g), P(b), P(hn), N, H * W, Cin, 32, 1e-5, 1, iters, C.byref(ms), st), "gn")
    t_g = ms.value * 1e3
    fl = 2.0 * N * H * W * Cout * 9 * Cin
    print(f"{N}x{H}x{W} {Cin}->{Cout}: fused({fused.value}) {t_f:7.1f} us ({fl / t_f / 1e6:6.0f} TF/s) | conv {t_c:7.1f} us ({fl / t_c / 1e6:6.0f} TF/s) "
          f"+ gn(stats+apply) {t_g:6.1f} us", flush=True)
