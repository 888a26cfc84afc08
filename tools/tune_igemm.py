#!/usr/bin/env python3
"""Times every LDS-DMA tile variant x split-K factor of the conv/linear kernel on the distinct
shapes of a model (GPU box only) and writes gpurun_out/tune_<preset>.json.

  python tools/tune_igemm.py --preset sd15 --batch 8 --latent 64 [--vae]

Run it with SD_BENCH_COLD_MB=1000: every timed launch then streams its weights from HBM (rotating
copies), as inside a forward; with the weights cache-resident the small-M layers look 15-30 % faster
than they run in the model and the ranking of the variants changes.
"""
import argparse
import ctypes as C
import json
import os
import sys
from collections import OrderedDict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stablediffusion_amd import _lib, config, shapes  # noqa: E402

NAMES = ["256x128s3", "128x128s2", "128x160s2", "128x64s2", "64x64s2", "256x160s3", "256x128stag", "256x160w8",
         "128x64s3", "128x160s3", "halo256x160", "128x80s2", "128x80s3", "ws128x160", "ws128x128g", "halo256x128"]
SKIP = (13, 14)      # wsgemm is not a table choice: launch_igemm2 takes it whenever wsgemm_supported()
HALO = (10, 15)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="sd15")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--vae", action="store_true")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--min-gflop", type=float, default=5.0)
    ap.add_argument("--only", type=int, nargs="*", default=None,
                    help="time only these variants and merge them into profiles/tune/<config>.json (a new variant "
                         "against an existing tuning run)")
    args = ap.parse_args()
    lib = _lib.load()
    ucfg, vcfg = (f() for f in config.PRESETS[args.preset])
    convs = (shapes.vae_decoder_convs(vcfg, args.batch, args.latent, args.latent) if args.vae
             else shapes.unet_convs(ucfg, args.batch, args.latent, args.latent))
    uniq = OrderedDict()
    for c in convs:
        if c.Cout % 8 or c.Cin % 64:
            continue
        e = uniq.setdefault(c.key(), {"shape": c, "count": 0})
        e["count"] += 1
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    name = f"tune_{args.preset}_{'vae' if args.vae else 'unet'}_b{args.batch}_l{args.latent}.json"
    prior = {}
    if args.only is not None:
        for r in json.load(open(os.path.join(ROOT, "profiles", "tune", name))):
            prior[tuple(r["shape"])] = r
    results = []
    total_best = total_flops = 0.0
    for key, e in uniq.items():
        c = e["shape"]
        if c.flops / 1e9 < args.min_gflop:
            continue
        x = torch.randn(c.N, c.H, c.W, c.Cin, device="cuda", dtype=torch.float16)
        w = (torch.randn(c.Cout, c.Cin, c.ks, c.ks, device="cuda") / c.K ** 0.5).half()
        oh, ow = c.out_hw
        y = torch.empty(c.N, oh, ow, c.Cout // 2 if c.geglu else c.Cout, device="cuda", dtype=torch.float16)
        row = {"shape": list(key), "tag": c.tag, "M": c.M, "N": c.Cout, "K": c.K, "count": e["count"],
               "gflop": c.flops / 1e9, "times_us": {}}
        if args.only is not None:
            if tuple(key) not in prior:
                continue
            row = prior[tuple(key)]
        nk = c.K // 64
        for v in range(len(NAMES)):
            if v in SKIP or (args.only is not None and v not in args.only):
                continue
            if c.geglu and v not in (0, 1, 6):
                continue
            if v in HALO and not (c.ks == 3 and c.stride == 1 and
                                  any(c.out_hw[1] % wt == 0 and c.out_hw[0] % (256 // wt) == 0 for wt in (64, 32, 16))):
                continue
            for sp in (1, 2, 3, 4, 6, 8):
                if sp > 1 and (c.geglu or nk // sp < 8):
                    continue
                lib.sd_igemm_force(v, sp)
                ms = C.c_float()
                rc = lib.sd_bench_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(y.data_ptr()),
                                         c.N, c.H, c.W, c.Cin, c.Cout, c.ks, c.stride, c.up, c.geglu, args.iters,
                                         C.byref(ms), st)
                if rc != 0:
                    print("error", key, v, sp, lib.sd_last_error(), flush=True)
                    continue
                row["times_us"][f"{v}/{sp}"] = ms.value * 1e3
        lib.sd_igemm_force(-1, 0)
        if args.only is None:
            ms = C.c_float()
            lib.sd_bench_conv2d(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(y.data_ptr()),
                                c.N, c.H, c.W, c.Cin, c.Cout, c.ks, c.stride, c.up, c.geglu, args.iters, C.byref(ms), st)
            row["heuristic_us"] = ms.value * 1e3
        best = min(row["times_us"], key=row["times_us"].get)
        row["best"] = best
        bt = row["times_us"][best]
        total_best += bt * e["count"]
        total_flops += c.flops * e["count"]
        bv, bs = best.split("/")
        print(f"M={c.M:7d} N={c.Cout:5d} K={c.K:6d} ks{c.ks} s{c.stride} u{c.up} g{c.geglu} x{e['count']:2d} "
              f"best {NAMES[int(bv)]:>10s}/k{bs} {bt:8.1f}us {c.flops / bt / 1e6:7.1f} TF | heur {row['heuristic_us']:8.1f}us | "
              + " ".join(f"{NAMES[v][:7]}:{row['times_us'].get(f'{v}/1', float('nan')):.0f}" for v in range(len(NAMES)) if v not in SKIP), flush=True)
        results.append(row)
    print(f"sum(best) = {total_best / 1e3:.3f} ms for {total_flops / 1e12:.3f} TFLOP -> {total_flops / total_best / 1e6:.1f} TF/s")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", name), "w") as f:
        json.dump(results, f)


if __name__ == "__main__":
    main()
