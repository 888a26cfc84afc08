#!/usr/bin/env python3
"""Generates tests/golden/hostlogic.npz + hostlogic.json: outputs of the REFERENCE'S OWN host-side functions on the
denoise path, on grids of inputs, so that the product's restatements in stablediffusion_amd/pipeline.py and the
device kernel sd_images_to_uint8 are pinned to the reference's code instead of to a hand re-typed copy.

Run in the build container only (needs /root/reference; the GPU box never runs this):
    python tests/golden/make_hostlogic.py

How (same as make_keymap.py): the modules cannot be imported (`runpod`, `diffusers` are not installed -- ordinary
ImportErrors), but these functions are pure torch / python.  Their definitions are taken out of the files with `ast`
(nothing else of the modules is executed) and called with a stub `self` that carries exactly the attributes they read:
    convert_pt_to_numpy     /root/reference/runpod-worker/handler_logic.py:21-29
    retrieve_timesteps      /root/reference/pipelines/sd_unified_pipeline.py:61-95
    get_timesteps           :722-761        (method)
    _get_add_time_ids       :979-1014       (method)
    prepare_mask_latents    :916-976        (method; masked_image = None or 4-channel latents: no VAE involved)
What is committed is DATA -- inputs (or the seeds / parameters that rebuild them) and the functions' outputs.  No
reference source text is stored.  tests/test_hostlogic_pinned.py compares the product against every row.
"""
import ast
import inspect
import json
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Union

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PIPE = "/root/reference/pipelines/sd_unified_pipeline.py"
HANDLER = "/root/reference/runpod-worker/handler_logic.py"


def extract(path, names, methods=()):
    tree = ast.parse(open(path).read(), path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    for cls in (n for n in tree.body if isinstance(n, ast.ClassDef)):
        body += [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in methods]
    got = {n.name for n in body}
    assert got == set(names) | set(methods), (got, names, methods)
    ns = {"torch": torch, "np": np, "inspect": inspect, "List": List, "Optional": Optional, "Union": Union, "Dict": Dict,
          "__name__": "reference_host_functions"}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


class StubScheduler:
    """Only what retrieve_timesteps / get_timesteps read: set_timesteps(num_inference_steps, device=), .timesteps,
    .order, .config.num_train_timesteps.  The schedule is DDIM's leading spacing with steps_offset 1 (the constants the
    reference restates at scripts/convert_from_A1111.py:947-959), written here independently of the product."""

    def __init__(self, order=1, num_train_timesteps=1000):
        self.order = order
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps)
        self.timesteps = None

    def set_timesteps(self, num_inference_steps, device=None):
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + 1
        if self.order == 2:                       # a 2nd-order scheduler repeats every timestep but the first
            ts = np.concatenate([ts[:1], np.repeat(ts[1:], 2)])
        self.timesteps = torch.from_numpy(ts)


def main():
    out_np, out_js = {}, {}

    # ---- convert_pt_to_numpy: fp16 edge values + a random image, [B,3,H,W] in and uint8 HWC out ----
    conv = extract(HANDLER, ["convert_pt_to_numpy"])["convert_pt_to_numpy"]
    edge = torch.tensor([-1.0, 1.0, 0.0, 0.99951171875, -0.99951171875, 1.5, -1.5, 0.5, -0.5, 0.00390625, -0.00390625, 0.999,
                         0.9961, 0.9922, 0.2, 1e-4, -1e-4, 65504.0, -65504.0, 0.49804688, 0.50195312, 0.7529297], dtype=torch.float16)
    g = torch.Generator().manual_seed(0)
    # every fp16 value in [-1.25, 1.25] would be 30k values; a dense sweep of 4096 of them plus noise is enough to hit
    # every output level's two neighbours
    sweep = torch.linspace(-1.25, 1.25, 4096).half()
    noise = (torch.randn(2 * 3 * 16 * 16 - sweep.numel() % 1, generator=g) * 0.6).half()
    flat = torch.cat([edge, sweep, noise])
    n = (flat.numel() // (3 * 8)) * (3 * 8)
    img = flat[:n].reshape(1, 3, 8, -1)
    img = torch.cat([img, img.flip(-1)], dim=0)                     # batch of 2
    res = conv(img)
    out_np["convert_in_f16"] = img.numpy()
    out_np["convert_out_u8"] = np.stack(res)
    img32 = img.float()                                             # the CPU float32 path (config C1) truncates differently
    out_np["convert_out_u8_from_f32"] = np.stack(conv(img32))

    # ---- retrieve_timesteps + get_timesteps over N x strength x denoising_start x scheduler order ----
    fns = extract(PIPE, ["retrieve_timesteps"], ["get_timesteps", "_get_add_time_ids", "prepare_mask_latents"])
    rows = []
    for order in (1, 2):
        for N in (1, 2, 10, 25, 30, 50):
            sch = StubScheduler(order)
            ts, n_out = fns["retrieve_timesteps"](sch, N, "cpu")
            rows.append({"fn": "retrieve_timesteps", "order": order, "N": N, "timesteps": ts.tolist(), "num": int(n_out)})
            stub = SimpleNamespace(model=SimpleNamespace(scheduler=sch))
            for strength in (0.0, 0.1, 0.3, 0.5, 0.75, 0.999, 1.0):
                t2, n2 = fns["get_timesteps"](stub, N, strength, None)
                rows.append({"fn": "get_timesteps", "order": order, "N": N, "strength": strength, "denoising_start": None,
                             "timesteps": t2.tolist(), "num": int(n2)})
            for ds in (0.0, 0.2, 0.5, 0.8, 0.95):
                t2, n2 = fns["get_timesteps"](stub, N, 0.3, ds)
                rows.append({"fn": "get_timesteps", "order": order, "N": N, "strength": 0.3, "denoising_start": ds,
                             "timesteps": t2.tolist(), "num": int(n2)})
    out_js["timesteps"] = rows

    # ---- _get_add_time_ids: SDXL-base dims, a refiner-like mismatch, a wrong config ----
    rows = []
    for (osz, crop, tsz, ad, exp, proj) in [((1024, 1024), (0, 0), (1024, 1024), 256, 2816, 1280),
                                            ((768, 512), (16, 32), (512, 768), 256, 2816, 1280),
                                            ((1024, 1024), (0, 0), (1024, 1024), 256, 2560, 1280),     # off by one embed: the aesthetic-score message
                                            ((1024, 1024), (0, 0), (1024, 1024), 256, 2000, 1280)]:    # plain mismatch
        row = {"fn": "_get_add_time_ids", "original_size": osz, "crop": crop, "target_size": tsz, "addition_time_embed_dim": ad,
               "expected": exp, "projection_dim": proj}
        try:
            a, bneg = fns["_get_add_time_ids"](None, osz, crop, tsz, osz, crop, tsz, ad, exp, torch.float16, text_encoder_projection_dim=proj)
            row.update(ids=a.tolist(), neg_ids=bneg.tolist(), dtype=str(a.dtype))
        except ValueError as e:
            row.update(error=type(e).__name__)
        rows.append(row)
    out_js["add_time_ids"] = rows

    # ---- prepare_mask_latents: mask resize (nearest) + repeats + CFG doubling; 4-channel "masked image" passes through ----
    rows = []
    g = torch.Generator().manual_seed(1)
    k = 0
    for (mb, mh, mw, bs, h, w, cfg, with_img) in [(1, 64, 64, 2, 8, 8, True, False), (2, 50, 70, 4, 16, 12, True, True),
                                                  (1, 32, 32, 1, 8, 8, False, True), (3, 24, 24, 2, 4, 4, True, False)]:
        mask = (torch.rand(mb, 1, mh, mw, generator=g) > 0.5).float()
        mimg = torch.randn(mb, 4, h, w, generator=g) if with_img else None
        stub = SimpleNamespace(device="cpu", do_classifier_free_guidance=cfg)
        row = {"fn": "prepare_mask_latents", "batch_size": bs, "height": h, "width": w, "cfg": cfg, "key": f"mask{k}"}
        out_np[f"mask{k}_in"] = mask.numpy()
        if mimg is not None:
            out_np[f"mask{k}_img_in"] = mimg.numpy()
        try:
            m_out, l_out = fns["prepare_mask_latents"](stub, mask, mimg, bs, h, w, torch.float32, None)
            out_np[f"mask{k}_out"] = m_out.numpy()
            if l_out is not None:
                out_np[f"mask{k}_img_out"] = l_out.numpy()
            row["latents"] = l_out is not None
        except ValueError as e:
            row["error"] = type(e).__name__
        rows.append(row)
        k += 1
    out_js["mask_latents"] = rows

    np.savez_compressed(os.path.join(HERE, "hostlogic.npz"), **out_np)
    with open(os.path.join(HERE, "hostlogic.json"), "w") as f:
        json.dump(out_js, f, indent=0, separators=(",", ":"))
    print("wrote hostlogic.npz", {k: v.shape for k, v in out_np.items()})
    print("wrote hostlogic.json", {k: len(v) for k, v in out_js.items()})


if __name__ == "__main__":
    main()
