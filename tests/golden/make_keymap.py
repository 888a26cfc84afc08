#!/usr/bin/env python3
"""Generates tests/golden/keymap_{sd15,sdxl}.json: the LDM (A1111 single-file) -> diffusers key-name table,
with shapes, and the diffusers UNet / VAE configs, AS THE REFERENCE'S OWN CONVERTER PRODUCES THEM.

Run in the build container only (needs /root/reference; the GPU box never runs this):
    python tests/golden/make_keymap.py

How: the converter module `/root/reference/scripts/convert_from_A1111.py` cannot be imported (its top-level
`from diffusers import ...` fails: diffusers is not installed), but its checkpoint-renaming functions are
pure dict / string code.  They are taken out of the file with `ast` (function definitions only, nothing
else of the module is executed) and run on
  * the published LDM configs of SD 1.5 (v1-inference.yaml) and SDXL-base (sd_xl_base.yaml), written out
    below as plain dicts, and
  * a synthetic LDM-keyed state dict of shape-only (meta) tensors, generated from those configs by the
    LDM architecture rules (CompVis `openaimodel.UNetModel`, `model.Encoder / Decoder`) -- written here
    from the LDM side, independently of stablediffusion_amd/checkpoints.py.
What is committed is DATA: key names, shapes and config values.  No reference source text is stored.
tests/test_checkpoints.py and tests/test_manifest.py check the product's key maps, manifests and configs
against every row of these tables.
"""
import ast
import json
import os
import sys

import torch

REF = "/root/reference/scripts/convert_from_A1111.py"
HERE = os.path.dirname(os.path.abspath(__file__))
WANT = ["assign_to_checkpoint", "shave_segments", "create_unet_diffusers_config", "renew_resnet_paths",
        "renew_attention_paths", "convert_ldm_unet_checkpoint", "create_vae_diffusers_config",
        "renew_vae_resnet_paths", "renew_vae_attention_paths", "conv_attn_to_linear", "convert_ldm_vae_checkpoint"]


def reference_functions():
    tree = ast.parse(open(REF).read(), REF)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANT]
    missing = set(WANT) - {n.name for n in body}
    assert not missing, missing
    mod = ast.Module(body=body, type_ignores=[])
    ns = {"torch": torch, "__name__": "reference_converter_functions"}
    from typing import Dict, Optional, Union
    ns.update(Dict=Dict, Optional=Optional, Union=Union)
    exec(compile(mod, REF, "exec"), ns)
    return ns


# ---- published LDM configs (inputs) ----
SD15 = {"model": {"params": {
    "unet_config": {"params": {"image_size": 32, "in_channels": 4, "out_channels": 4, "model_channels": 320,
                               "attention_resolutions": [4, 2, 1], "num_res_blocks": 2, "channel_mult": [1, 2, 4, 4],
                               "num_heads": 8, "use_spatial_transformer": True, "transformer_depth": 1,
                               "context_dim": 768, "use_checkpoint": True, "legacy": False}},
    "first_stage_config": {"params": {"embed_dim": 4, "ddconfig": {
        "double_z": True, "z_channels": 4, "resolution": 256, "in_channels": 3, "out_ch": 3, "ch": 128,
        "ch_mult": [1, 2, 4, 4], "num_res_blocks": 2, "attn_resolutions": [], "dropout": 0.0}}}}}}
SDXL = {"model": {"params": {
    "network_config": {"params": {"adm_in_channels": 2816, "num_classes": "sequential", "use_checkpoint": True,
                                  "in_channels": 4, "out_channels": 4, "model_channels": 320,
                                  "attention_resolutions": [4, 2], "num_res_blocks": 2, "channel_mult": [1, 2, 4],
                                  "num_head_channels": 64, "use_spatial_transformer": True,
                                  "use_linear_in_transformer": True, "transformer_depth": [1, 2, 10],
                                  "context_dim": 2048, "spatial_transformer_attn_type": "softmax-xformers",
                                  "legacy": False}},
    "first_stage_config": {"params": {"embed_dim": 4, "ddconfig": {
        "attn_type": "vanilla-xformers", "double_z": True, "z_channels": 4, "resolution": 256, "in_channels": 3,
        "out_ch": 3, "ch": 128, "ch_mult": [1, 2, 4, 4], "num_res_blocks": 2, "attn_resolutions": [],
        "dropout": 0.0}}}}}}


def meta(*shape):
    return torch.empty(*shape, device="meta")


# ---- LDM-side architecture rules -> key names + shapes ----
def ldm_unet_keys(up):
    """openaimodel.UNetModel.__init__ walked for its parameter names."""
    sd = {}
    mc, ted = up["model_channels"], up["model_channels"] * 4
    ctx = up["context_dim"]
    linear = up.get("use_linear_in_transformer", False)
    depth = up["transformer_depth"]
    depths = [depth] * len(up["channel_mult"]) if isinstance(depth, int) else list(depth)

    def lin(p, o, i):
        sd[p + ".weight"], sd[p + ".bias"] = meta(o, i), meta(o)

    def conv(p, o, i, k):
        sd[p + ".weight"], sd[p + ".bias"] = meta(o, i, k, k), meta(o)

    def norm(p, c):
        sd[p + ".weight"], sd[p + ".bias"] = meta(c), meta(c)

    def res(p, cin, cout):
        norm(p + ".in_layers.0", cin); conv(p + ".in_layers.2", cout, cin, 3)
        lin(p + ".emb_layers.1", cout, ted)
        norm(p + ".out_layers.0", cout); conv(p + ".out_layers.3", cout, cout, 3)
        if cin != cout:
            conv(p + ".skip_connection", cout, cin, 1)

    def st(p, c, d):
        norm(p + ".norm", c)
        (lin if linear else (lambda q, o, i: conv(q, o, i, 1)))(p + ".proj_in", c, c)
        for k in range(d):
            b = f"{p}.transformer_blocks.{k}"
            for a, kv in (("attn1", c), ("attn2", ctx)):
                sd[f"{b}.{a}.to_q.weight"] = meta(c, c)
                sd[f"{b}.{a}.to_k.weight"] = meta(c, kv)
                sd[f"{b}.{a}.to_v.weight"] = meta(c, kv)
                lin(f"{b}.{a}.to_out.0", c, c)
            lin(f"{b}.ff.net.0.proj", 8 * c, c)
            lin(f"{b}.ff.net.2", c, 4 * c)
            for n in ("norm1", "norm2", "norm3"):
                norm(f"{b}.{n}", c)
        (lin if linear else (lambda q, o, i: conv(q, o, i, 1)))(p + ".proj_out", c, c)

    lin("time_embed.0", ted, mc); lin("time_embed.2", ted, ted)
    if up.get("num_classes") == "sequential":
        lin("label_emb.0.0", ted, up["adm_in_channels"]); lin("label_emb.0.2", ted, ted)
    conv("input_blocks.0.0", mc, up["in_channels"], 3)
    chans, ch, ds, idx = [mc], mc, 1, 1
    for level, mult in enumerate(up["channel_mult"]):
        for _ in range(up["num_res_blocks"]):
            res(f"input_blocks.{idx}.0", ch, mult * mc)
            ch = mult * mc
            if ds in up["attention_resolutions"]:
                st(f"input_blocks.{idx}.1", ch, depths[level])
            chans.append(ch); idx += 1
        if level != len(up["channel_mult"]) - 1:
            conv(f"input_blocks.{idx}.0.op", ch, ch, 3)
            chans.append(ch); idx += 1; ds *= 2
    res("middle_block.0", ch, ch); st("middle_block.1", ch, depths[-1]); res("middle_block.2", ch, ch)
    idx = 0
    for level, mult in list(enumerate(up["channel_mult"]))[::-1]:
        for i in range(up["num_res_blocks"] + 1):
            res(f"output_blocks.{idx}.0", ch + chans.pop(), mult * mc)
            ch = mult * mc
            sub = 1
            if ds in up["attention_resolutions"]:
                st(f"output_blocks.{idx}.1", ch, depths[level]); sub = 2
            if level and i == up["num_res_blocks"]:
                conv(f"output_blocks.{idx}.{sub}.conv", ch, ch, 3); ds //= 2
            idx += 1
    norm("out.0", ch); conv("out.2", up["out_channels"], mc, 3)
    return {"model.diffusion_model." + k: v for k, v in sd.items()}


def ldm_vae_keys(dd, embed_dim):
    """ldm.modules.diffusionmodules.model.Encoder / Decoder + AutoencoderKL's quant convs."""
    sd = {}

    def conv(p, o, i, k):
        sd[p + ".weight"], sd[p + ".bias"] = meta(o, i, k, k), meta(o)

    def norm(p, c):
        sd[p + ".weight"], sd[p + ".bias"] = meta(c), meta(c)

    def res(p, cin, cout):
        norm(p + ".norm1", cin); conv(p + ".conv1", cout, cin, 3); norm(p + ".norm2", cout); conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".nin_shortcut", cout, cin, 1)

    def attn(p, c):
        norm(p + ".norm", c)
        for n in ("q", "k", "v", "proj_out"):
            conv(f"{p}.{n}", c, c, 1)

    ch, mults, nrb, z = dd["ch"], dd["ch_mult"], dd["num_res_blocks"], dd["z_channels"]
    conv("encoder.conv_in", ch, dd["in_channels"], 3)
    cin = ch
    for i, m in enumerate(mults):
        for j in range(nrb):
            res(f"encoder.down.{i}.block.{j}", cin, ch * m); cin = ch * m
        if i != len(mults) - 1:
            conv(f"encoder.down.{i}.downsample.conv", cin, cin, 3)
    res("encoder.mid.block_1", cin, cin); attn("encoder.mid.attn_1", cin); res("encoder.mid.block_2", cin, cin)
    norm("encoder.norm_out", cin); conv("encoder.conv_out", 2 * z if dd["double_z"] else z, cin, 3)
    cin = ch * mults[-1]
    conv("decoder.conv_in", cin, z, 3)
    res("decoder.mid.block_1", cin, cin); attn("decoder.mid.attn_1", cin); res("decoder.mid.block_2", cin, cin)
    for i in reversed(range(len(mults))):
        for j in range(nrb + 1):
            res(f"decoder.up.{i}.block.{j}", cin, ch * mults[i]); cin = ch * mults[i]
        if i != 0:
            conv(f"decoder.up.{i}.upsample.conv", cin, cin, 3)
    norm("decoder.norm_out", cin); conv("decoder.conv_out", dd["out_ch"], cin, 3)
    conv("quant_conv", 2 * embed_dim, 2 * z, 1); conv("post_quant_conv", z, embed_dim, 1)
    return {"first_stage_model." + k: v for k, v in sd.items()}


def table(converted, source, prefix):
    """diffusers key -> (LDM key it came from, shape): identity of the tensor objects tells the source,
    except where the converter re-indexes a tensor (VAE attention 1x1 convs -> linears): matched by name."""
    by_id = {id(v): k for k, v in source.items()}
    rows = {}
    for dk, t in converted.items():
        rows[dk] = {"ldm": by_id.get(id(t)), "shape": list(t.shape)}
    return rows


def main():
    ns = reference_functions()
    for name, cfg, image_size in (("sd15", SD15, 512), ("sdxl", SDXL, 1024)):
        p = cfg["model"]["params"]
        up = (p.get("unet_config") or p["network_config"])["params"]
        ucfg = ns["create_unet_diffusers_config"](cfg, image_size=image_size)
        vcfg = ns["create_vae_diffusers_config"](cfg, image_size=image_size)
        ldm_u = ldm_unet_keys(up)
        ldm_v = ldm_vae_keys(p["first_stage_config"]["params"]["ddconfig"], p["first_stage_config"]["params"]["embed_dim"])
        n_u = sum(v.numel() for v in ldm_u.values())
        src_u, src_v = dict(ldm_u), dict(ldm_v)
        conv_u = ns["convert_ldm_unet_checkpoint"](dict(ldm_u), ucfg)
        conv_v = ns["convert_ldm_vae_checkpoint"](dict(ldm_v), vcfg)
        rows_u = table(conv_u, src_u, "model.diffusion_model.")
        rows_v = table(conv_v, src_v, "first_stage_model.")
        # VAE attention: the converter slices [:, :, 0, 0] (new tensor objects): recover the source by the
        # converter's own documented renames (to_q <- q, ... :530-557)
        ren = {"to_q": "q", "to_k": "k", "to_v": "v", "to_out.0": "proj_out", "group_norm": "norm"}
        for dk, r in rows_v.items():
            if r["ldm"] is None:
                side = dk.split(".")[0]
                leaf = dk.split("attentions.0.")[1]
                for d, l in ren.items():
                    if leaf.startswith(d + "."):
                        r["ldm"] = f"first_stage_model.{side}.mid.attn_1.{l}.{leaf[len(d) + 1:]}"
                assert r["ldm"] in src_v, dk
        assert all(r["ldm"] for r in rows_u.values()) and len(rows_u) == len(src_u), (len(rows_u), len(src_u))
        assert len(rows_v) == len(src_v)
        out = {"_generator": "tests/golden/make_keymap.py (functions of the reference converter, run on shape-only tensors)",
               "unet_config": json.loads(json.dumps(ucfg)), "vae_config": json.loads(json.dumps(vcfg)),
               "unet_param_count": n_u, "vae_param_count": sum(v.numel() for v in ldm_v.values()),
               "unet": rows_u, "vae": rows_v}
        path = os.path.join(HERE, f"keymap_{name}.json")
        with open(path, "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)
        print(name, "unet keys", len(rows_u), "params", n_u, "| vae keys", len(rows_v), "params", out["vae_param_count"],
              "->", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    sys.exit(main())
