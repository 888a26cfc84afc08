"""The product's checkpoint key maps, weight manifests and model configs against the tables the REFERENCE'S OWN
converter functions produce (`/root/reference/scripts/convert_from_A1111.py:31-95, :97-203, :206-485, :490-677`,
run by tests/golden/make_keymap.py in the build container; committed as data: tests/golden/keymap_*.json).
Every key of every table is checked -- 686 / 1680 UNet keys and 248 VAE keys -- not a hand-picked sample, and
not against an inverse map written by the same author.  CPU only."""
import json
import os

import pytest
import torch

from stablediffusion_amd import checkpoints as ck
from stablediffusion_amd import config, weights

HERE = os.path.dirname(os.path.abspath(__file__))
PRESETS = {"sd15": (config.sd15_unet, config.sd15_vae), "sdxl": (config.sdxl_unet, config.sdxl_vae)}


@pytest.fixture(scope="module", params=["sd15", "sdxl"])
def table(request):
    with open(os.path.join(HERE, "golden", f"keymap_{request.param}.json")) as f:
        return request.param, json.load(f)


def test_manifests_equal_the_converters_output_key_for_key(table):
    name, t = table
    ucfg, vcfg = (f() for f in PRESETS[name])
    um, vm = weights.unet_manifest(ucfg), weights.vae_manifest(vcfg)
    assert set(um) == set(t["unet"]), (sorted(set(um) ^ set(t["unet"]))[:8])
    assert set(vm) == set(t["vae"]), (sorted(set(vm) ^ set(t["vae"]))[:8])
    for k, row in t["unet"].items():
        assert tuple(um[k]) == tuple(row["shape"]), k
    for k, row in t["vae"].items():
        assert tuple(vm[k]) == tuple(row["shape"]), k
    assert weights.param_count(um) == t["unet_param_count"] and weights.param_count(vm) == t["vae_param_count"]


def test_ldm_key_maps_equal_the_converters_renames_row_for_row(table):
    """checkpoints.ldm_to_diffusers_{unet,vae} on an LDM-keyed dict of shape stubs: every tensor must land under
    the diffusers name the reference's converter gives the same LDM key, with the converter's shape (the VAE
    attention 1x1 convs arrive as [C, C, 1, 1] and leave as [C, C])."""
    name, t = table
    ucfg, vcfg = (f() for f in PRESETS[name])
    for part, cfg, fn, lshape in (("unet", ucfg, ck.ldm_to_diffusers_unet, None), ("vae", vcfg, ck.ldm_to_diffusers_vae, 4)):
        rows = t[part]
        ldm = {}
        for dk, r in rows.items():
            shape = list(r["shape"])
            if part == "vae" and ".attentions.0.to_" in dk and dk.endswith(".weight"):
                shape = shape + [1, 1]                      # the LDM checkpoint stores these as 1x1 convolutions
            # the tensor's first element carries the row's identity through the rename
            ldm[r["ldm"]] = torch.empty(shape, device="meta")
        assert len(ldm) == len(rows)
        out = fn(dict(ldm), cfg)
        assert set(out) == set(rows)
        for dk, r in rows.items():
            assert out[dk] is ldm[r["ldm"]] or tuple(out[dk].shape) == tuple(r["shape"]), dk
            assert tuple(out[dk].shape) == tuple(r["shape"]), dk
        # and the name-level map itself, where the product exposes one
        if part == "unet":
            km = ck.ldm_unet_key_map(cfg)
            assert km, "empty key map"


def test_configs_equal_the_converters_configs(table):
    """create_unet_diffusers_config / create_vae_diffusers_config (:97-203, :490-511) on the published LDM
    configs vs stablediffusion_amd.config presets, field by field (the converter's `attention_head_dim` is
    diffusers' misnamed head COUNT for SD 1.5 and the per-block head count list for SDXL)."""
    name, t = table
    ucfg, vcfg = (f() for f in PRESETS[name])
    uc, vc = t["unet_config"], t["vae_config"]
    assert tuple(uc["block_out_channels"]) == tuple(ucfg.block_out_channels)
    assert tuple(uc["down_block_types"]) == tuple(ucfg.down_block_types)
    assert tuple(uc["up_block_types"]) == tuple(ucfg.up_block_types)
    assert uc["layers_per_block"] == ucfg.layers_per_block
    assert uc["cross_attention_dim"] == ucfg.cross_attention_dim
    assert uc["in_channels"] == ucfg.in_channels and uc["out_channels"] == ucfg.out_channels
    assert bool(uc["use_linear_projection"]) == bool(ucfg.use_linear_projection)
    assert uc["sample_size"] == ucfg.sample_size
    assert uc["addition_embed_type"] == ucfg.addition_embed_type
    if uc["addition_embed_type"] == "text_time":
        assert uc["addition_time_embed_dim"] == ucfg.addition_time_embed_dim
        assert uc["projection_class_embeddings_input_dim"] == ucfg.projection_class_embeddings_input_dim
    heads = uc["attention_head_dim"]
    heads = [heads] * len(ucfg.block_out_channels) if isinstance(heads, int) else list(heads)
    assert heads == [ucfg.heads_for_block(i) for i in range(len(ucfg.block_out_channels))]
    tl = uc["transformer_layers_per_block"]
    tl = [tl] * len(ucfg.block_out_channels) if isinstance(tl, int) else list(tl)
    assert tl == list(ucfg.transformer_layers_per_block)
    # the product reads the converter's dict directly too
    assert ck.unet_config_from_json(dict(uc)) == ucfg
    assert tuple(vc["block_out_channels"]) == tuple(vcfg.block_out_channels)
    assert vc["latent_channels"] == vcfg.latent_channels and vc["layers_per_block"] == vcfg.layers_per_block
    assert vc["in_channels"] == vcfg.in_channels and vc["out_channels"] == vcfg.out_channels
