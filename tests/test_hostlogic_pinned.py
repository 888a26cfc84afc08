"""The product's host-side restatements checked against OUTPUTS OF THE REFERENCE'S OWN CODE (tests/golden/hostlogic.*,
produced by tests/golden/make_hostlogic.py from ast-extracted function definitions of
/root/reference/runpod-worker/handler_logic.py:21-29 and /root/reference/pipelines/sd_unified_pipeline.py:61-95,
:722-761, :916-976, :979-1014).  Bit-exact for the uint8 bytes and the index work; the fixtures are data only."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from stablediffusion_amd import pipeline as P

HERE = os.path.dirname(os.path.abspath(__file__))
NPZ = np.load(os.path.join(HERE, "golden", "hostlogic.npz"))
JS = json.load(open(os.path.join(HERE, "golden", "hostlogic.json")))


def test_convert_pt_to_numpy_host_path_is_the_references_bytes():
    img = torch.from_numpy(NPZ["convert_in_f16"])
    assert img.dtype == torch.float16
    got = np.stack(P.convert_pt_to_numpy(img))
    assert np.array_equal(got, NPZ["convert_out_u8"])
    got32 = np.stack(P.convert_pt_to_numpy(img.float()))
    assert np.array_equal(got32, NPZ["convert_out_u8_from_f32"])
    assert (NPZ["convert_out_u8"] != NPZ["convert_out_u8_from_f32"]).any()      # the fixture does separate fp16 from fp32 rounding


@pytest.mark.gpu
def test_sd_images_to_uint8_kernel_is_the_references_bytes():
    """The device kernel behind convert_pt_to_numpy for fp16 CUDA tensors: every byte equals what the reference's op
    sequence produced on the same fp16 tensor (edge values: +-1, 0.9995, out of range, +-65504, half-level neighbours)."""
    img = torch.from_numpy(NPZ["convert_in_f16"]).cuda()
    got = np.stack(P.convert_pt_to_numpy(img))
    assert got.dtype == np.uint8 and np.array_equal(got, NPZ["convert_out_u8"])


def _stub_scheduler(timesteps, order):
    return SimpleNamespace(timesteps=torch.tensor(timesteps, dtype=torch.int64), order=order,
                           config=SimpleNamespace(num_train_timesteps=1000),
                           set_timesteps=lambda n, device=None: None)


def test_retrieve_and_get_timesteps_rows():
    rows = JS["timesteps"]
    sched = {(r["order"], r["N"]): r for r in rows if r["fn"] == "retrieve_timesteps"}
    n_checked = 0
    for r in rows:
        base = sched[(r["order"], r["N"])]
        sch = _stub_scheduler(base["timesteps"], r["order"])
        if r["fn"] == "retrieve_timesteps":
            ts, n = P.retrieve_timesteps(sch, r["N"], "cpu")
            assert ts.tolist() == r["timesteps"] and n == r["num"]
        else:
            pipe = P.StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu")
            pipe.model = SimpleNamespace(scheduler=sch)
            ts, n = pipe.get_timesteps(r["N"], r["strength"], r["denoising_start"])
            assert ts.tolist() == r["timesteps"], r
            assert int(n) == r["num"], r
        n_checked += 1
    assert n_checked == len(rows) >= 150


def test_add_time_ids_rows():
    for r in JS["add_time_ids"]:
        ad, proj = r["addition_time_embed_dim"], r["projection_dim"]
        pipe = P.StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu")
        pipe.model = SimpleNamespace(base=SimpleNamespace(
            config=SimpleNamespace(addition_time_embed_dim=ad, projection_class_embeddings_input_dim=proj + 6 * ad),
            add_embedding=SimpleNamespace(linear_1=SimpleNamespace(in_features=r["expected"]))))
        args = (tuple(r["original_size"]), tuple(r["crop"]), tuple(r["target_size"]), torch.float16)
        if "error" in r:
            with pytest.raises(ValueError):
                pipe._get_add_time_ids(*args)
        else:
            ids = pipe._get_add_time_ids(*args)
            assert ids.dtype == torch.float16 and ids.tolist() == r["ids"] == r["neg_ids"]     # (:411-417: negatives = positives)


def test_prepare_mask_latents_rows():
    for r in JS["mask_latents"]:
        k = r["key"]
        pipe = P.StableDiffusionUnifiedPipeline(do_cfg=r["cfg"], device="cpu")
        mask = torch.from_numpy(NPZ[k + "_in"])
        mimg = torch.from_numpy(NPZ[k + "_img_in"]) if (k + "_img_in") in NPZ else None
        if "error" in r:
            with pytest.raises(ValueError):
                pipe.prepare_mask_latents(mask, mimg, r["batch_size"], r["height"], r["width"], torch.float32, None)
            continue
        m, lat = pipe.prepare_mask_latents(mask, mimg, r["batch_size"], r["height"], r["width"], torch.float32, None)
        assert np.array_equal(m.numpy(), NPZ[k + "_out"])
        assert (lat is not None) == r["latents"]
        if lat is not None:
            assert np.array_equal(lat.numpy(), NPZ[k + "_img_out"])
