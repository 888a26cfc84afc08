"""N > 1 path on CPU: world_size-2 gloo processes shard a batch, run the pipeline host code on
oracle-backed doubles, all-gather -- result must equal the unsharded run sample for sample."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stablediffusion_amd import config, distributed as sdd, schedulers, weights
from stablediffusion_amd.pipeline import SDModelWrapper, StableDiffusionUnifiedPipeline


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(total):
    g = torch.Generator().manual_seed(9)
    ucfg = config.tiny_unet()
    return (torch.randn(total, 4, 8, 8, generator=g), torch.randn(total, 77, ucfg.cross_attention_dim, generator=g),
            torch.randn(total, 77, ucfg.cross_attention_dim, generator=g))


def _model():
    from doubles import OracleUNet, OracleVAE
    ucfg, vcfg = config.tiny_unet(), config.tiny_vae()
    uw = weights.synth_state_dict(weights.unet_manifest(ucfg), seed=4, perturb=0.1)
    vw = weights.synth_state_dict(weights.vae_manifest(vcfg), seed=5, perturb=0.1)
    return SDModelWrapper(base=OracleUNet(ucfg, uw), vae=OracleVAE(vcfg, vw), scheduler=schedulers.DDIMScheduler(),
                          device="cpu")


def _worker(rank, world, port, total, out_path):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    r, w = sdd.init("gloo")
    assert (r, w) == (rank, world)
    lat, pe, ne = _inputs(total)
    if rank != 0:
        pe.zero_(); ne.zero_()                      # must arrive through the broadcast
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu")
    imgs = sdd.sharded_txt2img(pipe, _model(), lat, pe, ne, rank, world, num_inference_steps=2, height=64, width=64)
    assert sdd.max_over_ranks(float(rank), "cpu") == world - 1
    if rank == 0:
        torch.save(imgs, out_path)
    sdd.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [2, 3])
def test_sharded_equals_unsharded(tmp_path, total):
    out = str(tmp_path / "imgs.pt")
    mp.spawn(_worker, args=(2, _free_port(), total, out), nprocs=2, join=True)
    sharded = torch.load(out, weights_only=True)
    torch.set_num_threads(2)
    lat, pe, ne = _inputs(total)
    pipe = StableDiffusionUnifiedPipeline(do_cfg=True, device="cpu")
    full = pipe(_model(), prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, num_inference_steps=2,
                height=64, width=64)
    assert sharded.shape == full.shape == (total, 3, 64, 64)
    assert torch.allclose(sharded, full, atol=1e-5)


def test_shard_bounds():
    assert [sdd.shard_bounds(32, r, 8) for r in range(8)] == [(4 * r, 4 * r + 4) for r in range(8)]
    assert [sdd.shard_bounds(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [sdd.shard_bounds(1, r, 2) for r in range(2)] == [(0, 1), (1, 1)]
