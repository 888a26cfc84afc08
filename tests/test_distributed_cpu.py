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


# ---- bench.py launcher: `python bench.py --gpus N` must itself produce N ranks ----

def _bench(*argv, **env):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=e, capture_output=True,
                       text=True, timeout=240)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_launcher_spawns_n_ranks():
    rc, js, err = _bench("--gpus", "2", "--rehearse")
    assert rc == 0, err
    assert js == {"rehearsal": True, "n_gpus": 2, "ranks_counted": 2}


def test_bench_launcher_fails_when_a_rank_fails():
    rc, js, err = _bench("--gpus", "2", "--rehearse", SD_BENCH_REHEARSE_FAIL_RANK="1")
    assert rc != 0 and "rank exit codes" in err


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_bench_refuses_more_gpus_than_visible():
    rc, js, err = _bench("--gpus", "2")
    assert rc != 0 and js is None and "refusing" in err


def test_bench_refuses_mislabelled_world():
    rc, js, err = _bench("--gpus", "4", WORLD_SIZE="2", RANK="0", MASTER_PORT="1")
    assert rc != 0 and js is None and "mislabelled" in err
