"""Batch-of-latents sharding over the GPUs of one node: one process per GPU, RCCL over xGMI.

The reference's inference path is single-device (`/root/reference/runpod-worker/rp_handler.py:15,36`);
sharding is new work.  Each latent's trajectory depends only on its own noise and
its own text embedding, so the batch is split contiguously over ranks with no communication inside
the denoise loop.  Two collectives per run, both tiny next to one xGMI link-second:
  1. broadcast of the text embeddings (cond + uncond) from rank 0,
  2. all-gather of the decoded images.
`backend="nccl"` is RCCL on ROCm; the CPU tests use gloo with world_size 2.
"""
from __future__ import annotations

import os
from typing import Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: Optional[str] = None) -> Tuple[int, int]:
    """Initialise torch.distributed from the torchrun environment; returns (rank, world_size)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # SD_DIST_BACKEND=gloo lets several ranks rehearse on a box with fewer GPUs than ranks
            backend = os.environ.get("SD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world


def world_size() -> int:
    """Number of ranks the collectives see (1 when torch.distributed is not initialised)."""
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_bounds(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split; ranks < total % world get one extra sample."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(t.shape[0], rank, world)
    return t[lo:hi]


def _host_staged() -> bool:
    """gloo moves host memory: device tensors are staged through the CPU (rehearsal path only)."""
    return dist.get_backend() == "gloo"


def broadcast_tensors(tensors: Sequence[torch.Tensor], src: int = 0):
    """In-place broadcast of already-allocated tensors (text embeddings) from `src`."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            if _host_staged() and t.is_cuda:
                h = t.cpu()
                dist.broadcast(h, src=src)
                t.copy_(h)
            else:
                dist.broadcast(t, src=src)
    return tensors


def all_gather_batch(local: torch.Tensor, total: int) -> torch.Tensor:
    """Concatenate per-rank shards along dim 0 on every rank.  Even shards (the benchmark's and any batch that is a
    multiple of the world size): ONE `all_gather_into_tensor` straight into the [total, ...] result -- one RCCL
    collective over xGMI, no list of temporaries, no host-side padding.  Uneven shards are padded to the largest."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return local
    world = dist.get_world_size()
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in sizes)
    dev = local.device
    if all(hi - lo == max_n for lo, hi in sizes):
        src = local.contiguous()
        if _host_staged() and src.is_cuda:
            src = src.cpu()
        out = torch.empty((total,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(out, src)
        return out.to(dev)
    pad = local
    if local.shape[0] < max_n:
        pad = torch.cat([local, local.new_zeros((max_n - local.shape[0],) + tuple(local.shape[1:]))])
    if _host_staged() and pad.is_cuda:
        pad = pad.cpu()
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad.contiguous())
    return torch.cat([o[: hi - lo] for o, (lo, hi) in zip(out, sizes)], dim=0).to(dev)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(x: float, device) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return x
    t = torch.tensor([x], dtype=torch.float64, device="cpu" if _host_staged() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sharded_txt2img(pipeline, model, latents_full: torch.Tensor, prompt_embeds_full: torch.Tensor,
                    negative_embeds_full: torch.Tensor, rank: int, world: int,
                    pooled_full: Optional[torch.Tensor] = None, negative_pooled_full: Optional[torch.Tensor] = None,
                    broadcast: bool = True, **call_kwargs) -> torch.Tensor:
    """Run the pipeline on this rank's slice of the batch and return the full gathered batch.

    `latents_full` is generated once for the whole batch from the single seeded generator
    (`sd_unified_pipeline.py:773-781` semantics) so sharded == unsharded per sample.  `broadcast=False`: the text
    embeddings were broadcast already (a caller that runs several passes over the same prompts broadcasts once).
    """
    total = latents_full.shape[0]
    extra = [t for t in (pooled_full, negative_pooled_full) if t is not None]
    if broadcast:
        broadcast_tensors([prompt_embeds_full, negative_embeds_full] + extra)
    lat = shard(latents_full, rank, world)
    pe = shard(prompt_embeds_full, rank, world)
    ne = shard(negative_embeds_full, rank, world)
    if pooled_full is not None:      # SDXL pooled text embeddings ride along with the same split
        call_kwargs = dict(call_kwargs, pooled_prompt_embeds=shard(pooled_full, rank, world),
                           negative_pooled_prompt_embeds=shard(negative_pooled_full, rank, world))
    images = pipeline(model, prompt_embeds=pe, negative_prompt_embeds=ne, latents=lat, **call_kwargs)
    return all_gather_batch(images, total)
