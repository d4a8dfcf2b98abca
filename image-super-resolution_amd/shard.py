"""One process per GPU: image sharding and weight distribution.

The reference's only multi-GPU mode is N independent processes, each reading all five checkpoints from disk and
taking ``paths[rank::world]`` (scripts/kaggle_inference_fixed.py:126-127, scripts/extract_val_cache.py:225).
Here rank 0 reads the checkpoints once and broadcasts ONE packed fp32 blob over RCCL/xGMI; the forward pass
has no collective.  Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU (tests).
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend: str = None):
    rank, world = rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    return rank, world


def shard(items: Sequence, rank: int, world: int) -> List:
    """The reference's strided partition (balances equal-sized DIV2K images)."""
    return list(items[rank::world])


def _layout(weights: Dict[str, dict]):
    return [(m, k, tuple(v.shape)) for m in sorted(weights) for k, v in sorted(weights[m].items())]


def broadcast_weights(weights: Dict[str, dict], device, src: int = 0) -> Dict[str, dict]:
    """Every rank passes a dict with identical keys/shapes (ranks other than `src` only need the shapes:
    weights.random_weights(shapes_only=True)); the values of rank `src` win.  One flat buffer -> one broadcast (743 MB fp32 for the full model: a few ms over xGMI)."""
    rank, world = rank_world()
    if world == 1 or not dist.is_initialized():
        return weights
    layout = _layout(weights)
    total = sum(int(torch.tensor(s).prod()) if s else 1 for _, _, s in layout)
    flat = torch.empty(total, dtype=torch.float32, device=device)
    if rank == src:
        off = 0
        for m, k, s in layout:
            n = weights[m][k].numel()
            flat[off:off + n] = weights[m][k].reshape(-1).to(device=device, dtype=torch.float32)
            off += n
    dist.broadcast(flat, src=src)
    out, off = {}, 0
    for m, k, s in layout:
        n = 1
        for d in s:
            n *= d
        out.setdefault(m, {})[k] = flat[off:off + n].reshape(s)
        off += n
    return out


def gather_stats(values: Sequence[float], device) -> List[List[float]]:
    """All ranks' (pixels, seconds, squared-error ...) tuples on every rank; single tiny collective at the end."""
    rank, world = rank_world()
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if world == 1 or not dist.is_initialized():
        return [t.tolist()]
    bufs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(bufs, t)
    return [b.tolist() for b in bufs]
