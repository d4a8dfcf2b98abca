"""One process per GPU: image sharding and weight distribution.

The reference's only multi-GPU mode is N independent processes, each reading all five checkpoints from disk and
taking ``paths[rank::world]`` (scripts/kaggle_inference_fixed.py:126-127, scripts/extract_val_cache.py:225).
Here rank 0 reads the checkpoints once and broadcasts ONE packed fp32 blob over RCCL/xGMI; the forward pass
has no collective.  Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU (tests).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
from typing import Dict, List, Optional, Sequence

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


def free_port() -> int:
    """A rendezvous port nobody listens on right now (a fixed port collides with the leftovers of an earlier run)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base: Optional[dict] = None) -> dict:
    """Environment of rank `rank` of a one-node, one-process-per-GPU job (what torch.distributed.run would export)."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC (RCCL between processes on this driver)
    return env


def launch_ranks(world: int, argv: Sequence[str], first_rank: int = 0, port: Optional[int] = None, relay=None):
    """Start ranks first_rank .. world-1 of a one-node job as fresh child processes running `argv` (one per GPU: the
    reference's own scheme, scripts/kaggle_inference_fixed.py:385-397 / scripts/extract_val_cache.py:421-426 -- N Popen's,
    then wait).  Plain fork + exec of NEW processes: the caller is never replaced and needs no GPU of its own.  Every child's
    stderr and stdout are relayed line by line (`relay(rank, line)`; default: rank 0's stdout to our stdout, everything
    else to stderr).  Returns (procs, threads, port); finish with `join_ranks`."""
    port = port or free_port()
    procs, threads = [], []

    def default_relay(rank, line, is_err):
        out = sys.stdout if (rank == 0 and not is_err) else sys.stderr
        out.write(line)
        out.flush()

    relay = relay or default_relay

    def pump(rank, stream, is_err):
        for line in stream:
            relay(rank, line, is_err)

    for r in range(first_rank, world):
        p = subprocess.Popen(list(argv), env=rank_env(r, world, port), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, bufsize=1)
        procs.append((r, p))
        for stream, is_err in ((p.stdout, False), (p.stderr, True)):
            t = threading.Thread(target=pump, args=(r, stream, is_err), daemon=True)
            t.start()
            threads.append(t)
    return procs, threads, port


def join_ranks(procs, threads, timeout: Optional[float] = None) -> int:
    """Wait for the children of `launch_ranks`; if one fails the others are terminated (a rank that died leaves the rest
    blocked in a collective).  Returns 0 or the first non-zero exit code."""
    import time
    rc, t0 = 0, time.monotonic()
    pending = dict(procs)
    while pending:
        for r, p in list(pending.items()):
            code = p.poll()
            if code is None:
                continue
            del pending[r]
            if code != 0 and rc == 0:
                rc = code
                print(f"[launch] rank {r} exited with code {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in pending.values():
                    q.terminate()
        if pending:
            if timeout is not None and time.monotonic() - t0 > timeout:
                for q in pending.values():
                    q.kill()
                rc = rc or 124
                break
            time.sleep(0.05)
    for _, p in procs:
        p.wait()
    for t in threads:
        t.join(timeout=5)
    return rc


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
