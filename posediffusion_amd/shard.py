"""Multi-GPU: independent sequences shard embarrassingly.

One process per GPU (torchrun-style RANK / LOCAL_RANK / WORLD_SIZE).  The sampling path has NO
exchange step -- attention is within a sequence, the DDPM update is elementwise, GGS is per
sequence -- so the only collective is ONE final gather of the [B_local, N, 9] pose encodings
(RCCL all_gather over xGMI; 46 KB total at B = 64, latency bound).  The reference has no
inference sharding at all (test.py:153 loops every sequence on every rank).
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch
import torch.distributed as dist


def partition(n_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Block partition [start, stop) of n_items over ranks; the first (n_items % world) ranks get
    one extra item, so any B works and results do not depend on the world size."""
    base, rem = divmod(n_items, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_distributed(backend: str | None = None) -> Tuple[int, int, int]:
    """-> (rank, world_size, local_rank); initialises torch.distributed when WORLD_SIZE > 1
    (backend "nccl" = RCCL on ROCm; "gloo" for the CPU tests)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_poses(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """all_gather of the per-rank [B_local, N, 9] results into [n_total, N, 9] in global sequence
    order (ranks may hold different B_local; shorter shards are padded for the collective)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = [partition(n_total, world, r) for r in range(world)]
    cap = max(b - a for a, b in sizes)
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([bufs[r][: b - a] for r, (a, b) in enumerate(sizes)], dim=0)


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
