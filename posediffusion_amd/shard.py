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


def init_distributed(backend: str | None = None, force: bool = False) -> Tuple[int, int, int]:
    """-> (rank, world_size, local_rank); initialises torch.distributed when WORLD_SIZE > 1
    (backend "nccl" = RCCL on ROCm; "gloo" for the CPU tests).  ``force``: initialise a process group of ONE rank too
    (bench.py --dry-dist: the RCCL call sites of the N > 1 path then execute on a single-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # PD_DIST_BACKEND=gloo: several ranks on ONE GPU (functional check of the N > 1 path on a 1-GPU box; RCCL needs
            # a device per rank)
            backend = os.environ.get("PD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_poses(local: torch.Tensor, n_total: int, force_collective: bool = False) -> torch.Tensor:
    """all_gather of the per-rank [B_local, N, 9] results into [n_total, N, 9] in global sequence
    order (ranks may hold different B_local; shorter shards are padded for the collective).
    ``force_collective``: run the collective in a group of one rank as well (bench.py --dry-dist)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return local
    world = dist.get_world_size()
    sizes = [partition(n_total, world, r) for r in range(world)]
    cap = max(b - a for a, b in sizes)
    dev = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:      # gloo gathers host tensors only
        local = local.cpu()
    pad = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs: List[torch.Tensor] = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([bufs[r][: b - a] for r, (a, b) in enumerate(sizes)], dim=0).to(dev)


def max_over_ranks(value: float, device, force_collective: bool = False) -> float:
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier(force_collective: bool = False):
    if dist.is_initialized() and (dist.get_world_size() > 1 or force_collective):
        dist.barrier()


def steps_per_pass(n_steps: int, b_step: int, engine_batch: int, min_passes: int = 2) -> int:
    """How many consecutive steps' shards (b_step sequences each) a rank runs as ONE engine pass: as many as fit
    `engine_batch` sequences, but no more than leaves the run's passes balanced -- ceil(S / engine_batch) passes (at least
    `min_passes` = two when there are two steps: a second context overlaps its denoiser steps with the first one's GGS
    launches) of equal length, instead of full passes plus a short straggler that runs alone at the end."""
    if b_step <= 0 or n_steps <= 0:
        return 1
    n_pass = max(-(-n_steps * b_step // engine_batch), min(max(1, min_passes), n_steps))
    return max(1, min(-(-n_steps // n_pass), max(1, engine_batch // b_step)))


def strong_schedule(n_steps: int, step_seqs: int, world: int, rank: int, engine_batch: int, min_passes: int = 2):
    """Strong scaling of `n_steps` steps of `step_seqs` independent sequences each over `world` ranks (bench.py):
    every step is block-partitioned, rank r owning rows [g0, g1) of each; a rank runs the shards of `group` consecutive
    steps as ONE engine pass of at most `engine_batch` sequences (`steps_per_pass`).  -> (g0, g1, group, passes) where
    passes[p] = number of this rank's sequences in pass p (group * (g1 - g0), the last pass possibly fewer)."""
    g0, g1 = partition(step_seqs, world, rank)
    b_step = g1 - g0
    if b_step <= 0:
        return g0, g1, 1, []
    if b_step > engine_batch:
        raise ValueError(f"rank {rank} holds {b_step} sequences of every step, more than one engine pass takes ({engine_batch}): "
                         "raise the engine batch or use more ranks")
    group = steps_per_pass(n_steps, b_step, engine_batch, min_passes)
    passes = [min(group, n_steps - s0) * b_step for s0 in range(0, n_steps, group)]
    return g0, g1, group, passes


def step_rows(step: int, group: int, b_step: int):
    """Where step `step`'s shard sits: (pass index, first row, end row) inside that pass's [passes[p], N, 9] result."""
    return step // group, (step % group) * b_step, (step % group + 1) * b_step
