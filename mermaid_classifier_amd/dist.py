"""Patch-sharded feature extraction over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference's only scale-out is job-level sharding with results meeting in S3
(scripts/launch_processing.py:59-66, chunk_items round-robin over source IDs); inside one node the
same idea needs no storage hop: patches are independent, weights (8 MB) are replicated, each rank
extracts a contiguous block of the patch index range and ONE collective -- an all-gather of the
(n_local, 1280) fp32 blocks -- reassembles the feature matrix in patch order on every rank.
xGMI is point-to-point (7 links per GPU): a direct all-gather moves each rank's slice one hop on its
own link, so the exchange is ~N/8 of the feature bytes per link and tiny next to the compute.

torch.distributed is plumbing only (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""

from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`: the first n % world ranks get one extra patch."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    if n < 0:
        raise ValueError(f"n = {n} is negative")
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def chunk_items(items, n_chunks: int):
    """Round-robin chunks, the reference's job-level sharding rule (scripts/launch_processing.py:59-66):
    item i goes to chunk i % n_chunks; empty chunks are dropped."""
    if n_chunks < 1:
        raise ValueError("n_chunks must be >= 1")
    chunks = [list(items[i::n_chunks]) for i in range(n_chunks)]
    return [c for c in chunks if c]


def gather_features(local, n_total: int, group=None):
    """All-gather ragged (n_local, D) blocks into the (n_total, D) matrix, rows in global patch order.
    `local` is a torch tensor (cuda for nccl/RCCL, cpu for gloo).  Blocks are padded to the largest
    shard so a single all_gather_into_tensor moves them; padding rows are dropped afterwards."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n_total, rank, world)
    if local.dim() != 2 or local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {tuple(local.shape)}, expected ({hi - lo}, D)")
    d = local.shape[1]
    per = -(-n_total // world) if n_total else 0
    if per == 0:
        return local.new_zeros((0, d))
    padded = local
    if local.shape[0] != per:
        padded = local.new_zeros((per, d))
        padded[: local.shape[0]] = local
    out = local.new_empty((world * per, d))
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if n_total % world == 0:
        return out
    rows = [out[r * per: r * per + (shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0])] for r in range(world)]
    return torch.cat(rows, dim=0)


def extract_sharded(extract_fn: Callable, patches, group=None):
    """Each rank extracts its block of `patches` (indexable, length n_total; only the local block is
    touched) with `extract_fn(block) -> (n_local, D) tensor` and returns the gathered (n_total, D)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_total = len(patches)
    lo, hi = shard_range(n_total, rank, world)
    local = extract_fn(patches[lo:hi])
    return gather_features(local, n_total, group=group)
