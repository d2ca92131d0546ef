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


class FeatureGatherer:
    """The all-gather of one rank block per rank, with every buffer allocated ONCE: the padded send block (ragged shards
    only) and the (world * per, D) receive matrix.  `gather(local)` issues one all_gather_into_tensor and returns the
    (n_total, D) matrix in global patch order -- a view of the receive buffer when the shards are even, otherwise one
    torch.cat of the valid rows.  `gather(local, async_op=True)` returns (work, finish): wait on `work`, then call
    `finish()` for the matrix (the collective runs beside whatever the caller launches meanwhile).
    BASELINE configs[3] is exactly one such call per rank: 1 M patches -> one gather of (125 000, 1280) blocks."""

    def __init__(self, n_total: int, d: int, like, group=None):
        import torch.distributed as dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.n_total, self.d = int(n_total), int(d)
        self.lo, self.hi = shard_range(self.n_total, self.rank, self.world)
        self.per = -(-self.n_total // self.world) if self.n_total else 0
        self.sizes = [shard_range(self.n_total, r, self.world) for r in range(self.world)]
        self.even = self.n_total % self.world == 0
        self.out = like.new_empty((self.world * self.per, self.d))
        self.padded = None if (self.hi - self.lo) == self.per else like.new_zeros((self.per, self.d))

    def _finish(self):
        import torch
        if self.even:
            return self.out
        return torch.cat([self.out[r * self.per: r * self.per + (b - a)] for r, (a, b) in enumerate(self.sizes)], dim=0)

    def gather(self, local, async_op: bool = False):
        import torch.distributed as dist
        if local.dim() != 2 or tuple(local.shape) != (self.hi - self.lo, self.d):
            raise ValueError(f"rank {self.rank} holds {tuple(local.shape)}, expected ({self.hi - self.lo}, {self.d})")
        if self.per == 0:
            return (None, lambda: local.new_zeros((0, self.d))) if async_op else local.new_zeros((0, self.d))
        send = local
        if self.padded is not None:
            self.padded[: local.shape[0]].copy_(local)
            send = self.padded
        work = dist.all_gather_into_tensor(self.out, send.contiguous(), group=self.group, async_op=async_op)
        if async_op:
            return work, self._finish
        return self._finish()


class NativeGatherer:
    """The same exchange through the C ABI alone (include/mmc.h: mmc_dist_unique_id / mmc_dist_create /
    mmc_gather_features -- RCCL resolved by the library, no torch.distributed): what a non-Python host binds.  Rank 0
    makes the id with `NativeGatherer.unique_id()` and hands the 128 bytes to the other ranks out of band; every rank
    then constructs `NativeGatherer(id, rank, world, device)` (a collective) and calls `gather(local, n_total)` with its
    contiguous block (shard_range) as a cuda tensor; the (n_total, D) matrix comes back on every rank, asynchronously on
    the current stream."""

    ID_BYTES = 128

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        buf = (C.c_ubyte * NativeGatherer.ID_BYTES)()
        _lib.check(_lib.lib().mmc_dist_unique_id(buf))
        return bytes(buf)

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int = 0):
        import ctypes as C
        from . import _lib
        if len(unique_id) != self.ID_BYTES:
            raise ValueError(f"unique id must be {self.ID_BYTES} bytes, got {len(unique_id)}")
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        self._lib = _lib
        self._h = C.c_void_p()
        buf = (C.c_ubyte * self.ID_BYTES).from_buffer_copy(unique_id)
        _lib.check(_lib.lib().mmc_dist_create(buf, self.rank, self.world, self.device, C.byref(self._h)))

    def gather(self, local, n_total: int):
        import ctypes as C
        import torch
        lo, hi = shard_range(int(n_total), self.rank, self.world)
        if local.dim() != 2 or local.shape[0] != hi - lo or local.dtype != torch.float32 or not local.is_cuda:
            raise ValueError(f"rank {self.rank} must hold a cuda float32 ({hi - lo}, D) block, got {tuple(local.shape)} {local.dtype}")
        local = local.contiguous()
        d = local.shape[1]
        out = local.new_empty((int(n_total), d))
        counts = (C.c_int64 * self.world)(*[b - a for a, b in (shard_range(int(n_total), r, self.world) for r in range(self.world))])
        even = int(n_total) % self.world == 0
        st = torch.cuda.current_stream(local.device).cuda_stream
        self._lib.check(self._lib.lib().mmc_gather_features(self._h, local.data_ptr(), hi - lo, d, None if even else counts,
                                                            out.data_ptr(), st))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lib().mmc_dist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gather_features(local, n_total: int, group=None):
    """All-gather ragged (n_local, D) blocks into the (n_total, D) matrix, rows in global patch order.
    `local` is a torch tensor (cuda for nccl/RCCL, cpu for gloo).  Blocks are padded to the largest
    shard so a single all_gather_into_tensor moves them; padding rows are dropped afterwards.
    One-shot form (allocates its buffers): a loop that gathers repeatedly keeps a FeatureGatherer."""
    if local.dim() != 2:
        raise ValueError(f"expected an (n_local, D) block, got {tuple(local.shape)}")
    return FeatureGatherer(n_total, local.shape[1], local, group=group).gather(local)


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
