"""Multi-GPU set build: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI).

Reads shard embarrassingly (each rank counts and later corrects its own reads); the only
exchange step of the path is the k-mer counts, once per job:

  dense strategy   every rank holds the full 2^(2k-1)-byte u8 table.  Counts are clamped to
                   min(c, a+1) first, so that with world*(a+1) <= 255 an u8 SUM all-reduce cannot
                   wrap and  sum_r min(c_r, a+1) > a  <=>  sum_r c_r > a  (exact); otherwise the
                   chunks are widened to int32 for the reduction.  After it every rank thresholds
                   locally: the solidity bitset is replicated without a second collective.

The exchange is written against a tiny "counter-like" protocol (clamp(cap, stream) and
counts_tensor()), so the world_size-2 gloo tests drive exactly this code on CPU tensors.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

CHUNK_BYTES = 1 << 30


class _DevBuf:
    """zero-copy torch view over a raw device pointer owned by libbrx"""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_view(ptr: int, nbytes: int) -> torch.Tensor:
    return torch.as_tensor(_DevBuf(ptr, nbytes), device="cuda")


class GpuCounterAdapter:
    """counter-like wrapper over br_amd.Counter (dense strategy)"""

    def __init__(self, counter):
        self.counter = counter

    def clamp(self, cap: int, stream: Optional[int]) -> None:
        self.counter.clamp(cap, stream)

    def counts_tensor(self) -> torch.Tensor:
        ptr, n = self.counter.device_counts()
        return device_view(ptr, n)


def exact_cap(abundance: int, world: int) -> Optional[int]:
    """largest clamp that keeps an u8 SUM over `world` ranks exact, or None if there is none"""
    cap = abundance + 1
    return cap if world * cap <= 255 else None


def allreduce_counts(counter_like, abundance: int, world: int, stream: Optional[int] = None,
                     chunk_bytes: int = CHUNK_BYTES) -> None:
    """in place: counts <- min(255, sum over ranks) as far as `> abundance` can tell."""
    if world <= 1:
        return
    cap = exact_cap(abundance, world)
    counter_like.clamp(cap if cap is not None else 255, stream)
    t = counter_like.counts_tensor()
    n = t.numel()
    for lo in range(0, n, chunk_bytes):
        part = t[lo:lo + chunk_bytes]
        if cap is not None:
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
        else:
            wide = part.to(torch.int32)
            dist.all_reduce(wide, op=dist.ReduceOp.SUM)
            part.copy_(wide.clamp_(max=255).to(torch.uint8))


class SetExchange:
    def __init__(self, world: int, rank: int):
        self.world, self.rank = world, rank

    def reduce_counts(self, counter, abundance: int, stream: Optional[int] = None) -> None:
        allreduce_counts(GpuCounterAdapter(counter), abundance, self.world, stream)


def shard_range(n_items: int, world: int, rank: int):
    """contiguous block of records for a rank: concatenating the ranks restores input order"""
    lo = n_items * rank // world
    hi = n_items * (rank + 1) // world
    return lo, hi
