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

import os
from typing import Optional

import torch
import torch.distributed as dist

CHUNK_BYTES = 1 << 30


# The collectives below are torch.distributed's, on device tensors over RCCL.  With the "gloo" backend (two ranks
# rehearsing the exchange on ONE GPU, where RCCL cannot run; tests/test_gpu_parity.py) device tensors are staged
# through host memory, since gloo's all_to_all takes CPU tensors only.
def _staged(t: torch.Tensor) -> bool:
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_reduce(t: torch.Tensor, op) -> None:
    if _staged(t):
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)


def _all_gather(outs, t: torch.Tensor) -> None:
    if _staged(t):
        hs = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        dist.all_gather(hs, t.cpu())
        for o, h in zip(outs, hs):
            o.copy_(h)
    else:
        dist.all_gather(outs, t)


def _all_to_all_single(out: torch.Tensor, inp: torch.Tensor, output_split_sizes, input_split_sizes) -> None:
    if _staged(inp):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h, inp.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes)
        out.copy_(h)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes)


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
            _all_reduce(part, dist.ReduceOp.SUM)
        else:
            wide = part.to(torch.int32)
            _all_reduce(wide, dist.ReduceOp.SUM)
            part.copy_(wide.clamp_(max=255).to(torch.uint8))


class SetExchange:
    """per-job exchange step for bench.py / multi-GPU drivers"""

    def __init__(self, world: int, rank: int):
        self.world, self.rank = world, rank

    def reduce_counts(self, counter, abundance: int, stream: Optional[int] = None) -> None:
        """dense strategy: in-place all-reduce of the u8 table; the caller thresholds afterwards"""
        allreduce_counts(GpuCounterAdapter(counter), abundance, self.world, stream)

    def build_partitioned(self, counter, solid, abundance: int, stream: Optional[int] = None) -> None:
        """partitioned strategy: `counter` has counted this rank's reads (one add_batch); on return
        `solid` holds the set of ALL ranks' reads and the counter is empty again"""
        eng = GpuPartitionedEngine(counter, solid, stream)
        exchange_partitioned(eng, abundance, self.world, self.rank)
        torch.cuda.current_stream().synchronize()
        counter.reset(stream)
        eng.release()


class AbiExchange:
    """The exchange step behind the C ABI (include/brx.h brx_comm_* / brx_exchange_*): libbrx calls librccl itself
    (ncclSend/ncclRecv groups for the key all-to-all, all-gather of the solid-hash lists).  All this class adds is
    the hand-over of the 128-byte communicator id from rank 0 to the other ranks, through the process group the
    host already has (any out-of-band channel would do)."""

    def __init__(self, world: int, rank: int, device: int, ident: Optional[bytes] = None):
        """`ident`: the 128 bytes of `unique_id()` as rank 0 made them, when the host has its own channel for them
        (a file, a socket); without it they travel through the torch.distributed process group."""
        import ctypes as C
        from . import _lib
        self.world, self.rank, self.device = world, rank, device
        L = _lib.lib()
        if ident is not None:
            ident = (C.c_uint8 * 128).from_buffer_copy(ident)
        else:
            ident = (C.c_uint8 * 128)()
            if rank == 0:
                _lib.check(L.brx_comm_unique_id(ident))
            if world > 1:
                box = [bytes(ident)]
                dist.broadcast_object_list(box, src=0)
                ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        self._h = C.c_void_p()
        _lib.check(L.brx_comm_init(ident, world, rank, device, C.byref(self._h)))

    @staticmethod
    def unique_id() -> bytes:
        """brx_comm_unique_id: rank 0 makes it, every rank passes the same bytes to the constructor"""
        import ctypes as C
        from . import _lib
        ident = (C.c_uint8 * 128)()
        _lib.check(_lib.lib().brx_comm_unique_id(ident))
        return bytes(ident)

    def build_partitioned(self, counter, solid, abundance: int, stream: Optional[int] = None) -> None:
        from . import _lib
        _lib.check(_lib.lib().brx_exchange_build_partitioned(self._h, counter._h, abundance, solid._h, stream))

    def reduce_counts(self, counter, abundance: int, stream: Optional[int] = None) -> None:
        from . import _lib
        _lib.check(_lib.lib().brx_exchange_reduce_counts(self._h, counter._h, abundance, stream))

    def last_stats(self) -> dict:
        import ctypes as C
        from . import _lib
        v = (C.c_uint64 * 8)()
        _lib.check(_lib.lib().brx_comm_last_stats(self._h, v))
        names = ["key_bytes_sent", "key_bytes_received", "keys_counted_here", "solid_here", "solid_job", "all_to_all_us",
                 "exchange_us", "largest_message_keys"]
        return dict(zip(names, [int(x) for x in v]))

    def close(self) -> None:
        from . import _lib
        if getattr(self, "_h", None) and self._h.value:
            _lib.lib().brx_comm_free(self._h)
            self._h.value = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_range(n_items: int, world: int, rank: int):
    """contiguous block of records for a rank: concatenating the ranks restores input order"""
    lo = n_items * rank // world
    hi = n_items * (rank + 1) // world
    return lo, hi


# ---------------------------------------------------------------------------------------------------
# partitioned strategy: exchange KEYS, not the count vector
#
# After its local level-1 pass every rank holds its canonical hashes grouped by their first radix
# digit.  Rank r owns the contiguous digit range [r*B1/world, (r+1)*B1/world): one all_to_all moves
# every key to the owner of its digit (4 bytes per k-mer, (world-1)/world of them leave the GPU),
# the owner finishes counting its range, and the solid set is replicated SPARSELY: each rank lists the
# set bits of its range (8 bytes per solid k-mer) and one all_gather hands every list to everybody,
# who ORs them into a bitset that the final counting pass already zeroed outside the owned range.
# For k=19 / 1 Gbp per GPU that is ~3.5 GB + ~0.15 GB per rank over xGMI instead of the 120 GB + 15 GB
# of a reduce-scatter + all-gather on the dense u8 vector.
# ---------------------------------------------------------------------------------------------------
def owner_bounds(n_buckets: int, world: int):
    return [r * n_buckets // world for r in range(world + 1)]


A2A_CHUNK_ELEMS = 1 << 27  # per peer and per call: 512 MiB of int32 keys


def all_to_all_chunked(recv: torch.Tensor, send: torch.Tensor, recv_counts, send_counts, chunk: Optional[int] = None) -> None:
    """all_to_all_single with per-peer messages capped at `chunk` elements.  One 4 GB message per peer
    (1e9 int32 keys, the single-rank case of the k=19 / 1 Gbp job) was observed to arrive TRUNCATED at
    2^31 bytes through torch.distributed + RCCL on this stack, silently; smaller pieces are exact."""
    chunk = chunk or A2A_CHUNK_ELEMS
    world = len(send_counts)
    send_off = [0]
    recv_off = [0]
    for r in range(world):
        send_off.append(send_off[-1] + send_counts[r])
        recv_off.append(recv_off[-1] + recv_counts[r])
    rounds = max([(c + chunk - 1) // chunk for c in list(send_counts) + list(recv_counts)] + [0])
    # every rank must run the same number of rounds
    t = torch.tensor([rounds], dtype=torch.int64, device=send.device)
    _all_reduce(t, dist.ReduceOp.MAX)
    rounds = int(t.item())
    for c in range(rounds):
        s_parts, r_sizes, s_sizes = [], [], []
        for r in range(world):
            lo = min(c * chunk, send_counts[r])
            hi = min((c + 1) * chunk, send_counts[r])
            s_parts.append(send[send_off[r] + lo:send_off[r] + hi])
            s_sizes.append(hi - lo)
            lo = min(c * chunk, recv_counts[r])
            hi = min((c + 1) * chunk, recv_counts[r])
            r_sizes.append(hi - lo)
        sbuf = torch.cat(s_parts) if world > 1 else s_parts[0].contiguous()
        rbuf = torch.empty(sum(r_sizes), dtype=recv.dtype, device=recv.device)
        _all_to_all_single(rbuf, sbuf, r_sizes, s_sizes)
        pos = 0
        for r in range(world):
            lo = min(c * chunk, recv_counts[r])
            recv[recv_off[r] + lo:recv_off[r] + lo + r_sizes[r]] = rbuf[pos:pos + r_sizes[r]]
            pos += r_sizes[r]


def exchange_partitioned(engine, abundance: int, world: int, rank: int) -> None:
    """engine protocol (GPU: GpuPartitionedEngine below; tests drive it with CPU tensors over gloo):
       l1() -> (keys int32[n], l1off int64[B1+1]); add_segment(keys, l1off); finish(abundance);
       extract(first_hash, n_hashes) -> int64[m]; or_keys(int64[m]); n_hashes (int)."""
    keys, l1off = engine.l1()
    B1 = l1off.numel() - 1
    bounds = owner_bounds(B1, world)
    tables = [torch.empty_like(l1off) for _ in range(world)]
    _all_gather(tables, l1off)
    tab_h = [t.cpu() for t in tables]
    mine_h = tab_h[rank]
    send_counts = [int(mine_h[bounds[r + 1]] - mine_h[bounds[r]]) for r in range(world)]
    lo, hi = bounds[rank], bounds[rank + 1]
    recv_counts = [int(t[hi] - t[lo]) for t in tab_h]
    recv = torch.empty(sum(recv_counts), dtype=keys.dtype, device=keys.device)
    all_to_all_chunked(recv, keys[:sum(send_counts)], recv_counts, send_counts)
    pos = 0
    for s in range(world):
        t = tables[s]
        seg = t.clamp(min=t[lo], max=t[hi]) - t[lo]  # full table: 0 below the owned range, the count above it
        engine.add_segment(recv[pos:pos + recv_counts[s]], seg.contiguous())
        pos += recv_counts[s]
    engine.finish(abundance)

    per_bucket = engine.n_hashes // B1
    solid = engine.extract(lo * per_bucket, (hi - lo) * per_bucket)
    n_mine = torch.tensor([solid.numel()], dtype=torch.int64, device=keys.device)
    counts = [torch.empty_like(n_mine) for _ in range(world)]
    _all_gather(counts, n_mine)
    counts_h = [int(c.item()) for c in counts]
    maxn = max(max(counts_h), 1)
    padded = torch.zeros(maxn, dtype=torch.int64, device=keys.device)
    padded[:solid.numel()] = solid
    gathered = [torch.empty_like(padded) for _ in range(world)]
    _all_gather(gathered, padded)
    for s in range(world):
        if s != rank and counts_h[s]:
            engine.or_keys(gathered[s][:counts_h[s]])
    if hasattr(engine, "all_keys"):
        # every solid k-mer of the job is in hand as a list: hand it to the set (probe index without a scan)
        engine.all_keys([solid] + [gathered[s][:counts_h[s]] for s in range(world) if s != rank and counts_h[s]])


class GpuPartitionedEngine:
    """exchange_partitioned's engine over ONE br_amd.Counter (partitioned strategy): it has counted this
    rank's reads (level 1); after the keys have been exchanged it is reset and re-used to finish the
    segments of the owned digit range, so the radix workspace is allocated once."""

    def __init__(self, counter, solid, stream: Optional[int] = None):
        self.counter, self.solid, self.stream = counter, solid, stream
        self.n_hashes = solid.n_hashes()
        self._keep = []
        self._sent = False

    def l1(self):
        pk, po, nb, nk = self.counter.l1_view()
        keys = device_view(pk, max(nk, 1) * 4).view(torch.int32)[:nk]
        l1off = device_view(po, (nb + 1) * 8).view(torch.int64)
        return keys, l1off

    def add_segment(self, keys: torch.Tensor, l1off: torch.Tensor) -> None:
        if not self._sent:
            # the all_to_all that read the local level-1 buffer has been issued on this stream; make sure
            # it is complete before the counter forgets (and later re-uses) that buffer
            torch.cuda.current_stream().synchronize()
            self.counter.reset(self.stream)
            self._sent = True
        self._keep.append((keys, l1off))
        if keys.numel():
            self.counter.add_partitioned_device(keys.data_ptr(), l1off.data_ptr(), keys.numel())

    def finish(self, abundance: int) -> None:
        self.counter.finish_into(abundance, self.solid, self.stream)

    def extract(self, first_hash: int, n_hashes: int) -> torch.Tensor:
        # the partitioned finish leaves the list of solid hashes with the set (only owned buckets were counted)
        kl = self.solid.keylist_device(self.stream)
        if kl is not None:
            ptr, n = kl
            return device_view(ptr, max(n, 1) * 8).view(torch.int64)[:n].clone()
        # everything outside the owned range is still zero, so the whole-set popcount sizes the list
        torch.cuda.current_stream().synchronize()
        cap = self.solid.popcount() + 64
        buf = torch.empty(cap, dtype=torch.int64, device="cuda")
        n = self.solid.extract_keys_device(first_hash, n_hashes, buf.data_ptr(), cap, self.stream)
        return buf[:n]

    def or_keys(self, keys: torch.Tensor) -> None:
        if self.solid.bits_state() != 0:
            return  # no (current) bit vector to OR into: all_keys() rebuilds the probe index from every rank's list
        keys = keys.contiguous()
        self._keep.append(keys)
        self.solid.or_keys_device(keys.data_ptr(), keys.numel(), self.stream)

    def all_keys(self, lists) -> None:
        if self.solid.index_info()["wanted"] or self.solid.bits_state() != 0:
            keys = torch.cat(lists) if len(lists) > 1 else lists[0]
            self._keep.append(keys)
            self.solid.index_build_from_keys_device(keys.data_ptr(), keys.numel(), 0, 0, self.stream)

    def release(self) -> None:
        """drop the references that kept received segments alive (after the set is final)"""
        self._keep.clear()
