"""world_size-2 gloo tests of the multi-GPU exchange logic (br_amd/dist.py) on CPU tensors.
The 'counter' here is an oracle-backed stand-in defined in this test; the product path uses
br_amd.Counter (HIP) behind the same protocol."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleCounter:
    def __init__(self, counts: np.ndarray):
        self.t = torch.from_numpy(counts)

    def clamp(self, cap, stream):
        self.t.clamp_(max=cap)

    def counts_tensor(self):
        return self.t


def _worker(rank, world, port, k, abundance, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from br_amd import dist as bd
    from oracle import oracle as O
    from tests.conftest import read_fasta
    reads = read_fasta(os.path.join(ROOT, "tests", "golden", "raw.fasta"))[1][:60]
    lo, hi = bd.shard_range(len(reads), world, rank)
    c = OracleCounter(O.count_reads(k, reads[lo:hi]))
    bd.allreduce_counts(c, abundance, world, None, chunk_bytes=1 << 16)
    mine = O.Solid.from_count(k, c.t.numpy(), abundance).to_bytes()
    ref = O.Solid.from_count(k, O.count_reads(k, reads), abundance).to_bytes()
    q.put((rank, mine == ref))
    dist.destroy_process_group()


@pytest.mark.parametrize("abundance", [2, 200])   # 200: world*(a+1) > 255 -> widened reduction
def test_sharded_count_allreduce_matches_single(abundance):
    world, k = 2, 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + abundance
    procs = [ctx.Process(target=_worker, args=(r, world, port, k, abundance, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_exact_cap_and_shards():
    sys.path.insert(0, ROOT)
    from br_amd import dist as bd
    assert bd.exact_cap(3, 8) == 4 and bd.exact_cap(30, 8) == 31 and bd.exact_cap(31, 8) is None
    cover = []
    for r in range(8):
        lo, hi = bd.shard_range(1003, 8, r)
        cover += list(range(lo, hi))
    assert cover == list(range(1003))


# ---- partitioned strategy: key exchange (br_amd.dist.exchange_partitioned) over gloo -------------------
class NumpyPartEngine:
    """CPU stand-in for GpuPartitionedEngine: same protocol, numpy + the oracle's hashes.  The level-1
    layout is what libbrx produces: hashes grouped by their top `l1_bits` bits, digit stripped."""

    def __init__(self, k, reads, l1_bits):
        from oracle import oracle as O
        self.k, self.nbits, self.l1_bits = k, 2 * k - 1, l1_bits
        self.n_hashes = 1 << self.nbits
        h = np.concatenate([O.hashes(k, r) for r in reads] + [np.zeros(0, dtype=np.uint64)])
        rem = self.nbits - l1_bits
        d = (h >> np.uint64(rem)).astype(np.int64)
        order = np.argsort(d, kind="stable")
        self._keys = torch.from_numpy((h[order] & np.uint64((1 << rem) - 1)).astype(np.int32))
        off = np.zeros((1 << l1_bits) + 1, dtype=np.int64)
        off[1:] = np.cumsum(np.bincount(d, minlength=1 << l1_bits))
        self._l1off = torch.from_numpy(off)
        self.segments = []
        self.bits = None

    def l1(self):
        return self._keys, self._l1off

    def add_segment(self, keys, l1off):
        self.segments.append((keys.numpy().astype(np.uint64), l1off.numpy()))

    def finish(self, abundance):
        rem = self.nbits - self.l1_bits
        full = []
        for keys, off in self.segments:
            digit = np.repeat(np.arange(off.size - 1, dtype=np.uint64), np.diff(off))
            full.append((digit << np.uint64(rem)) | keys)
        allh = np.concatenate(full) if full else np.zeros(0, dtype=np.uint64)
        vals, cnt = np.unique(allh, return_counts=True)
        self.bits = np.zeros(self.n_hashes, dtype=bool)
        self.bits[vals[np.minimum(cnt, 255) > abundance].astype(np.int64)] = True

    def extract(self, first_hash, n_hashes):
        idx = np.nonzero(self.bits[first_hash:first_hash + n_hashes])[0] + first_hash
        return torch.from_numpy(idx.astype(np.int64))

    def or_keys(self, keys):
        self.bits[keys.numpy()] = True


def _part_worker(rank, world, port, k, abundance, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from br_amd import dist as bd
    from oracle import oracle as O
    from tests.conftest import read_fasta
    reads = read_fasta(os.path.join(ROOT, "tests", "golden", "raw.fasta"))[1][:50] + [b"", b"ACGT"]
    lo, hi = bd.shard_range(len(reads), world, rank)
    eng = NumpyPartEngine(k, reads[lo:hi], l1_bits=3)
    bd.A2A_CHUNK_ELEMS = 4099   # force many all_to_all rounds (the cap is a constant shared by all ranks)
    bd.exchange_partitioned(eng, abundance, world, rank)
    ref = O.Solid.from_count(k, O.count_reads(k, reads), abundance)
    mine = np.packbits(eng.bits, bitorder="little").tobytes()
    q.put((rank, bytes([k]) + mine == ref.to_bytes()))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_key_exchange_matches_single(world):
    k, abundance = 9, 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 200) + world
    procs = [ctx.Process(target=_part_worker, args=(r, world, port, k, abundance, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(r, True) for r in range(world)]
