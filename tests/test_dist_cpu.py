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
