"""The shipped multi-GPU exchange (br_amd/csrc/brx_exchange.hip: brx_comm_init, brx_exchange_build_partitioned,
brx_exchange_reduce_counts) run with world 2 and 3 -- for real, not at world 1: `world` fresh processes share the one
card of the box, each drives the C ABI, and the library's ten librccl entry points are served by the test transport
tests/libfake_rccl.so (device -> host -> socket -> host -> device; selected with BRX_RCCL_PATH).  Everything above
those entry points -- owner bounds, per-peer send / recv counts, capped rounds, the clamped segment tables, the
finish of borrowed segments, the all-gather-v of the solid lists, the OR / index build from them, the clamp / widen /
narrow of the dense reduction, the status agreement -- is the product's code, checked bit for bit against the oracle.
Reference: there is no counterpart (one process, src/main.rs:30-33); src/lib.rs:72-139 is the loop being sharded."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

import br_amd
from br_amd import _lib, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
FAKE = os.path.join(HERE, "libfake_rccl.so")
WORKER = os.path.join(HERE, "abi_exchange_worker.py")


def _run_world(tmp_path, world, k, a, n_reads, strategy, extra_env=None):
    assert os.path.exists(FAKE), "tests/libfake_rccl.so not built: __graft_entry__.build()"
    prefix = str(tmp_path / "x")
    env = dict(os.environ)
    env["BRX_RCCL_PATH"] = FAKE
    env["FAKE_RCCL_DIR"] = str(tmp_path)
    env["FAKE_RCCL_STATS"] = prefix + ".traffic"
    env.update(extra_env or {})
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(k), str(a), str(n_reads), prefix, strategy],
                              env=env) for r in range(world)]
    try:
        codes = [p.wait(timeout=280) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0] * world
    out = []
    for r in range(world):
        with open("%s.rank%d.pkl" % (prefix, r), "rb") as f:
            out.append(pickle.load(f))
        with open("%s.traffic.rank%d" % (prefix, r)) as f:
            out[-1]["msgs"], out[-1]["bytes"] = (int(x) for x in f.read().split())
    return out


def _check_set(res, ref, k):
    if k <= 15:
        assert res["solid_bytes"] == ref.to_bytes()               # every rank holds the set of ALL reads
    else:
        assert res["members"] == [ref.get(x) for x in res["sample"]]
        assert any(res["members"]) and not all(res["members"])


@pytest.mark.parametrize("world,k", [(2, 13), (3, 13), (2, 15), (3, 19), (2, 19), (2, 21), (3, 21)])
def test_shipped_exchange_partitioned(tmp_path, raw_reads, world, k):
    """brx_exchange_build_partitioned at world 2 / 3 (uneven owner bounds: 512 first digits over 3 ranks), several
    capped rounds of the key all-to-all and of the list gather (BRX_A2A_CHUNK far below the per-peer message)."""
    a, n_reads = 2, 45
    res = _run_world(tmp_path, world, k, a, n_reads, "part", {"BRX_A2A_CHUNK": "30000"})
    reads = raw_reads[:n_reads]
    ref = O.Solid.from_count(k, O.count_reads(k, reads), a) if k <= 15 else O.Solid.sparse_from_count(k, reads, a)
    om = O.build_methods(ref, ["one", "graph"], 5, 7)
    expect = [O.correct_record(om, r, False) for r in reads]
    got = []
    n_kmers = sum(max(0, len(r) - k + 1) for r in reads)
    for r, x in enumerate(res):
        _check_set(x, ref, k)
        assert x["corrected_0"] == x["corrected_1"]
        st = x["stats_1"]
        assert st["solid_job"] == x["popcount"] and 0 < st["solid_here"] < st["solid_job"]
        assert st["key_bytes_sent"] > 0 and st["key_bytes_received"] > 0
        assert st["largest_message_keys"] > 30000 and x["msgs"] > 4 * (world - 1)      # really several rounds
        if k >= 15:
            assert x["index"]["valid"]
        got += x["corrected_0"]
    assert sum(x["stats_1"]["keys_counted_here"] for x in res) == n_kmers              # every key reached ONE owner
    assert got == expect                                                               # shards concatenate in input order


def test_shipped_exchange_with_an_empty_shard(tmp_path, raw_reads):
    """2 reads over 3 ranks: rank 0 counts nothing and must still join every collective (it used to leave its peers
    waiting in the first all-gather), own a digit range, and end with the whole set."""
    k, a = 15, 0
    res = _run_world(tmp_path, 3, k, a, 2, "part")
    assert [x["n_mine"] for x in res] == [0, 1, 1]
    ref = O.Solid.from_count(k, O.count_reads(k, raw_reads[:2]), a)
    for x in res:
        _check_set(x, ref, k)
        assert x["popcount"] == x["stats_1"]["solid_job"]
    assert res[0]["stats_1"]["key_bytes_sent"] == 0 and res[0]["stats_1"]["keys_counted_here"] > 0


@pytest.mark.parametrize("world,a", [(2, 2), (3, 2), (2, 200), (3, 100)])
def test_shipped_exchange_dense_reduce(tmp_path, raw_reads, world, a):
    """brx_exchange_reduce_counts, north_star's form (all-reduce of the u8 count vector, then every rank thresholds):
    exact u8 sums while world * (a + 1) <= 255 -- against a transport whose u8 SUM wraps like RCCL's -- and int32
    slices beyond (a = 200, and 3 x 101 = 303).  k = 11 on the whole fixture: five k-mers are counted 256 ... 1730 times."""
    k = 11
    res = _run_world(tmp_path, world, k, a, len(raw_reads), "dense")
    ref = O.Solid.from_count(k, O.count_reads(k, raw_reads), a)
    for x in res:
        assert x["solid_bytes"] == ref.to_bytes()
    assert res[0]["popcount"] > 0


def test_dense_reduce_wide_path_at_world_one(monkeypatch):
    """the widen -> int32 all-reduce -> saturating narrow path through REAL librccl (world 1 is all a one-GPU box can
    give it), forced with BRX_EXCHANGE_FORCE_WIDE: the table must come back unchanged"""
    from br_amd import dist as D
    k, a = 13, 3
    cfg = synth.config(genome_len=30_000, read_len=2_000)
    g = synth.genome_host(cfg)
    bases, offs = synth.reads_host(cfg, g, 0, 300)
    plain = br_amd.Counter(k, 0, _lib.COUNT_DENSE)
    plain.add_batch(bases, offs)
    ref = plain.finish(a)
    monkeypatch.setenv("BRX_EXCHANGE_FORCE_WIDE", "1")
    ex = D.AbiExchange(1, 0, 0)
    cnt = br_amd.Counter(k, 0, _lib.COUNT_DENSE)
    cnt.add_batch(bases, offs)
    ex.reduce_counts(cnt, a, None)
    assert cnt.finish(a).to_solid_bytes() == ref.to_solid_bytes()
    ex.close()
