"""BASELINE.json configs[2] and configs[4] at their real sizes on one GPU, through size-independent properties plus an
oracle sample (the oracle cannot run on 10 Gbp):

  configs[2]  1e6 synthetic 10 kb reads (10 Gbp, 1e10 k-mers > 2^32, dense level-2 buckets -> final_count_kernel), k=19,
              correct::greedy: both build strategies give the same 16 GiB bit vector, all-solid reads come back unchanged,
              splitting the batch does not change a byte, 64 sampled reads equal the oracle run against the exported set.
  configs[4]  one GPU's share of the 8-GPU job: 625 000 reads (6.25 Gbp), k=21 (sparse set, four radix levels),
              correct::graph then correct::gap_size: batch-split invariance and 64 sampled reads against the sparse oracle.
"""
import numpy as np
import pytest

import br_amd
from br_amd import _lib, synth
from br_amd import dist as bd
from oracle import oracle as O

pytestmark = pytest.mark.gpu

READ_LEN, ABUNDANCE = 10_000, 3


def _make_reads(n_reads, coverage=50):
    import torch
    cfg = synth.config(genome_len=n_reads * READ_LEN // coverage, read_len=READ_LEN)
    stream = torch.cuda.current_stream().cuda_stream
    dg = torch.empty(cfg.genome_len, dtype=torch.uint8, device="cuda")
    synth.genome_device(cfg, 0, dg.data_ptr(), stream)
    cap = int(n_reads * READ_LEN * 1.03) + (1 << 20)
    db = torch.empty(cap, dtype=torch.uint8, device="cuda")
    do = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")
    total = synth.reads_device(cfg, 0, dg.data_ptr(), 0, n_reads, db.data_ptr(), cap, do.data_ptr(), stream)
    del dg
    return cfg, db, do, total, stream


def _correct(chain, db, do, n, nb, stream, first=0):
    import torch
    off = do[first:first + n + 1].contiguous()
    out = torch.empty(int(nb * 1.06) + (1 << 20), dtype=torch.uint8, device="cuda")
    oo = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    tot = chain.correct_batch_device(db.data_ptr(), off.data_ptr(), n, nb, out.data_ptr(), out.numel(), oo.data_ptr(), stream)
    return out, oo, tot


def test_config2_10gbp_greedy():
    import torch
    K, N = 19, 1_000_000
    cfg, db, do, total, stream = _make_reads(N)
    assert total > (1 << 33)                                     # 1e10 k-mers: key offsets far beyond 2^32
    cnt = br_amd.Counter(K, 0, _lib.COUNT_SORTED)
    cnt.add_batch_device(db.data_ptr(), do.data_ptr(), N, total, stream)
    solid = cnt.finish(ABUNDANCE, stream)
    del cnt
    n_solid = solid.popcount()
    assert 0.95 * cfg.genome_len < n_solid < 1.25 * cfg.genome_len

    # the reference's own structure at this size: 128 GiB of u8 counters, thresholded into 16 GiB of bits
    cnt = br_amd.Counter(K, 0, _lib.COUNT_DENSE)
    cnt.add_batch_device(db.data_ptr(), do.data_ptr(), N, total, stream)
    dense = cnt.finish(ABUNDANCE, stream)
    del cnt
    torch.cuda.synchronize()
    assert dense.popcount() == n_solid
    pa, na = solid.device_bits()
    pb, nb_ = dense.device_bits()
    assert na == nb_ and torch.equal(bd.device_view(pa, na).view(torch.int64), bd.device_view(pb, nb_).view(torch.int64))
    del dense

    off_h = do.cpu().numpy()
    chain = br_amd.Chain(solid, [("greedy", 5, 7)], two_side=False)
    out, oo, tot = _correct(chain, db, do, N, total, stream)
    st = chain.last_stats()
    assert st["fixes"] > 5_000_000 and st["overflow_retries"] == 0
    oo_h = oo.cpu().numpy()
    assert int(oo_h[-1]) == tot

    # batch-split invariance: the last 300 000 reads on their own give the same bytes
    a = 700_000
    nb2 = int(off_h[N] - off_h[a])
    sub_off = (do[a:] - do[a]).contiguous()
    out2 = torch.empty(int(nb2 * 1.06) + (1 << 20), dtype=torch.uint8, device="cuda")
    oo2 = torch.empty(N - a + 1, dtype=torch.int64, device="cuda")
    tot2 = chain.correct_batch_device(db.data_ptr() + int(off_h[a]), sub_off.data_ptr(), N - a, nb2, out2.data_ptr(), out2.numel(),
                                      oo2.data_ptr(), stream)
    assert tot2 == tot - int(oo_h[a])
    assert torch.equal(out2[:tot2], out[int(oo_h[a]):tot])
    del out2, oo2

    # 64 reads against the oracle with the very set the GPU built (16 GiB exported from HBM)
    bits = solid.export_bits()
    om = O.build_methods(O.Solid.wrap(K, bits), ["greedy"], 5, 7)
    rng = np.random.default_rng(7)
    sample = sorted(set(rng.integers(0, N, size=62).tolist()) | {0, N - 1})
    changed = 0
    for r in sample:
        src = db[int(off_h[r]):int(off_h[r + 1])].cpu().numpy().tobytes()
        got = out[int(oo_h[r]):int(oo_h[r + 1])].cpu().numpy().tobytes()
        assert got == O.correct_record(om, src, False), r
        changed += got != src
    assert changed > len(sample) // 2
    del bits, om

    # all-solid identity: with abundance 0 nothing can trigger (200 000 reads)
    n0 = 200_000
    nb0 = int(off_h[n0])
    cnt = br_amd.Counter(K, 0, _lib.COUNT_SORTED)
    cnt.add_batch_device(db.data_ptr(), do[:n0 + 1].contiguous().data_ptr(), n0, nb0, stream)
    every = cnt.finish(0, stream)
    del cnt
    ch0 = br_amd.Chain(every, [("greedy", 5, 7)], two_side=True)
    out0, oo0, tot0 = _correct(ch0, db, do, n0, nb0, stream)
    assert tot0 == nb0 and torch.equal(oo0, do[:n0 + 1]) and torch.equal(out0[:nb0], db[:nb0])
    assert ch0.last_stats()["triggers"] == 0


def test_config4_share_k21_graph_gap_size():
    import torch
    K, N = 21, 625_000
    cfg, db, do, total, stream = _make_reads(N)
    cnt = br_amd.Counter(K, 0, _lib.COUNT_SORTED)
    cnt.add_batch_device(db.data_ptr(), do.data_ptr(), N, total, stream)
    solid = cnt.finish(ABUNDANCE, stream)
    del cnt
    assert solid.is_sparse() and solid.bits_state() == 2
    n_solid = solid.popcount()
    assert 0.95 * cfg.genome_len < n_solid < 1.25 * cfg.genome_len
    kl = solid.keylist_device(stream)                 # before anything rebuilds the set's index from another list
    assert kl is not None and kl[1] == n_solid
    keylist_host = bd.device_view(kl[0], kl[1] * 8).view(torch.int64).cpu().numpy().view(np.uint64).copy()

    methods = ["graph", "gap_size"]
    chain = br_amd.Chain(solid, [(m, 5, 7) for m in methods], two_side=False)
    out, oo, tot = _correct(chain, db, do, N, total, stream)
    st = chain.last_stats()
    assert st["fixes"] > 10_000_000
    off_h, oo_h = do.cpu().numpy(), oo.cpu().numpy()
    assert int(oo_h[-1]) == tot

    # batch-split invariance on the first 200 000 reads (a second chain: its own workspace and walk lists)
    n1 = 200_000
    nb1 = int(off_h[n1])
    chain2 = br_amd.Chain(solid, [(m, 5, 7) for m in methods], two_side=False)
    out1, oo1, tot1 = _correct(chain2, db, do, n1, nb1, stream)
    assert tot1 == int(oo_h[n1]) and torch.equal(out1[:tot1], out[:tot1])
    del out1, oo1, chain2

    # 64 reads against the sparse oracle
    rng = np.random.default_rng(9)
    sample = sorted(set(rng.integers(0, N, size=62).tolist()) | {0, N - 1})
    reads = [db[int(off_h[r]):int(off_h[r + 1])].cpu().numpy().tobytes() for r in sample]

    # The oracle needs a C-side set with the SAME members: a walk (graph.rs:61-82) may follow the genome for thousands
    # of k-mers before it dies, so nothing short of the whole set will do -- the key list the partitioned finish left
    # with the set (every solid hash, ~1 GB), sorted on the host, is the sparse oracle's array.
    kl = keylist_host
    assert kl.size == n_solid
    members = np.sort(kl)
    assert np.all(members[1:] != members[:-1])      # a set: every hash once
    osolid = O.Solid(K, _h=O.lib().bro_solid_new_sparse(K, members.ctypes.data, members.size))
    om = O.build_methods(osolid, methods, 5, 7)
    changed = 0
    for r, src in zip(sample, reads):
        got = out[int(oo_h[r]):int(oo_h[r + 1])].cpu().numpy().tobytes()
        assert got == O.correct_record(om, src, False), r
        changed += got != src
    assert changed > len(sample) // 2


def test_config3_share_k19_one():
    """BASELINE configs[3]'s per-GPU share: 625 000 reads (6.25 Gbp of a 125 Mbp genome), k = 19, set build THROUGH THE
    SHIPPED EXCHANGE (brx_exchange_build_partitioned, real librccl, world 1) + correct::one forward and reverse.  This
    size takes paths configs[1] never reaches: the 9-bit second digit, hash-count passes over key ranges, ~130 M solid
    k-mers -> 16-mer minimizers and more than 2^25 index lines under `one_kernel`."""
    import torch
    K, N = 19, 625_000
    cfg, db, do, total, stream = _make_reads(N)
    cnt = br_amd.Counter(K, 0, _lib.COUNT_SORTED)
    cnt.add_batch_device(db.data_ptr(), do.data_ptr(), N, total, stream)
    solid = br_amd.Pcon.new(K, 0)
    ex = bd.AbiExchange(1, 0, 0)
    ex.build_partitioned(cnt, solid, ABUNDANCE, stream)
    st = ex.last_stats()
    ex.close()
    n_solid = solid.popcount()
    assert st["solid_job"] == n_solid and st["keys_counted_here"] > 6_000_000_000
    assert 0.95 * cfg.genome_len < n_solid < 1.25 * cfg.genome_len
    info = solid.index_info()
    assert info["valid"] and info["m"] == 16 and info["log2_lines"] > 25

    off_h = do.cpu().numpy()
    chain = br_amd.Chain(solid, [("one", 5, 7)], two_side=False)
    out, oo, tot = _correct(chain, db, do, N, total, stream)
    cst = chain.last_stats()
    assert cst["fixes"] > 20_000_000
    oo_h = oo.cpu().numpy()
    assert int(oo_h[-1]) == tot

    # the plain finish of the same counter contents, and the reference's own structure (128 GiB of u8 counters): same set
    cnt.add_batch_device(db.data_ptr(), do.data_ptr(), N, total, stream)
    plain = cnt.finish(ABUNDANCE, stream)
    del cnt
    assert plain.popcount() == n_solid
    pa, na = solid.device_bits()
    pb, nb_ = plain.device_bits()
    assert na == nb_ and torch.equal(bd.device_view(pa, na).view(torch.int64), bd.device_view(pb, nb_).view(torch.int64))
    del plain
    cnt = br_amd.Counter(K, 0, _lib.COUNT_DENSE)
    cnt.add_batch_device(db.data_ptr(), do.data_ptr(), N, total, stream)
    dense = cnt.finish(ABUNDANCE, stream)
    del cnt
    torch.cuda.synchronize()
    pb, nb_ = dense.device_bits()
    assert torch.equal(bd.device_view(pa, na).view(torch.int64), bd.device_view(pb, nb_).view(torch.int64))
    del dense

    # batch-split invariance: the last 225 000 reads on their own give the same bytes
    a = 400_000
    nb2 = int(off_h[N] - off_h[a])
    sub_off = (do[a:] - do[a]).contiguous()
    out2 = torch.empty(int(nb2 * 1.06) + (1 << 20), dtype=torch.uint8, device="cuda")
    oo2 = torch.empty(N - a + 1, dtype=torch.int64, device="cuda")
    tot2 = chain.correct_batch_device(db.data_ptr() + int(off_h[a]), sub_off.data_ptr(), N - a, nb2, out2.data_ptr(), out2.numel(),
                                      oo2.data_ptr(), stream)
    assert tot2 == tot - int(oo_h[a])
    assert torch.equal(out2[:tot2], out[int(oo_h[a]):tot])
    del out2, oo2

    # 64 reads against the oracle with the very set the GPU built (16 GiB exported from HBM)
    bits = solid.export_bits()
    om = O.build_methods(O.Solid.wrap(K, bits), ["one"], 5, 7)
    rng = np.random.default_rng(11)
    sample = sorted(set(rng.integers(0, N, size=62).tolist()) | {0, N - 1})
    changed = 0
    for r in sample:
        src = db[int(off_h[r]):int(off_h[r + 1])].cpu().numpy().tobytes()
        got = out[int(oo_h[r]):int(oo_h[r + 1])].cpu().numpy().tobytes()
        assert got == O.correct_record(om, src, False), r
        changed += got != src
    assert changed > len(sample) // 2
