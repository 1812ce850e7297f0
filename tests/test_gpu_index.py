"""The probe index (include/brx.h, br_amd/csrc/brx_index.hpp) only changes WHERE `KmerSet::get` reads:
every answer and every corrected byte must be what the bitset path / the oracle give, also when lines
overflow at build time and probes fall back to the bitset (forced here with tiny tables)."""
import numpy as np
import pytest

import br_amd
from br_amd import _lib, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

METHODS = ["one", "two", "graph", "greedy", "gap_size"]


def _solid_kmers(solid: O.Solid, k: int) -> np.ndarray:
    """forward k-mers (canonical representatives) of every set bit."""
    bits = np.unpackbits(np.frombuffer(solid.to_bytes()[1:], dtype=np.uint8), bitorder="little")
    h = np.flatnonzero(bits).astype(np.uint64)
    par = np.array([bin(int(x)).count("1") & 1 for x in h], dtype=np.uint64)
    return (h << np.uint64(1)) | par


def _queries(solid_kmers: np.ndarray, k: int, rng) -> np.ndarray:
    mask = np.uint64((1 << (2 * k)) - 1)
    subs = []
    for pos in (0, k // 2, k - 1):  # one-substitution neighbours: mostly absent, same minimizer lines
        for d in (1, 2, 3):
            subs.append(solid_kmers ^ (np.uint64(d) << np.uint64(2 * pos)))
    rnd = rng.integers(0, 1 << (2 * k), size=20000, dtype=np.uint64)
    rc_like = (~solid_kmers) & mask
    return np.concatenate([solid_kmers] + subs + [rnd, rc_like]) & mask


@pytest.mark.parametrize("k,log2_lines", [(11, 0), (11, 5), (15, 0), (15, 8), (19, 0), (5, 0), (7, 4)])
def test_indexed_get_equals_bitset_get(raw_reads, k, log2_lines):
    rng = np.random.default_rng(k * 100 + log2_lines)
    reads = raw_reads[:150] if k >= 15 else raw_reads
    gs = br_amd.Pcon.from_count(reads, k, 1 if k >= 15 else 2)
    if k <= 15:
        ref = O.Solid.from_count(k, O.count_reads(k, reads), 1 if k >= 15 else 2)
        assert gs.to_solid_bytes() == ref.to_bytes()
        sk = _solid_kmers(ref, k)
    else:  # 2^37 bits: take the solid k-mers from the reads themselves
        sk = np.array(sorted({O.seq2bit(r[i:i + k]) for r in reads[:20] for i in range(0, len(r) - k + 1, 7)
                              if set(r[i:i + k]) <= set(b"ACGT")}), dtype=np.uint64)
    info = gs.index_build(0, log2_lines)
    assert info["valid"] and info["keys"] == gs.popcount()
    if log2_lines and k > 5:
        assert info["overflow_keys"] > 0  # the tiny table really exercises the fallback
    q = _queries(sk, k, rng)
    want = gs.get_many(q)
    got, n_fallback = gs.get_batch_indexed(q)
    assert np.array_equal(got, want)
    if info["overflow_keys"] == 0:
        assert n_fallback == 0
    if log2_lines and k > 5:
        assert n_fallback > 0
    # every key the index says it holds + the overflowed ones are all found
    assert got[:len(sk)].sum() == want[:len(sk)].sum()


def test_index_dropped_by_mutation(solid_fixture_bytes):
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    assert gs.index_build()["valid"]
    km = O.seq2bit(b"ACGTTGCAAGT")
    before = gs.get(km)
    gs.set(km, not before)
    assert not gs.index_info()["valid"]
    assert gs.index_build()["valid"]
    got, _ = gs.get_batch_indexed([km])
    assert bool(got[0]) == (not before)


@pytest.mark.parametrize("log2_lines", ["0", "6"])
@pytest.mark.parametrize("method", METHODS)
def test_correctors_through_the_index_fixture(raw_reads, solid_fixture_bytes, method, log2_lines, monkeypatch):
    """the reference's integration data (k=11) with every probe going through the index; "6" = 64 lines
    for 25k solid k-mers, i.e. nearly every probe takes the overflow -> group re-run path."""
    monkeypatch.setenv("BRX_INDEX_MIN_K", "5")
    monkeypatch.setenv("BRX_INDEX_LOG_LINES", log2_lines)
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    om = O.build_methods(os_, [method], 5, 7)
    reads = raw_reads if method != "greedy" else raw_reads[:120]
    chain = br_amd.Chain(gs, [(method, 5, 7)], two_side=False)
    got = chain.correct_reads(reads)
    info = gs.index_info()
    assert info["valid"] and (info["log2_lines"] == 6) == (log2_lines == "6")
    for r, g in zip(reads, got):
        assert g == O.correct_record(om, r, False)


@pytest.mark.parametrize("group", ["8", "16", "32", "64"])
def test_one_through_the_index_every_group_width(raw_reads, solid_fixture_bytes, group, monkeypatch):
    monkeypatch.setenv("BRX_INDEX_MIN_K", "5")
    monkeypatch.setenv("BRX_INDEX_LOG_LINES", "9")
    monkeypatch.setenv("BRX_GROUP", group)
    monkeypatch.setenv("BRX_GROUP_REV", group)
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    om = O.build_methods(os_, ["one"], 5, 7)
    got = br_amd.Chain(gs, [("one", 5, 7)], two_side=False).correct_reads(raw_reads)
    assert gs.index_info()["overflow_keys"] > 0
    for r, g in zip(raw_reads, got):
        assert g == O.correct_record(om, r, False)


@pytest.mark.parametrize("method", METHODS)
def test_synthetic_k15_through_the_index(method):
    """k=15 is indexed by default: count -> threshold -> correct, compared with the oracle."""
    k, n_reads, read_len = 15, 200, 1500
    cfg = synth.config(genome_len=n_reads * read_len // 30, read_len=read_len, sub=0.01, ins=0.006, dele=0.006)
    g = synth.genome_host(cfg)
    hb, ho = synth.reads_host(cfg, g, 0, n_reads)
    reads = [hb[int(ho[i]):int(ho[i + 1])].tobytes() for i in range(n_reads)]
    gs = br_amd.Pcon.from_count(reads, k, 2)
    ref = O.Solid.from_count(k, O.count_reads(k, reads), 2)
    om = O.build_methods(ref, [method], 5, 7)
    chain = br_amd.Chain(gs, [(method, 5, 7)], two_side=False)
    out, oo = chain.correct_batch(hb, ho)
    assert gs.index_info()["valid"]
    exp, exp_o = O.correct_batch(om, hb, ho, False)
    assert np.array_equal(oo, exp_o)
    assert np.array_equal(out, exp)
    assert om[0].stats()["fixes"] > 50


def test_index_off_switch(raw_reads, monkeypatch):
    monkeypatch.setenv("BRX_INDEX", "0")
    gs = br_amd.Pcon.from_count(raw_reads[:50], 15, 1)
    br_amd.Chain(gs, [("one", 5, 7)], two_side=True).correct_reads(raw_reads[:5])
    assert not gs.index_info()["valid"]


def test_lazy_bit_vector(raw_reads, monkeypatch):
    """a partitioned finish lists the solid hashes and leaves the bit vector unwritten (bits_state 1): membership,
    popcount and One answer from the chained index -- also for keys that overflowed their line -- and the vector
    appears, identical to the oracle's, when something asks for it"""
    k, a = 15, 1
    reads = raw_reads[:150]
    ref = O.Solid.from_count(k, O.count_reads(k, reads), a)
    cnt = br_amd.Counter(k, 0)
    cnt.add_reads(reads)
    gs = cnt.finish(a)
    assert gs.bits_state() == 1
    assert gs.popcount() == ref.popcount()
    monkeypatch.setenv("BRX_INDEX_LOG_LINES", "12")  # far too few lines: the build takes the smallest table that fits
    sk = _solid_kmers(ref, k)
    q = _queries(sk, k, np.random.default_rng(7))
    want = np.array([ref.get(int(x)) for x in q])
    assert np.array_equal(gs.get_many(q), want)  # builds the chained index from the key list
    info = gs.index_info()
    assert info["valid"] and info["overflow_keys"] > info["keys"] // 10
    assert gs.bits_state() == 1
    om = O.build_methods(ref, ["one"], 5, 7)
    got = br_amd.Chain(gs, [("one", 5, 7)], two_side=False).correct_reads(reads)
    assert gs.bits_state() == 1  # One never needed the vector
    for r, g in zip(reads, got):
        assert g == O.correct_record(om, r, False)
    om = O.build_methods(ref, ["graph"], 5, 7)
    got = br_amd.Chain(gs, [("graph", 5, 7)], two_side=False).correct_reads(reads[:40])
    assert gs.bits_state() == 0  # a walking method materialised it
    for r, g in zip(reads[:40], got):
        assert g == O.correct_record(om, r, False)
    assert gs.to_solid_bytes() == ref.to_bytes()


def test_fingerprint_is_a_property_of_the_set(monkeypatch):
    """brx_set_fingerprint (members, sum of hashes, sum of squares, mod 2^64) -- what the ranks of a multi-GPU job compare
    after the exchange -- must not depend on what holds the set: the reference's bit vector (dense count), the solid-hash
    list of a partitioned finish (bit vector lazy), the chained table of a sparse set, a table built from a handed-over
    list (the exchange's last step: the set's own list is void then).  All equal the oracle's members."""
    import torch
    from br_amd import dist as bd
    k, a = 15, 2
    cfg = synth.config(genome_len=40_000, read_len=2_000)
    g = synth.genome_host(cfg)
    bases, offs = synth.reads_host(cfg, g, 0, 400)
    reads = [bases[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(400)]
    ref = O.Solid.from_count(k, O.count_reads(k, reads), a)
    members = np.flatnonzero(np.unpackbits(ref.bits().view(np.uint8), bitorder="little")).astype(np.uint64)
    with np.errstate(over="ignore"):
        want = (int(members.size), int(members.sum(dtype=np.uint64)), int((members * members).sum(dtype=np.uint64)))
    assert want[0] == ref.popcount()

    def counted(strategy, sparse=False):
        if sparse:
            monkeypatch.setenv("BRX_FORCE_SPARSE", "1")
        try:
            c = br_amd.Counter(k, 0, strategy)
            c.add_reads(reads)
            return c.finish(a)
        finally:
            monkeypatch.delenv("BRX_FORCE_SPARSE", raising=False)

    dense = counted(_lib.COUNT_DENSE)
    assert dense.popcount() == want[0]
    assert dense.fingerprint() == want, ("bit vector", dense.fingerprint(), want)
    lazy = counted(_lib.COUNT_SORTED)
    assert lazy.bits_state() == 1                                                  # solid-hash list, bit vector unwritten
    assert lazy.fingerprint() == want, ("list", lazy.fingerprint(), want)
    sparse = counted(_lib.COUNT_SORTED, sparse=True)
    assert sparse.is_sparse()
    assert sparse.fingerprint() == want, ("sparse list", sparse.fingerprint(), want)
    kl = sparse.keylist_device()
    keys = bd.device_view(kl[0], kl[1] * 8).view(torch.int64).clone()
    sp2 = counted(_lib.COUNT_SORTED, sparse=True)
    sp2.index_build_from_keys_device(keys.data_ptr(), keys.numel())               # the exchange's last step: a handed-over list
    assert sp2.keylist_device() is None                                            # ... the chained table alone holds the set
    assert sp2.fingerprint() == want, ("table", sp2.fingerprint(), want)
