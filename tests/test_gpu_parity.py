"""Parity of the HIP path (through the C ABI) against the CPU oracle, on a real MI355X.

Bit-exact everywhere: this is integer/byte work.  Sizes are chosen so the oracle finishes in
seconds; full-size cases use size-independent properties (see test_gpu_scale.py).
"""
import gzip
import os

import numpy as np
import pytest

import br_amd
from br_amd import _lib, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

IMPLEMENTED = {"one", "two", "graph", "greedy", "gap_size"}  # methods with a GPU kernel; the others must fail loudly


def _oracle_set(v):
    s = O.Solid(v["k"])
    for q in v["set_seqs"]:
        s.set_seq(q.encode())
    for q in v["set_kmers"]:
        s.set(O.seq2bit(q.encode()))
    return s


def _gpu_set_from_oracle(s):
    return br_amd.Pcon.from_pcon_solid(s.to_bytes())


def _corrector(gs, v):
    m = v["method"]
    if m == "one":
        return br_amd.One(gs, v["confirm"])
    if m == "two":
        return br_amd.Two(gs, v["confirm"])
    if m == "graph":
        return br_amd.Graph(gs)
    if m == "greedy":
        return br_amd.Greedy(gs, v["max_search"], v["confirm"])
    return br_amd.GapSize(gs, v["confirm"])


# ---------------------------------------------------------------- set ---------------------------
def test_device_present():
    assert _lib.device_count() >= 1


def test_solid_roundtrip(solid_fixture_bytes):
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    assert gs.k() == 11
    assert gs.to_solid_bytes() == solid_fixture_bytes
    assert gs.popcount() == 123072
    # gzip'ed stream, as the reference's fixture is stored
    gz = br_amd.Pcon.from_pcon_solid(gzip.compress(solid_fixture_bytes))
    assert gz.to_solid_bytes() == solid_fixture_bytes
    with pytest.raises(_lib.BrxError):
        br_amd.Pcon.from_pcon_solid(solid_fixture_bytes[:-1])


def test_get_matches_oracle(solid_fixture_bytes):
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    rng = np.random.default_rng(1)
    kmers = rng.integers(0, 4 ** 11, size=20000, dtype=np.uint64)
    got = gs.get_many(kmers)
    exp = np.array([os_.get(int(x)) for x in kmers])
    assert np.array_equal(got, exp)
    for x in kmers[:50]:
        assert gs.get(int(x)) == os_.get(int(x))
    assert not br_amd.Pcon.new(11).get(0)           # src/set/pcon.rs:239-241 (absence)


def test_pcon_set_vectors(unit_vectors):
    d = unit_vectors["set"]["pcon"]
    k, seq = d["k"], d["seq"].encode()
    gs = br_amd.Pcon.from_fasta([seq], k)
    fw = [O.seq2bit(seq[i:i + k]) for i in range(len(seq) - k + 1)]
    assert gs.get_many(fw).all()                                     # forward
    assert gs.get_many([O.canonical(x, k) for x in fw]).all()        # canonical
    assert gs.get_many([O.revcomp(x, k) for x in fw]).all()
    assert not gs.get(d["absent_kmer"])
    assert gs.k() == k
    ref = O.Solid(k)
    ref.set_seq(seq)
    assert gs.to_solid_bytes() == ref.to_bytes()


def test_set_and_found_alt_kmer(unit_vectors):
    d = unit_vectors["set"]["found_alt_kmer"]
    gs = br_amd.Pcon.new(d["k"])
    for q in d["set_kmers"]:
        gs.set(br_amd.seq2bit(q.encode()))
    base = br_amd.seq2bit(d["query"].encode()) >> 2
    alts = [a for a in range(4) if gs.get((base << 2) | a)]
    assert alts == d["alt_nucs"]
    gs.set(br_amd.seq2bit(d["set_kmers"][0].encode()), False)
    assert not gs.get(br_amd.seq2bit(d["set_kmers"][0].encode()))


STRATEGIES = [_lib.COUNT_DENSE, _lib.COUNT_SORTED]


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_set_build_kat(raw_reads, solid_fixture_bytes, strategy):
    """raw.fasta, k=11, count > 2  ==  tests/data/raw.k11.a2.solid, bit for bit."""
    gs = br_amd.Pcon.from_count(raw_reads, 11, 2, strategy=strategy)
    assert gs.to_solid_bytes() == solid_fixture_bytes
    # batching must not matter (count_fasta(reader, 8192))
    gs2 = br_amd.Pcon.from_count(raw_reads, 11, 2, batch=7, strategy=strategy)
    assert gs2.to_solid_bytes() == solid_fixture_bytes


@pytest.mark.parametrize("strategy", STRATEGIES)
@pytest.mark.parametrize("k,abundance", [(5, 0), (7, 1), (9, 1), (13, 3), (15, 2), (17, 0)])
def test_set_build_vs_oracle(raw_reads, k, abundance, strategy):
    if strategy == _lib.COUNT_SORTED and k < 7:
        with pytest.raises(_lib.BrxError):
            br_amd.Counter(k, 0, strategy)
        return
    reads = raw_reads[:40] + [b"", b"ACG", b"N" * 40, b"acgtacgtacgtacgtacgtacgt"]
    gs = br_amd.Pcon.from_count(reads, k, abundance, strategy=strategy)
    ref = O.Solid.from_count(k, O.count_reads(k, reads), abundance)
    assert gs.to_solid_bytes() == ref.to_bytes()


@pytest.mark.parametrize("strategy,k", [(_lib.COUNT_DENSE, 5), (_lib.COUNT_DENSE, 7), (_lib.COUNT_SORTED, 7)])
def test_counter_saturates_at_255(strategy, k):
    reads = [b"A" * (k + 299)] * 3 + [b"ACGTACGTAC"]      # AAAAA.. seen 900 times
    for a in (0, 200, 254):
        gs = br_amd.Pcon.from_count(reads, k, a, strategy=strategy)
        ref = O.Solid.from_count(k, O.count_reads(k, reads), a)
        assert gs.to_solid_bytes() == ref.to_bytes()
        assert gs.get(0)
    assert not br_amd.Pcon.from_count(reads, k, 255, strategy=strategy).get(0)   # nothing exceeds 255


def test_partitioned_huge_bucket():
    """> 65 000 occurrences of k-mers inside one fine bucket: the chunk-and-clamp path of the LDS
    counters (u16 halves must never carry) against the saturating oracle."""
    k = 9
    reads = [b"A" * 2000] * 80 + [b"AAAAAAAAC" * 30] * 40 + [b"ACGTTGCAAGGCTTACCGATAGGCAT" * 20]
    for a in (0, 3, 254):
        gs = br_amd.Pcon.from_count(reads, k, a, strategy=_lib.COUNT_SORTED)
        ref = O.Solid.from_count(k, O.count_reads(k, reads), a)
        assert gs.to_solid_bytes() == ref.to_bytes()


@pytest.mark.parametrize("blocks", ["1", "3", "7"])
@pytest.mark.parametrize("k", [13, 15, 19])
def test_partitioned_level1_tile_ranges(raw_reads, monkeypatch, k, blocks):
    """Level 1 of the partitioned build gives every block a contiguous range of 4096-position tiles and carries state from
    tile to tile (the read index, the keys behind each digit's last 32-byte sector).  Small inputs are one tile per block:
    BRX_L1_GRID shrinks the grid so that ranges of many tiles, their first-sector phantoms and their final flush are run
    against the oracle -- with short reads (many boundaries per tile), empty reads, and low-complexity reads that put
    whole tiles into one digit.  k = 19: against the sparse oracle (the dense one would need 128 GiB of counters)."""
    monkeypatch.setenv("BRX_L1_GRID", blocks)
    rng = np.random.default_rng(100 + k)
    short = [bytes(rng.choice(list(b"ACGT"), size=int(n)).astype(np.uint8)) for n in rng.integers(0, 60, size=600)]
    reads = raw_reads[:30] + [b"A" * 9000, b"", b"ACACACACAC" * 700, b"T" * 5000] + short + raw_reads[30:45] + [b"G" * 4100]
    assert sum(len(r) for r in reads) > 20 * 4096
    for a in (0, 2):
        gs = br_amd.Pcon.from_count(reads, k, a, strategy=_lib.COUNT_SORTED)
        if k <= 15:
            ref = O.Solid.from_count(k, O.count_reads(k, reads), a)
            assert gs.to_solid_bytes() == ref.to_bytes()
        else:
            ref = O.Solid.sparse_from_count(k, reads, a)
            assert gs.popcount() == ref.popcount() > 1000
            kmers = _kmers_of(reads[:40:3], k)
            got = gs.get_many(kmers)
            assert np.array_equal(got, np.array([ref.get(int(x)) for x in kmers]))
            assert got.any() and (a == 0 or not got.all())


def _kmers_of(reads, k):
    out = []
    for r in reads:
        for i in range(0, len(r) - k + 1, 5):
            w = r[i:i + k]
            if set(w) <= set(b"ACGT"):
                out.append(O.seq2bit(w))
    return np.array(out, dtype=np.uint64)


def test_presence_build_vs_oracle(raw_reads):
    k = 13
    reads = raw_reads[:25]
    gs = br_amd.Pcon.from_fasta(reads, k)
    ref = O.Solid(k)
    for r in reads:
        ref.set_seq(r)
    assert gs.to_solid_bytes() == ref.to_bytes()


def test_even_k_rejected_for_build():
    with pytest.raises(_lib.BrxError):
        br_amd.Counter(12)


# ---------------------------------------------------------------- correction ---------------------
def test_unit_vectors_on_gpu(unit_vectors):
    """the reference's own corrector tests, run through the HIP path."""
    ran = 0
    for v in unit_vectors["vectors"]:
        if v["ignored"]:
            continue
        gs = _gpu_set_from_oracle(_oracle_set(v))
        if v["method"] not in IMPLEMENTED:
            with pytest.raises(_lib.BrxError) as e:
                _corrector(gs, v).correct(v["cases"][0][0].encode())
            assert e.value.status == _lib.BRX_ERR_UNSUPPORTED
            continue
        c = _corrector(gs, v)
        for a, b in v["cases"]:
            assert c.correct(a.encode()).decode() == b, v["name"]
            ran += 1
    assert ran >= 80


@pytest.mark.parametrize("group", ["8", "16", "32", "64"])
def test_one_raw_fasta_fixture_set(raw_reads, solid_fixture_bytes, group, monkeypatch):
    """the reference's `solid` integration config (tests/br.rs:35-59): raw.fasta corrected with
    raw.k11.a2.solid, method One, c=5, forward + reverse pass; byte-identical to the oracle."""
    monkeypatch.setenv("BRX_GROUP", group)
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    om = O.build_methods(os_, ["one"], 5, 7)
    for two_side in (False, True):
        chain = br_amd.Chain(gs, [("one", 5, 7)], two_side=two_side)
        got = chain.correct_reads(raw_reads)
        for r, g in zip(raw_reads, got):
            assert g == O.correct_record(om, r, two_side)
    if group == "16":
        st = chain.last_stats()
        assert st["probes"] > 0 and st["rounds"] > 0


@pytest.mark.parametrize("method", ["two", "graph", "gap_size", "greedy"])
@pytest.mark.parametrize("group", ["16", "64"])
def test_method_raw_fasta_fixture_set(raw_reads, solid_fixture_bytes, method, group, monkeypatch):
    """every corrector on the reference's integration data, forward + reverse, vs the oracle."""
    monkeypatch.setenv("BRX_GROUP", group)
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    om = O.build_methods(os_, [method], 5, 7)
    chain = br_amd.Chain(gs, [(method, 5, 7)], two_side=False)
    got = chain.correct_reads(raw_reads)
    changed = 0
    for r, g in zip(raw_reads, got):
        assert g == O.correct_record(om, r, False)
        changed += g != r
    assert changed > 100


def test_default_like_chain(raw_reads, solid_fixture_bytes):
    """all five methods chained in the reference's default order, src/cli.rs:121-131"""
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    names = ["one", "two", "graph", "greedy", "gap_size"]
    om = O.build_methods(os_, names, 5, 7)
    reads = raw_reads[:80]
    got = br_amd.Chain(gs, [(m, 5, 7) for m in names], two_side=False).correct_reads(reads)
    for r, g in zip(reads, got):
        assert g == O.correct_record(om, r, False)


def test_one_chained_methods(raw_reads, solid_fixture_bytes):
    """method chaining: the output of one corrector feeds the next (src/lib.rs:44-46)."""
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    reads = raw_reads[:60]
    spec = [("one", 5, 7), ("one", 2, 7), ("one", 9, 7)]
    om = [O.Corrector(os_, m, c, ms) for m, c, ms in spec]
    got = br_amd.Chain(gs, spec, two_side=False).correct_reads(reads)
    for r, g in zip(reads, got):
        assert g == O.correct_record(om, r, False)


def test_edge_reads(solid_fixture_bytes):
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    om = O.build_methods(os_, ["one"], 5, 7)
    chain = br_amd.Chain(gs, [("one", 5, 7)], two_side=False)
    assert chain.correct_reads([]) == []
    reads = [b"", b"A", b"ACGTACGTAC", b"ACGTACGTACG", b"ACGTACGTACGT", b"acgtnACGTNNxyzACGTACGTTTGACCA",
             b"N" * 300, b"ACGT" * 100]
    got = chain.correct_reads(reads)
    for r, g in zip(reads, got):
        assert g == O.correct_record(om, r, False)


def test_large_confirm(raw_reads, solid_fixture_bytes):
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    reads = raw_reads[:30]
    for c in (1, 12, 40):
        om = O.build_methods(os_, ["one"], c, 7)
        got = br_amd.Chain(gs, [("one", c, 7)], two_side=False).correct_reads(reads)
        for r, g in zip(reads, got):
            assert g == O.correct_record(om, r, False), c


def test_run_correction_end_to_end(tmp_path, golden_dir, solid_fixture_bytes):
    """br -i raw.fasta -o corr.fasta -c one solid -i raw.k11.a2.solid -f solid, via the python mirror."""
    gs = br_amd.Pcon.from_pcon_solid(open(os.path.join(golden_dir, "raw.k11.a2.solid"), "rb").read())
    methods = br_amd.build_methods(["one"], gs, 5, 7)
    out_path = tmp_path / "corr.fasta"
    with open(os.path.join(golden_dir, "raw.fasta"), "rb") as fi, open(out_path, "wb") as fo:
        br_amd.run_correction([fi], [fo], methods, two_side=False)
    from br_amd import fasta
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    om = O.build_methods(os_, ["one"], 5, 7)
    with open(os.path.join(golden_dir, "raw.fasta"), "rb") as fi, open(out_path, "rb") as fo:
        src = list(fasta.read_records(fi))
        got = list(fasta.read_records(fo))
    assert len(src) == len(got) == 206
    total = 0
    for (n0, d0, s0), (n1, d1, s1) in zip(src, got):
        assert (n0, d0) == (n1, d1)
        assert s1 == O.correct_record(om, s0, False)
        total += len(s1)
    assert total == 2520330


# ---------------------------------------------------------------- synthetic -----------------------
def _torch():
    import torch
    return torch


def test_synth_device_equals_host():
    torch = _torch()
    cfg = synth.config(genome_len=300_000, read_len=3_000)
    g = synth.genome_host(cfg)
    hb, ho = synth.reads_host(cfg, g, 5, 200)
    dg = torch.empty(cfg.genome_len, dtype=torch.uint8, device="cuda")
    synth.genome_device(cfg, 0, dg.data_ptr())
    assert np.array_equal(dg.cpu().numpy(), g)
    db = torch.empty(200 * 7000, dtype=torch.uint8, device="cuda")
    do = torch.empty(201, dtype=torch.int64, device="cuda")
    tot = synth.reads_device(cfg, 0, dg.data_ptr(), 5, 200, db.data_ptr(), db.numel(), do.data_ptr())
    assert tot == hb.size
    assert np.array_equal(do.cpu().numpy().astype(np.uint64), ho)
    assert np.array_equal(db[:tot].cpu().numpy(), hb)


@pytest.mark.parametrize("k", [13, 15])
def test_synthetic_build_and_correct(k):
    """small synthetic job end to end on device buffers: count -> threshold -> One fwd+rev."""
    torch = _torch()
    n_reads, read_len = 400, 2_000
    cfg = synth.config(genome_len=n_reads * read_len // 25, read_len=read_len)
    g = synth.genome_host(cfg)
    hb, ho = synth.reads_host(cfg, g, 0, n_reads)
    reads = [hb[int(ho[i]):int(ho[i + 1])].tobytes() for i in range(n_reads)]

    dg = torch.from_numpy(g).cuda()
    db = torch.empty(n_reads * (2 * read_len + 8), dtype=torch.uint8, device="cuda")
    do = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")
    tot = synth.reads_device(cfg, 0, dg.data_ptr(), 0, n_reads, db.data_ptr(), db.numel(), do.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream

    ref = O.Solid.from_count(k, O.count_reads(k, reads), 3)
    for strategy in STRATEGIES:
        cnt = br_amd.Counter(k, 0, strategy)
        cnt.add_batch_device(db.data_ptr(), do.data_ptr(), n_reads, tot, stream)
        gs = cnt.finish(3, stream)
        assert gs.to_solid_bytes() == ref.to_bytes()
        # reset + recount into an existing set (the bench's steady state)
        cnt.reset(stream)
        cnt.add_batch_device(db.data_ptr(), do.data_ptr(), n_reads, tot, stream)
        cnt.finish_into(3, gs, stream)
        torch.cuda.synchronize()
        assert gs.to_solid_bytes() == ref.to_bytes()
    assert gs.popcount() > 0

    chain = br_amd.Chain(gs, [("one", 5, 7)], two_side=False)
    d_out = torch.empty(int(tot * 1.1) + 4096, dtype=torch.uint8, device="cuda")
    d_oo = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")
    total = chain.correct_batch_device(db.data_ptr(), do.data_ptr(), n_reads, tot, d_out.data_ptr(), d_out.numel(),
                                       d_oo.data_ptr(), stream)
    out = d_out[:total].cpu().numpy()
    oo = d_oo.cpu().numpy()
    om = O.build_methods(ref, ["one"], 5, 7)
    exp, exp_o = O.correct_batch(om, hb, ho, False)
    assert np.array_equal(oo.astype(np.uint64), exp_o)
    assert np.array_equal(out, exp)
    st = chain.last_stats()
    assert st["fixes"] > 0 and st["triggers"] >= st["fixes"]


@pytest.mark.parametrize("method", ["one", "two", "graph", "greedy", "gap_size"])
def test_synthetic_every_method(method):
    """synthetic reads where most triggers are real isolated errors (lots of positive fixes)."""
    k, n_reads, read_len = 13, 160, 1500
    cfg = synth.config(genome_len=n_reads * read_len // 30, read_len=read_len, sub=0.01, ins=0.006, dele=0.006)
    g = synth.genome_host(cfg)
    hb, ho = synth.reads_host(cfg, g, 0, n_reads)
    reads = [hb[int(ho[i]):int(ho[i + 1])].tobytes() for i in range(n_reads)]
    gs = br_amd.Pcon.from_count(reads, k, 2)
    ref = O.Solid.from_count(k, O.count_reads(k, reads), 2)
    assert gs.to_solid_bytes() == ref.to_bytes()
    om = O.build_methods(ref, [method], 5, 7)
    chain = br_amd.Chain(gs, [(method, 5, 7)], two_side=False)
    out, oo = chain.correct_batch(hb, ho)
    exp, exp_o = O.correct_batch(om, hb, ho, False)
    assert np.array_equal(oo, exp_o)
    assert np.array_equal(out, exp)
    assert om[0].stats()["fixes"] > (50 if method == "greedy" else 200)


def test_greedy_parameters(raw_reads, solid_fixture_bytes):
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    reads = raw_reads[:40]
    for max_search, nbv in [(1, 2), (3, 0), (12, 3), (7, 20)]:
        om = [O.Corrector(os_, "greedy", nbv, max_search)]
        got = br_amd.Chain(gs, [("greedy", nbv, max_search)], two_side=False).correct_reads(reads)
        for r, g in zip(reads, got):
            assert g == O.correct_record(om, r, False), (max_search, nbv)


@pytest.mark.parametrize("redo_max", ["4096", "0"])
def test_output_slot_overflow_retry(monkeypatch, redo_max):
    """a read whose correction grows it far beyond its staging slot (Graph re-inserting a 150-base deletion).  Up to
    4096 such reads are taken out and redone by a second chain with more slack; one that still does not fit its slot of
    the batch is copied over its place in the compact output after the compaction (the rest of the batch is final).
    More than that (BRX_REDO_MAX=0 forces it) and the whole batch is redone with 4x the slack, which the chain keeps.
    Either way the oracle's bytes, in input order, with neighbours that must not be disturbed."""
    monkeypatch.setenv("BRX_REDO_MAX", redo_max)
    rng = np.random.default_rng(7)
    k = 15   # large enough that the 3 kb random genome has no branching (k-1)-mers
    genome = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=3000).tolist())
    ref = O.Solid(k)
    ref.set_seq(genome)
    gs = br_amd.Pcon.from_pcon_solid(ref.to_bytes())
    reads = [genome[2000:2300], genome[100:160] + genome[310:370], genome[1500:1530], genome[500:560] + genome[900:960],
             genome[1000:1400], b"", genome[2500:2600]]
    for method in ("graph", "gap_size"):
        for two_side in (True, False):
            om = O.build_methods(ref, [method], 5, 7)
            chain = br_amd.Chain(gs, [(method, 5, 7)], two_side=two_side)
            got = chain.correct_reads(reads)
            exp = [O.correct_record(om, r, two_side) for r in reads]
            assert got == exp
            if method == "graph":
                st = chain.last_stats()
                assert got[1] == genome[100:370] and got[3] == genome[500:960]
                assert st["slot_overflow_reads"] >= 2
                if redo_max == "0":
                    assert st["overflow_retries"] >= 1
                    # the chain remembers the workspace it needed: the same batch again runs once
                    assert chain.correct_reads(reads) == exp and chain.last_stats()["overflow_retries"] == 0
                else:
                    assert st["overflow_retries"] == 0          # the two reads were redone on their own
                    assert chain.correct_reads(reads) == exp


@pytest.mark.parametrize("redo_max", ["4096", "0"])
def test_walk_list_overflow_retry(raw_reads, solid_fixture_bytes, monkeypatch, redo_max):
    """a walk that outgrows its visited list poisons its read (BRX_MAXPATH=2: nearly every successful walk does).  Up
    to 4096 such reads are taken out, corrected by a second chain with a longer list and written back into their slots
    (the rest of the batch is final); more than that (BRX_REDO_MAX=0 forces it) and the whole batch runs again with a
    list 8x as long, which the chain then keeps.  Either way: the oracle's bytes, for every walking method, chained,
    with the reverse pass."""
    monkeypatch.setenv("BRX_MAXPATH", "2")
    monkeypatch.setenv("BRX_REDO_MAX", redo_max)
    monkeypatch.setenv("BRX_LANE_WALK", "0")   # (the visited lists are the group kernel's; the lane form keeps none)
    ref = O.Solid.from_bytes(solid_fixture_bytes)
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    reads = raw_reads[:24] + [b"", b"ACGT"]
    for methods in (["graph"], ["gap_size"], ["one", "gap_size", "graph"]):
        om = O.build_methods(ref, methods, 5, 7)
        chain = br_amd.Chain(gs, [(m, 5, 7) for m in methods], two_side=False)
        got = chain.correct_reads(reads)
        assert got == [O.correct_record(om, r, False) for r in reads]
        st = chain.last_stats()
        assert st["walk_list_overflows"] > 0 and st["slot_overflow_reads"] == 0
        if redo_max == "0":
            assert st["overflow_retries"] >= 2
            assert chain.correct_reads(reads) == got and chain.last_stats()["overflow_retries"] == 0
        else:
            assert st["overflow_retries"] == 0
            assert chain.correct_reads(reads) == got          # (the chain starts its next batch with a longer list)


def test_partitioned_exchange_primitives_two_shards(raw_reads):
    """the data path of the multi-GPU key exchange, emulated on one GPU: two shards counted separately,
    every 'owner' receives both shards' segments of its digit range (l1_view / add_partitioned), finishes
    its range, and the solid set is replicated sparsely (extract_keys / or_keys).  Same bits as one build."""
    import torch
    from br_amd import dist as bd
    k, a = 13, 2
    reads = raw_reads[:60]
    shards = [reads[:30], reads[30:]]
    stream = torch.cuda.current_stream().cuda_stream
    locals_ = []
    for sh in shards:
        c = br_amd.Counter(k, 0, _lib.COUNT_SORTED)
        bases, offs = br_amd.pack_reads(sh)
        db = torch.from_numpy(bases.copy()).cuda()
        do = torch.from_numpy(offs.astype(np.int64)).cuda()
        c.add_batch_device(db.data_ptr(), do.data_ptr(), len(sh), int(offs[-1]), stream)
        locals_.append((c, db, do))
    world = 2
    final = br_amd.Pcon.new(k)
    keep = []
    per_owner_sets = []
    for owner in range(world):
        owned = br_amd.Counter(k, 0, _lib.COUNT_SORTED)
        solid = br_amd.Pcon.new(k)
        for c, _, _ in locals_:
            pk, po, nb, nk = c.l1_view()
            keys = bd.device_view(pk, max(nk, 1) * 4).view(torch.int32)[:nk]
            t = bd.device_view(po, (nb + 1) * 8).view(torch.int64)
            bounds = bd.owner_bounds(nb, world)
            lo, hi = bounds[owner], bounds[owner + 1]
            seg_keys = keys[int(t[lo]):int(t[hi])].clone()
            seg_off = (t.clamp(min=t[lo], max=t[hi]) - t[lo]).contiguous()
            keep.append((seg_keys, seg_off))
            owned.add_partitioned_device(seg_keys.data_ptr(), seg_off.data_ptr(), seg_keys.numel())
        owned.finish_into(a, solid, stream)
        torch.cuda.synchronize()
        per_bucket = solid.n_hashes() // nb
        buf = torch.empty(1 << 20, dtype=torch.int64, device="cuda")
        n = solid.extract_keys_device(lo * per_bucket, (hi - lo) * per_bucket, buf.data_ptr(), buf.numel(), stream)
        final.or_keys_device(buf.data_ptr(), n, stream)
        torch.cuda.synchronize()
        per_owner_sets.append(n)
    ref = O.Solid.from_count(k, O.count_reads(k, reads), a)
    assert final.to_solid_bytes() == ref.to_bytes()
    assert sum(per_owner_sets) == ref.popcount() and min(per_owner_sets) > 0


def test_build_partitioned_over_rccl_single_rank(raw_reads):
    """SetExchange.build_partitioned end to end on device tensors with the real "nccl" (RCCL) backend,
    world_size 1 (the GPU box has one card; N > 1 is covered by the gloo tests and the emulation above)."""
    import torch
    import torch.distributed as dist
    from br_amd import dist as bd
    k, a = 15, 2
    reads = raw_reads[:50]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        stream = torch.cuda.current_stream().cuda_stream
        local = br_amd.Counter(k, 0, _lib.COUNT_SORTED)
        solid = br_amd.Pcon.new(k)
        bases, offs = br_amd.pack_reads(reads)
        db = torch.from_numpy(bases.copy()).cuda()
        do = torch.from_numpy(offs.astype(np.int64)).cuda()
        for _ in range(2):  # twice: the bench resets and reuses the same objects every step
            local.reset(stream)
            local.add_batch_device(db.data_ptr(), do.data_ptr(), len(reads), int(offs[-1]), stream)
            bd.SetExchange(1, 0).build_partitioned(local, solid, a, stream)
            torch.cuda.synchronize()
            ref = O.Solid.from_count(k, O.count_reads(k, reads), a)
            assert solid.to_solid_bytes() == ref.to_bytes()
    finally:
        dist.destroy_process_group()


def test_cli_reference_integration_commands(tmp_path, golden_dir, raw_reads, solid_fixture_bytes):
    """the reference's integration tests (tests/br.rs:9-59) as CLI invocations of the HIP path: exit 0,
    and -- which the reference never checks -- output equal to the oracle's."""
    from br_amd import cli, fasta
    raw = os.path.join(golden_dir, "raw.fasta")
    out1, out2 = str(tmp_path / "corr1.fasta"), str(tmp_path / "corr2.fasta")
    # tests/br.rs:35-59: br -i raw.fasta -o corr.fasta solid -i raw.k11.a2.solid -f solid   (all five methods)
    assert cli.main(["-i", raw, "-o", out1, "solid", "-i", os.path.join(golden_dir, "raw.k11.a2.solid"), "-f", "solid"]) == 0
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    om = O.build_methods(os_, ["one", "two", "graph", "greedy", "gap_size"], 5, 7)
    got = list(fasta.read_records(open(out1, "rb")))
    assert len(got) == 206
    for (_, _, seq), r in zip(got[:40], raw_reads[:40]):
        assert seq == O.correct_record(om, r, False)
    # tests/br.rs:9-33: br -i raw.fasta -o corr.fasta fasta -i raw.fasta -k 11 first-minimum
    assert cli.main(["-i", raw, "-o", out2, "-c", "one", "fasta", "-i", raw, "-k", "11", "first-minimum"]) == 0
    counts = O.count_reads(11, raw_reads)
    spec = np.bincount(counts, minlength=256)
    thr = cli.first_minimum(spec)
    assert thr == 6
    ref = O.Solid.from_count(11, counts, thr)
    om = O.build_methods(ref, ["one"], 5, 7)
    got = list(fasta.read_records(open(out2, "rb")))
    for (_, _, seq), r in zip(got[:60], raw_reads[:60]):
        assert seq == O.correct_record(om, r, False)


def test_config0_raw_fasta_k13_one(tmp_path, golden_dir, raw_reads):
    """BASELINE configs[0] as a whole: `br -i raw.fasta -o corr.fasta -c one fasta -i raw.fasta -k 13 -a 2` through the
    CLI path (count -> threshold -> correct::one forward + reverse -> 80-column FASTA), every one of the 206 records byte
    for byte against the oracle (src/main.rs:72-115, src/lib.rs:22-69); and `-k 14` is forced odd like the reference's
    Fasta::kmer_size (src/cli.rs:277-279), so it gives the same file."""
    from br_amd import cli, fasta
    raw = os.path.join(golden_dir, "raw.fasta")
    out, out14 = str(tmp_path / "corr.fasta"), str(tmp_path / "corr14.fasta")
    assert cli.main(["-i", raw, "-o", out, "-c", "one", "fasta", "-i", raw, "-k", "13", "-a", "2"]) == 0
    ref = O.Solid.from_count(13, O.count_reads(13, raw_reads), 2)
    om = O.build_methods(ref, ["one"], 5, 7)
    got = list(fasta.read_records(open(out, "rb")))
    assert len(got) == len(raw_reads) == 206
    changed = 0
    for (_, _, seq), r in zip(got, raw_reads):
        assert seq == O.correct_record(om, r, False)
        changed += seq != r
    assert changed > 150
    assert cli.main(["-i", raw, "-o", out14, "-c", "one", "fasta", "-i", raw, "-k", "14", "-a", "2"]) == 0
    assert open(out14, "rb").read() == open(out, "rb").read()


@pytest.mark.parametrize("strategy", [_lib.COUNT_DENSE, _lib.COUNT_SORTED])
def test_spectrum_matches_oracle(raw_reads, strategy):
    cnt = br_amd.Counter(11, 0, strategy)
    cnt.add_reads(raw_reads)
    spec = cnt.spectrum()
    exp = np.bincount(O.count_reads(11, raw_reads), minlength=256)
    assert np.array_equal(spec.astype(np.int64), exp.astype(np.int64))
    # the spectrum leaves the counter as it was: the set of the chosen threshold follows from the same counts
    s = cnt.finish(2)
    assert s.to_solid_bytes() == O.Solid.from_count(11, O.count_reads(11, raw_reads), 2).to_bytes()


@pytest.mark.parametrize("k", [13, 15, 17])
def test_spectrum_sorted_saturation_and_batches(k):
    """sorted-strategy spectrum over several batches, with hashes seen more than 255 times (bin 255 = 255 or more)"""
    rng = np.random.default_rng(500 + k)
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 3000).tobytes()
    reads = [genome[i:i + 400] for i in rng.integers(0, 2600, 3000)]            # ~400x coverage: counts beyond 255
    reads += [rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 300).tobytes() for _ in range(200)]
    reads += [b"", b"ACG", genome[:k], genome[:k - 1]]
    cnt = br_amd.Counter(k, 0, _lib.COUNT_SORTED)
    for lo in range(0, len(reads), 700):
        cnt.add_reads(reads[lo:lo + 700])
    spec = cnt.spectrum().astype(np.int64)
    exp = np.bincount(O.count_reads(k, reads), minlength=256).astype(np.int64)
    assert exp[255] > 0
    assert np.array_equal(spec, exp)
    assert int(spec.sum()) == 1 << (2 * k - 1)


def test_cli_abundance_methods(tmp_path, golden_dir, raw_reads):
    """`br ... fasta -k 11 rarefaction|percent-most|percent-least P` (src/main.rs:98-107): threshold from the GPU
    spectrum through br_amd.spectrum (pcon's formulas, unpinned), then the same set and corrections as `-a thr`"""
    from br_amd import cli, fasta, spectrum
    raw = os.path.join(golden_dir, "raw.fasta")
    counts = O.count_reads(11, raw_reads)
    spec = np.bincount(counts, minlength=256)
    for method, pct in (("rarefaction", "0.001"), ("percent-least", "0.2"), ("percent-most", "0.2")):
        thr = spectrum.get_threshold(spec, method, float(pct))
        assert thr is not None and 0 < thr < 255
        out = str(tmp_path / f"corr_{method}.fasta")
        assert cli.main(["-i", raw, "-o", out, "-c", "one", "fasta", "-i", raw, "-k", "11", method, pct]) == 0
        om = O.build_methods(O.Solid.from_count(11, counts, thr), ["one"], 5, 7)
        got = list(fasta.read_records(open(out, "rb")))
        assert len(got) == 206
        for (_, _, seq), r in zip(got[:25], raw_reads[:25]):
            assert seq == O.correct_record(om, r, False)
    assert spec[:8].tolist() == [1436018, 442564, 95498, 19526, 4458, 1221, 460, 494]


def test_cli_count_subcommand(tmp_path, golden_dir, raw_reads, solid_fixture_bytes):
    """`br ... count -i table -a 2` (src/main.rs:59-70): a count table ([k][u8 counters], layout unpinned) thresholded
    like `fasta` -- from the oracle's counts of raw.fasta at k = 11 it must give the reference's own .solid fixture"""
    import gzip
    from br_amd import cli, fasta
    raw = os.path.join(golden_dir, "raw.fasta")
    counts = O.count_reads(11, raw_reads)
    table = str(tmp_path / "raw.k11.pcon")
    with open(table, "wb") as f:
        f.write(bytes([11]) + counts.tobytes())
    with open(table, "rb") as f:
        cnt = br_amd.Counter.from_count_stream(f, 0, chunk=100000)
    assert np.array_equal(cnt.spectrum().astype(np.int64), np.bincount(counts, minlength=256).astype(np.int64))
    assert cnt.finish(2).to_solid_bytes() == solid_fixture_bytes
    table_gz = table + ".gz"
    with gzip.open(table_gz, "wb") as f:
        f.write(bytes([11]) + counts.tobytes())
    out_c, out_s = str(tmp_path / "c.fasta"), str(tmp_path / "s.fasta")
    assert cli.main(["-i", raw, "-o", out_c, "-c", "one", "count", "-i", table_gz, "-a", "2"]) == 0
    assert cli.main(["-i", raw, "-o", out_s, "-c", "one", "solid", "-i", os.path.join(golden_dir, "raw.k11.a2.solid"), "-f", "solid"]) == 0
    assert open(out_c, "rb").read() == open(out_s, "rb").read()
    out_m = str(tmp_path / "m.fasta")
    assert cli.main(["-i", raw, "-o", out_m, "-c", "one", "count", "-i", table, "first-minimum"]) == 0
    om = O.build_methods(O.Solid.from_count(11, counts, 6), ["one"], 5, 7)
    got = list(fasta.read_records(open(out_m, "rb")))
    for (_, _, seq), r in zip(got[:20], raw_reads[:20]):
        assert seq == O.correct_record(om, r, False)
    with open(table, "rb") as f:
        short = f.read(5000)
    import io
    with pytest.raises(_lib.BrxError):
        br_amd.Counter.from_count_stream(io.BytesIO(short), 0)


def test_cli_solid_fastq_and_csv(tmp_path, golden_dir, raw_reads):
    """`solid -f fastq|csv` / `large-kmer -f fastq` (src/set/pcon.rs:27-45,114-181; optional features of the reference):
    the same presence-only set as the FASTA form of the same sequences"""
    from br_amd import cli
    k = 11
    reads = raw_reads[:30]
    fa, fq, cs = (str(tmp_path / n) for n in ("r.fa", "r.fq", "k.csv"))
    with open(fa, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b">r%d\n%s\n" % (i, r))
    with open(fq, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@r%d desc\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))
        f.write(b"@broken\nACGT\n+\nII\n@never\nACGTACGTACGTACGT\n+\nIIIIIIIIIIIIIIII\n")   # stream ends at the bad record
    kmers = sorted({r[j:j + k] for r in reads[:5] for j in range(len(r) - k + 1)})
    with open(cs, "wb") as f:
        f.write(b"kmer,count\n" + b"".join(km + b",1\n" for km in kmers))
    ref = O.Solid(k)
    for r in reads:
        ref.set_seq(r)
    assert cli.presence_set(fa, "fasta", k, 0).to_solid_bytes() == ref.to_bytes()
    assert cli.presence_set(fq, "fastq", k, 0).to_solid_bytes() == ref.to_bytes()
    ref5 = O.Solid(k)
    for r in reads[:5]:
        ref5.set_seq(r)
    assert cli.presence_set(cs, "csv", k, 0).to_solid_bytes() == ref5.to_bytes()
    with open(cs, "ab") as f:
        f.write(b"ACGT,1\n")
    with pytest.raises(SystemExit):
        cli.presence_set(cs, "csv", k, 0)
    a = cli.parser().parse_args(["large-kmer", "-i", fq, "-f", "fastq", "-k", "21"])
    s21 = cli.build_set(a)
    assert s21.is_sparse()
    om = O.Solid.sparse_from_count(21, reads, 0)
    probe = [O.seq2bit(reads[3][j:j + 21]) for j in range(0, 400, 7)] + [O.seq2bit(b"ACGTTGCAACGTTGCAACGTA")]
    assert [s21.get(x) for x in probe] == [om.get(x) for x in probe]


@pytest.mark.parametrize("k", [13, 15, 19, 21])
def test_two_rank_exchange_on_one_gpu(tmp_path, raw_reads, k):
    """SetExchange.build_partitioned with world_size 2 for real: two processes share the card and talk over gloo
    (tests/dist_gpu_worker.py); each counts its half of the reads, the keys go to their owners, the solid lists come
    back, and each rank corrects its own shard against the set of ALL reads.  k = 13: bit vector + OR of the other
    rank's list; k = 15: no bit vector at finish time, the probe index is built from both ranks' lists; k = 19: the
    owner's finish is the LDS hash-count over its half of the digit range; k = 21 (BASELINE configs[4]'s k): sparse sets,
    four radix levels, the chained index built from both ranks' lists IS the set."""
    import pickle
    import subprocess
    import sys
    a, n_reads, world = 2, 60, 2
    port = 29600 + (os.getpid() + k) % 300
    prefix = str(tmp_path / "x")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_gpu_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(k), str(a), str(n_reads), prefix])
             for r in range(world)]
    try:
        codes = [p.wait(timeout=240) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert codes == [0, 0]
    reads = raw_reads[:n_reads]
    ref = O.Solid.from_count(k, O.count_reads(k, reads), a) if k <= 15 else O.Solid.sparse_from_count(k, reads, a)
    om = O.build_methods(ref, ["one", "graph"], 5, 7)
    expect = [O.correct_record(om, r, False) for r in reads]
    got = []
    for r in range(world):
        with open("%s.rank%d.pkl" % (prefix, r), "rb") as f:
            res = pickle.load(f)
        if k <= 15:
            assert res["solid_bytes"] == ref.to_bytes()          # every rank holds the set of all reads
        else:
            assert res["members"] == [ref.get(x) for x in res["sample"]]
        assert res["corrected_0"] == res["corrected_1"]
        if k >= 15:
            assert res["index"]["valid"]
        got += res["corrected_0"]
    assert got == expect                                       # shards concatenate to the input order


# ---------------------------------------------------------------- boundary ----------------------
def test_abi_smoke_c_host(tmp_path, raw_reads, solid_fixture_bytes):
    """A host with no Python and no torch in its process (tests/abi_smoke.c, plain C built by gcc) links libbrx.so,
    loads the reference's k=11 fixture, corrects 20 reads with the reference's default chain and compares with the
    bytes the oracle wrote to a file -- what a Rust `br` binding does."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "abi_smoke")
    assert os.path.exists(exe), "tests/abi_smoke not built: __graft_entry__.build()"
    reads = raw_reads[:20]
    bases, offs = br_amd.pack_reads(reads)
    s = O.Solid.from_bytes(solid_fixture_bytes)
    (tmp_path / "set.solid").write_bytes(solid_fixture_bytes)
    (tmp_path / "reads.bin").write_bytes(np.uint32(len(reads)).tobytes() + offs.tobytes() + bases.tobytes())
    env = {k: v for k, v in os.environ.items() if k not in ("LD_LIBRARY_PATH", "PYTHONPATH", "LD_PRELOAD")}
    for ids, names, two_side in (("0", ["one"], 0), ("0,1,2,3,4", ["one", "two", "graph", "greedy", "gap_size"], 0),
                                 ("2,4", ["graph", "gap_size"], 1)):
        exp, exp_o = O.correct_batch(O.build_methods(s, names, 5, 7), bases, offs, bool(two_side))
        (tmp_path / "expect.bin").write_bytes(exp_o.tobytes() + exp.tobytes())
        r = subprocess.run([exe, str(tmp_path / "set.solid"), str(tmp_path / "reads.bin"), str(tmp_path / "expect.bin"), ids,
                            str(two_side)], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode == 0 and "identical to the oracle" in r.stdout, (ids, r.returncode, r.stdout, r.stderr)


def test_two_host_threads_on_one_chain(raw_reads, solid_fixture_bytes):
    """include/brx.h: "a brx_chain_t owns its workspace and serialises concurrent calls" -- two host threads pushing
    different batches through ONE chain (host-pointer entry: upload, correct, download) each get their own result"""
    import threading
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    s = O.Solid.from_bytes(solid_fixture_bytes)
    chain = br_amd.Chain(gs, [("one", 5, 7), ("graph", 5, 7)], two_side=False)
    jobs = [raw_reads[0:30], raw_reads[30:45], raw_reads[45:90], raw_reads[90:100]]
    packed = [br_amd.pack_reads(j) for j in jobs]
    expect = [O.correct_batch(O.build_methods(s, ["one", "graph"], 5, 7), b, o, False) for b, o in packed]
    results = [None] * len(jobs)
    errors = []

    def work(tid):
        try:
            for rep in range(3):
                for j in range(tid, len(jobs), 2):
                    results[j] = chain.correct_batch(*packed[j])
        except Exception as e:  # pragma: no cover
            errors.append(e)

    ts = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    for (out, oo), (exp, exp_o) in zip(results, expect):
        assert np.array_equal(oo, exp_o) and np.array_equal(out, exp)


def test_exchange_behind_the_abi_single_rank():
    """brx_comm_* / brx_exchange_build_partitioned (librccl called from libbrx.so) with a world of one: the keys go
    through ncclSend/ncclRecv to self, the lists through the all-gather, and the set equals the plain finish"""
    import ctypes as C
    from br_amd import dist as D
    k, a = 15, 2
    cfg = synth.config(genome_len=60_000, read_len=3_000)
    g = synth.genome_host(cfg)
    bases, offs = synth.reads_host(cfg, g, 0, 400)
    plain = br_amd.Counter(k, 0, _lib.COUNT_SORTED)
    plain.add_batch(bases, offs)
    ref = plain.finish(a)
    os.environ["BRX_EXCHANGE_SELF_SEND"] = "1"
    try:
        ex = D.AbiExchange(1, 0, 0)
        for chunk in ("0", "100000"):       # one message / many rounds of capped messages
            os.environ["BRX_A2A_CHUNK"] = chunk
            cnt = br_amd.Counter(k, 0, _lib.COUNT_SORTED)
            cnt.add_batch(bases, offs)
            gs = br_amd.Pcon.new(k, 0)
            ex.build_partitioned(cnt, gs, a, None)
            st = ex.last_stats()
            assert st["solid_job"] == ref.popcount() == gs.popcount() and st["keys_counted_here"] > 0
            assert gs.to_solid_bytes() == ref.to_solid_bytes()
            # and the set corrects like the plain one (its probe index came from the gathered list)
            c1 = br_amd.Chain(gs, [("one", 5, 7)], two_side=False).correct_batch(bases, offs)
            c2 = br_amd.Chain(ref, [("one", 5, 7)], two_side=False).correct_batch(bases, offs)
            assert np.array_equal(c1[0], c2[0]) and np.array_equal(c1[1], c2[1])
        ex.close()
    finally:
        os.environ.pop("BRX_EXCHANGE_SELF_SEND", None)
        os.environ.pop("BRX_A2A_CHUNK", None)


def test_comm_init_all_one_process():
    """brx_comm_init_all: the one-process-many-GPUs form of the communicator (the reference's shape: one process,
    threads).  One device here; the exchange through it equals the plain finish."""
    import ctypes as C
    L = _lib.lib()
    devs = (C.c_int * 1)(0)
    comms = (C.c_void_p * 1)()
    _lib.check(L.brx_comm_init_all(1, devs, comms))
    try:
        w, r, d = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        _lib.check(L.brx_comm_info(comms[0], C.byref(w), C.byref(r), C.byref(d)))
        assert (w.value, r.value, d.value) == (1, 0, 0)
        k, a = 15, 1
        cfg = synth.config(genome_len=30_000, read_len=2_000)
        g = synth.genome_host(cfg)
        bases, offs = synth.reads_host(cfg, g, 0, 150)
        for strategy in (_lib.COUNT_SORTED, _lib.COUNT_DENSE):
            plain = br_amd.Counter(k, 0, strategy)
            plain.add_batch(bases, offs)
            ref = plain.finish(a)
            cnt = br_amd.Counter(k, 0, strategy)
            cnt.add_batch(bases, offs)
            if strategy == _lib.COUNT_SORTED:
                gs = br_amd.Pcon.new(k, 0)
                _lib.check(L.brx_exchange_build_partitioned(comms[0], cnt._h, a, gs._h, None))
            else:
                _lib.check(L.brx_exchange_reduce_counts(comms[0], cnt._h, a, None))   # world 1: nothing to add up
                gs = cnt.finish(a)
            assert gs.to_solid_bytes() == ref.to_solid_bytes()
    finally:
        L.brx_comm_free(comms[0])


def test_device_block_pool_reuse_and_trim(raw_reads):
    """Device memory of the library is pooled (brx_devpool.hip): a set built, dropped and built again reuses the parked
    blocks -- same bytes as the first build --, brx_devpool_trim hands everything parked back, and a third build after that
    is the same again.  (Blocks under 32 MiB bypass the pool: k = 15's 64 MiB bit vector / 512 MiB counter do not.)"""
    L = _lib.lib()
    k, a = 15, 1
    reads = raw_reads[:60]
    first = br_amd.Pcon.from_count(reads, k, a, strategy=_lib.COUNT_DENSE).to_solid_bytes()
    import gc
    gc.collect()
    parked = L.brx_devpool_bytes()
    assert parked >= (1 << 29)          # the dense counter table of k = 15 at least
    second = br_amd.Pcon.from_count(reads, k, a, strategy=_lib.COUNT_DENSE).to_solid_bytes()
    assert second == first
    gc.collect()
    L.brx_devpool_trim()
    assert L.brx_devpool_bytes() == 0
    third = br_amd.Pcon.from_count(reads, k, a, strategy=_lib.COUNT_SORTED).to_solid_bytes()
    assert third == first
