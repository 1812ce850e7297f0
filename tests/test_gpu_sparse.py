"""Sparse sets (include/brx.h brx_set_sparse): k >= 21, where the 2^(2k-1)-bit vector does not fit, keep their
solid k-mers as a key list + the probe index with chained lines.  Checked against the sparse oracle at k = 21, and
-- forced at small k (BRX_FORCE_SPARSE=1) -- against the pinned bitset oracle, for every corrector."""
import numpy as np
import pytest

import br_amd
from br_amd import _lib, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

METHODS = ["one", "two", "graph", "greedy", "gap_size"]


def _queries(k, reads, rng):
    fw = []
    for r in reads[:6]:
        for i in range(0, len(r) - k + 1, 3):
            w = r[i:i + k]
            if set(w) <= set(b"ACGT"):
                fw.append(O.seq2bit(w))
    fw = np.array(fw, dtype=np.uint64)
    mask = np.uint64((1 << (2 * k)) - 1)
    return np.concatenate([fw, fw ^ np.uint64(1), fw ^ (np.uint64(2) << np.uint64(2 * (k - 1))),
                           rng.integers(0, 1 << (2 * k), 5000, dtype=np.uint64)]) & mask


@pytest.mark.parametrize("log2_lines", ["0", "4"])
@pytest.mark.parametrize("k,a", [(15, 1), (11, 2)])
def test_forced_sparse_equals_bitset_oracle(raw_reads, k, a, log2_lines, monkeypatch):
    """"4" asks for a 16-line table: the build enlarges it just enough to hold the keys, so lines are full and
    most keys sit somewhere down a chain"""
    monkeypatch.setenv("BRX_FORCE_SPARSE", "1")
    monkeypatch.setenv("BRX_INDEX_LOG_LINES", log2_lines)
    reads = raw_reads[:120]
    cnt = br_amd.Counter(k, 0, _lib.COUNT_SORTED)
    cnt.add_reads(reads)
    gs = cnt.finish(a)
    assert gs.is_sparse()
    ref = O.Solid.from_count(k, O.count_reads(k, reads), a)
    assert gs.popcount() == ref.popcount()
    q = _queries(k, reads, np.random.default_rng(k))
    want = np.array([ref.get(int(x)) for x in q])
    assert np.array_equal(gs.get_many(q), want)
    info = gs.index_info()
    assert info["valid"] and info["keys"] == ref.popcount()
    if log2_lines == "4":
        assert info["overflow_keys"] > info["keys"] // 10
    for method in METHODS:
        om = O.build_methods(ref, [method], 5, 7)
        sub = reads[:40] if method == "greedy" else reads
        got = br_amd.Chain(gs, [(method, 5, 7)], two_side=False).correct_reads(sub)
        for r, g in zip(sub, got):
            assert g == O.correct_record(om, r, False), method
    with pytest.raises(_lib.BrxError):
        gs.to_solid_bytes()
    with pytest.raises(_lib.BrxError):
        gs.set(0, True)


def test_k21_small_vs_sparse_oracle(raw_reads):
    k, a = 21, 1
    reads = raw_reads[:150]
    gs = br_amd.Pcon.from_count(reads, k, a)
    assert gs.is_sparse() and gs.k() == 21
    ref = O.Solid.sparse_from_count(k, reads, a)
    assert gs.popcount() == ref.popcount() > 1000
    q = _queries(k, reads, np.random.default_rng(21))
    want = np.array([ref.get(int(x)) for x in q])
    assert np.array_equal(gs.get_many(q), want)
    assert want.any() and not want.all()
    for method in METHODS:
        om = O.build_methods(ref, [method], 5, 7)
        sub = reads[:40] if method == "greedy" else reads
        chain = br_amd.Chain(gs, [(method, 5, 7)], two_side=False)
        got = chain.correct_reads(sub)
        for r, g in zip(sub, got):
            assert g == O.correct_record(om, r, False), method
    names = ["graph", "gap_size"]  # BASELINE configs[4]'s chain
    om = O.build_methods(ref, names, 5, 7)
    got = br_amd.Chain(gs, [(m, 5, 7) for m in names], two_side=False).correct_reads(reads[:60])
    for r, g in zip(reads[:60], got):
        assert g == O.correct_record(om, r, False)


def test_k21_synthetic_properties_and_oracle_sample():
    """8e3 synthetic 10 kb reads at k = 21 (4 partition levels, no bit vector): nearly every genome k-mer is solid,
    random 21-mers are not, all-solid input comes back unchanged, and a sample of reads matches the sparse oracle
    built from the very same reads."""
    import torch
    k, a, n_reads, read_len = 21, 3, 8_000, 10_000
    cfg = synth.config(genome_len=n_reads * read_len // 50, read_len=read_len)
    g = synth.genome_host(cfg)
    hb, ho = synth.reads_host(cfg, g, 0, n_reads)
    stream = torch.cuda.current_stream().cuda_stream
    db, do = torch.from_numpy(hb).cuda(), torch.from_numpy(ho.astype(np.int64)).cuda()
    cnt = br_amd.Counter(k, 0)
    cnt.add_batch_device(db.data_ptr(), do.data_ptr(), n_reads, int(ho[-1]), stream)
    gs = cnt.finish(a, stream)
    assert gs.is_sparse()
    n_solid = gs.popcount()
    assert 0.9 * cfg.genome_len < n_solid < 1.2 * cfg.genome_len
    code = (g[:100_000] >> 1) & 3
    kmers = np.zeros(len(code) - k + 1, dtype=np.uint64)
    for j in range(k):
        kmers = (kmers << np.uint64(2)) | code[j:j + len(kmers)].astype(np.uint64)
    assert gs.get_many(kmers).mean() > 0.97
    assert gs.get_many(np.random.default_rng(1).integers(0, 1 << 42, 50_000, dtype=np.uint64)).mean() < 0.01
    # oracle on a sample: the sparse oracle needs every read's k-mers (8e7 hashes: np.unique handles it)
    reads = [hb[int(ho[i]):int(ho[i + 1])].tobytes() for i in range(n_reads)]
    ref = O.Solid.sparse_from_count(k, reads, a)
    assert ref.popcount() == n_solid
    sample = list(range(0, n_reads, n_reads // 24))
    for names in (["one"], ["graph", "gap_size"]):
        om = O.build_methods(ref, names, 5, 7)
        got = br_amd.Chain(gs, [(m, 5, 7) for m in names], two_side=False).correct_reads([reads[i] for i in sample])
        for i, gseq in zip(sample, got):
            assert gseq == O.correct_record(om, reads[i], False), (names, i)


@pytest.mark.parametrize("k", [21, 25, 31])
def test_large_kmer_presence_set_vs_oracle(raw_reads, k):
    """`br large-kmer -f fasta -k K` (set::Hash::from_fasta, src/set/hash.rs:40-60): every canonical k-mer of every
    record, inserted k-mer by k-mer into the chained table (several batches, so the table regrows); membership and
    every corrector against the sparse oracle with the same presence semantics (count > 0)."""
    trusted = raw_reads[:60]
    gs = br_amd.Pcon.from_fasta(trusted, k, batch=16)
    assert gs.is_sparse()
    ref = O.Solid.sparse_from_count(k, trusted, 0)
    assert gs.popcount() == ref.popcount() > 10_000
    q = _queries(k, trusted, np.random.default_rng(k))
    want = np.array([ref.get(int(x)) for x in q])
    assert np.array_equal(gs.get_many(q), want)
    # correct OTHER reads (and the trusted ones) against the presence set
    reads = raw_reads[40:100]
    for method in METHODS:
        om = O.build_methods(ref, [method], 5, 7)
        sub = reads[:25] if method == "greedy" else reads
        got = br_amd.Chain(gs, [(method, 5, 7)], two_side=False).correct_reads(sub)
        for r, g in zip(sub, got):
            assert g == O.correct_record(om, r, False), (k, method)


def test_large_kmer_cli(tmp_path, golden_dir, raw_reads):
    import os
    from br_amd import cli
    src = os.path.join(golden_dir, "raw.fasta")
    dst = tmp_path / "corr.fasta"
    assert cli.main(["-i", src, "-o", str(dst), "-c", "one", "-s", "large-kmer", "-i", src, "-f", "fasta", "-k", "23"]) == 0
    ref = O.Solid.sparse_from_count(23, raw_reads, 0)
    # every k-mer of the input is in the set: nothing triggers, the reads come back unchanged (wrapped at 80 columns)
    from br_amd import fasta
    got = [seq for _, _, seq in fasta.read_records(open(dst, "rb"))]
    assert got == raw_reads and ref.popcount() > 0
    with pytest.raises(SystemExit):
        cli.main(["-i", src, "-o", str(dst), "large-kmer", "-i", src, "-f", "fasta", "-k", "24"])


@pytest.mark.parametrize("k", [15, 23])
def test_presence_insert_one_long_record(k):
    """`solid -f fasta` / `large-kmer` on genome-like input: a single 1.5 Mbp record (plus records shorter than k and
    an empty one) is spread over the whole chip by tiles of the flat base stream; same set as the oracle's"""
    cfg = synth.config(genome_len=1_500_000, read_len=1000)
    g = synth.genome_host(cfg).tobytes()
    recs = [b"ACGT", g, b"", b"ACGTACGTACGTAC", g[1000:1000 + k], g[5000:5000 + k - 1]]
    gs = br_amd.Pcon.from_fasta(recs, k)
    if k == 15:
        ref = O.Solid(k)
        for r in recs:
            if len(r) >= k:
                ref.set_seq(r)
        assert gs.to_solid_bytes() == ref.to_bytes()
    else:
        assert gs.is_sparse()
        ref = O.Solid.sparse_from_count(k, recs, 0)
        assert gs.popcount() == ref.popcount()
        rng = np.random.default_rng(2)
        starts = rng.integers(0, len(g) - k, 4000)
        q = np.array([O.seq2bit(g[s:s + k]) for s in starts] + rng.integers(0, 1 << (2 * k), 4000, dtype=np.uint64).tolist(),
                     dtype=np.uint64)
        want = np.array([ref.get(int(x)) for x in q])
        assert np.array_equal(gs.get_many(q), want) and want[:4000].all()
