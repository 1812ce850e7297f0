"""The native host pipeline (brx_run_correction_fd / brx_count_fasta_fd: FASTA parse -> GPU -> FASTA write on
threads) against the record-by-record Python driver and the oracle: same bytes, whatever the batch size,
the line structure of the input or the kind of file object."""
import gzip
import io
import os

import numpy as np
import pytest

import br_amd
from br_amd import fasta
from br_amd.driver import run_correction
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _expected(text: bytes, om, two_side: bool) -> bytes:
    out = io.BytesIO()
    for name, desc, seq in fasta.read_records(io.BytesIO(text)):
        fasta.write_record(out, name, desc, O.correct_record(om, seq, two_side))
    return out.getvalue()


@pytest.mark.parametrize("writers", ["1", "3"])
@pytest.mark.parametrize("batch_records", [0, 7, 64])
@pytest.mark.parametrize("two_side", [False, True])
def test_native_pipeline_fixture(tmp_path, golden_dir, solid_fixture_bytes, batch_records, two_side, writers, monkeypatch):
    """writers = 3: BRX_PIPE_WRITERS, positional writes of the batches from three threads (regular file outputs only);
    the file and the position the descriptor is left at must be the same as with the one in-order writer"""
    monkeypatch.setenv("BRX_PIPE_WRITERS", writers)
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    om = O.build_methods(os_, ["one", "graph"], 5, 7)
    methods = br_amd.build_methods(["one", "graph"], gs, 5, 7)
    src = os.path.join(golden_dir, "raw.fasta")
    dst = tmp_path / "corr.fasta"
    with open(src, "rb") as fi, open(dst, "wb") as fo:
        fo.write(b"#head\n")  # the output starts where the descriptor stands ...
        st = run_correction([fi], [fo], methods, two_side, native=True, batch_records=batch_records)
        fo.write(b"#tail\n")  # ... and the descriptor is left at the end of what was written
    want = _expected(open(src, "rb").read(), om, two_side)
    got = dst.read_bytes()
    assert got[:6] == b"#head\n" and got[-6:] == b"#tail\n"
    assert got[6:-6] == want
    assert st["records"] == 206 and st["bases_in"] == 2519592
    if batch_records == 7:
        assert st["batches"] >= 206 // 7
    # the Python driver says the same
    out = io.BytesIO()
    with open(src, "rb") as fi:
        run_correction([fi], [out], methods, two_side, native=False)
    assert out.getvalue() == want


EDGE = (b">r1 first read \t with   description  \r\n"
        b"ACGTACGTTTGACCAGTACGATCGATCGGGATCAGCTAGCATCGACTAGCTAGCATCGATCAGCATCGACTAGCATCGACTAGCTACGACTAGCATCAGCATCAGCT\r\n"
        b"acgtnnACGTTGCA\n"
        b"\n"
        b"GGGTTTAAACCC\r\r\n"
        b">r2\n"
        b">r3\tdesc\n"
        b"ACGT\n"
        b">r4 last one without newline at the end\n"
        b"TTGACCAGTACGATCGATCGGGATCAGCTAGCATCGACTAGCTAGCATCGATCAGCATCGACTAGCATCGACTAGCTACGACTAGCATCAGCATCAGCTAAAAAAAAAAAAAAAAAAAAAAAAAAAACCCCCCCCCCCGGGGGGGT")


@pytest.mark.parametrize("tail", [b"", b"\n>\nACGT\n>r6\nACGT\n", b"\n> leading space\nACGT\n"])
def test_native_pipeline_edge_cases(solid_fixture_bytes, tail):
    """CRLF, blank and wrapped sequence lines, descriptions, an empty record, a missing final newline; a
    definition without a name ends the stream silently after the records before it (src/lib.rs:35)."""
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    om = O.build_methods(os_, ["one"], 5, 7)
    methods = br_amd.build_methods(["one"], gs, 5, 7)
    text = EDGE + tail
    want = _expected(text, om, False)
    assert want.count(b">") == 4
    for batch_records in (0, 1, 3):
        out = io.BytesIO()  # BytesIO on both sides: pipe bridges
        st = run_correction([io.BytesIO(text)], [out], methods, False, native=True, batch_records=batch_records)
        assert out.getvalue() == want
        assert st["records"] == 4


def test_native_pipeline_rejects_data_before_definition(solid_fixture_bytes):
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    methods = br_amd.build_methods(["one"], gs, 5, 7)
    out = io.BytesIO()
    st = run_correction([io.BytesIO(b"ACGT\n>r1\nACGT\n")], [out], methods, False, native=True)
    assert out.getvalue() == b"" and st["records"] == 0
    out = io.BytesIO()
    run_correction([io.BytesIO(b"")], [out], methods, False, native=True)
    assert out.getvalue() == b""


def test_native_pipeline_gzip_input_and_many_batches(tmp_path, golden_dir, solid_fixture_bytes):
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    os_ = O.Solid.from_bytes(solid_fixture_bytes)
    om = O.build_methods(os_, ["one"], 5, 7)
    methods = br_amd.build_methods(["one"], gs, 5, 7)
    raw = open(os.path.join(golden_dir, "raw.fasta"), "rb").read()
    gz = tmp_path / "raw.fasta.gz"
    with gzip.open(gz, "wb") as f:
        f.write(raw)
    dst = tmp_path / "corr.fasta"
    with gzip.open(gz, "rb") as fi, open(dst, "wb") as fo:
        st = run_correction([fi], [fo], methods, False, native=True, batch_records=5)
    assert dst.read_bytes() == _expected(raw, om, False)
    assert st["batches"] >= 41


@pytest.mark.parametrize("strategy", ["auto", "dense"])
def test_count_fasta_native(golden_dir, solid_fixture_bytes, strategy):
    """`br fasta -k 11 -a 2` through brx_count_fasta_fd rebuilds the reference's own .solid fixture."""
    from br_amd import _lib
    cnt = br_amd.Counter(11, 0, {"auto": _lib.COUNT_AUTO, "dense": _lib.COUNT_DENSE}[strategy])
    with open(os.path.join(golden_dir, "raw.fasta"), "rb") as f:
        st = cnt.count_fasta(f)
    assert st["records"] == 206 and st["bases_in"] == 2519592
    assert cnt.finish(2).to_solid_bytes() == solid_fixture_bytes
    # k = 15 goes the partitioned way; compare with the oracle's count
    cnt = br_amd.Counter(15, 0)
    with open(os.path.join(golden_dir, "raw.fasta"), "rb") as f:
        cnt.count_fasta(f)
    reads = [seq for _, _, seq in fasta.read_records(open(os.path.join(golden_dir, "raw.fasta"), "rb"))][:]
    ref = O.Solid.from_count(15, O.count_reads(15, reads), 1)
    assert cnt.finish(1).to_solid_bytes() == ref.to_bytes()


@pytest.mark.parametrize("k", [11, 23])
def test_presence_set_from_fasta_stream(golden_dir, raw_reads, k):
    """`br solid -f fasta` / `br large-kmer -f fasta` through brx_set_insert_fasta_fd: the same set as inserting the
    parsed records batch by batch, and as the oracle's presence-only set (bit vector at k = 11, sparse at k = 23)"""
    with open(os.path.join(golden_dir, "raw.fasta"), "rb") as f:
        gs = br_amd.Pcon.from_fasta_file(f, k)
    if k == 11:
        ref = O.Solid(k)
        for r in raw_reads:
            ref.set_seq_canonical(r) if hasattr(ref, "set_seq_canonical") else ref.set_seq(r)
        assert gs.to_solid_bytes() == br_amd.Pcon.from_fasta(raw_reads, k).to_solid_bytes()
        assert gs.popcount() == ref.popcount()
    else:
        assert gs.is_sparse()
        ref = O.Solid.sparse_from_count(k, raw_reads, 0)
        assert gs.popcount() == ref.popcount()
        rng = np.random.default_rng(5)
        q = np.concatenate([np.array([O.seq2bit(r[i:i + k]) for r in raw_reads[:5] for i in range(0, 400, 7)], dtype=np.uint64),
                            rng.integers(0, 1 << (2 * k), 3000, dtype=np.uint64)])
        assert np.array_equal(gs.get_many(q), np.array([ref.get(int(x)) for x in q]))


def test_async_batches_rotate_over_three_chains(raw_reads, solid_fixture_bytes):
    """brx_chain_correct_batch_async / _wait: one host thread keeps three chains of a set busy in turn (the
    double-buffered form of run_correction's batch loop, src/lib.rs:84-132).  Every batch equals the synchronous call
    and the oracle; a second _async on a busy chain and a _wait on an idle one are refused; freeing a chain with a
    batch in flight waits for it."""
    from br_amd import _lib
    from br_amd.correct import pack_reads
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    ref = O.Solid.from_bytes(solid_fixture_bytes)
    methods = [("one", 5, 7), ("graph", 5, 7)]
    om = O.build_methods(ref, ["one", "graph"], 5, 7)
    batches = [raw_reads[i:i + 20] for i in range(0, 200, 20)]
    packed = [pack_reads(b) for b in batches]
    chains = [br_amd.Chain(gs, methods, two_side=False) for _ in range(3)]
    got = [None] * len(batches)
    for j, (bases, offs) in enumerate(packed):
        ch = chains[j % 3]
        if j >= 3:
            got[j - 3] = ch.correct_batch_wait()
        ch.correct_batch_async(bases, offs)
    with pytest.raises(_lib.BrxError):
        chains[(len(packed) - 1) % 3].correct_batch_async(*packed[0])     # busy
    for j in range(len(packed) - 3, len(packed)):
        got[j] = chains[j % 3].correct_batch_wait()
    with pytest.raises(_lib.BrxError):
        chains[0].correct_batch_wait()                                    # nothing in flight
    sync = br_amd.Chain(gs, methods, two_side=False)
    for b, (out, oo) in zip(batches, got):
        reads = [out[int(oo[i]):int(oo[i + 1])].tobytes() for i in range(len(b))]
        assert reads == sync.correct_reads(b)
        assert reads == [O.correct_record(om, r, False) for r in b]
    chains[1].correct_batch_async(*packed[0])
    del chains                                                            # brx_chain_free joins the batch in flight
