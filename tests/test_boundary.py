"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/brx.h declares, fails loudly without a GPU, and the host logic (FASTA I/O, packing,
synthetic generator) behaves.  No GPU compute here."""
import io
import os
import re

import numpy as np
import pytest

import br_amd
from br_amd import _lib, fasta, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "brx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(brx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/brx.h but not exported by libbrx.so"
    # and the python binding table covers the header exactly
    assert sorted(_lib.SIGNATURES) == syms


def test_no_gpu_fails_loudly():
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_lib.BrxError) as e:
        br_amd.Pcon.new(11)
    assert e.value.status == _lib.BRX_ERR_NODEVICE
    assert "no CPU fallback" in str(e.value)
    with pytest.raises(_lib.BrxError):
        br_amd.Counter(11)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "br_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "br_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_pack_reads_and_seq2bit():
    bases, offs = br_amd.pack_reads([b"ACGT", b"", b"TTG"])
    assert bases.tobytes() == b"ACGTTTG" and offs.tolist() == [0, 4, 4, 7]
    assert br_amd.seq2bit(b"ACTG") == 0b00011011


def test_fasta_roundtrip_and_wrap():
    src = b">r1 some description\nACGT\nAC\n>r2\n" + b"A" * 170 + b"\n>r3\n\n"
    recs = list(fasta.read_records(io.BytesIO(src)))
    assert recs[0] == (b"r1", b"some description", b"ACGTAC")
    assert recs[1] == (b"r2", None, b"A" * 170)
    assert recs[2] == (b"r3", None, b"")
    out = io.BytesIO()
    for r in recs:
        fasta.write_record(out, *r)
    lines = out.getvalue().split(b"\n")
    assert lines[0] == b">r1 some description" and lines[1] == b"ACGTAC"
    assert lines[2] == b">r2" and [len(l) for l in lines[3:6]] == [80, 80, 10]
    # malformed input ends the stream silently (src/lib.rs:35)
    assert list(fasta.read_records(io.BytesIO(b"ACGT\n>r\nAC\n"))) == []


def test_synth_host_deterministic_and_error_rates():
    cfg = synth.config(genome_len=200_000, read_len=2_000)
    g = synth.genome_host(cfg)
    assert set(np.unique(g).tolist()) == {65, 67, 71, 84}
    b1, o1 = synth.reads_host(cfg, g, 0, 50)
    b2, o2 = synth.reads_host(cfg, g, 0, 50)
    assert np.array_equal(b1, b2) and np.array_equal(o1, o2)
    # slices regenerate identically (used by the CPU baseline)
    b3, o3 = synth.reads_host(cfg, g, 10, 5)
    assert np.array_equal(b3, b1[int(o1[10]):int(o1[15])])
    lens = np.diff(o1.astype(np.int64))
    assert abs(lens.mean() - 2000) < 30            # ins and del rates cancel
    # an error-free config reproduces reference windows (either strand)
    cfg0 = synth.config(genome_len=200_000, read_len=500, sub=0, ins=0, dele=0)
    b0, o0 = synth.reads_host(cfg0, g, 0, 20)
    gs = g.tobytes()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    for i in range(20):
        r = b0[int(o0[i]):int(o0[i + 1])].tobytes()
        assert len(r) == 500 and (r in gs or r.translate(comp)[::-1] in gs)


def test_cli_surface_matches_reference():
    """flags/defaults/quirks of src/cli.rs, without touching the GPU."""
    from br_amd import cli
    p = cli.parser()
    a = p.parse_args(["-i", "a.fa", "-i", "b.fa", "-o", "o.fa", "-s", "-c", "one", "-c", "gap-size", "-C", "3", "-M", "9",
                      "-t", "4", "fasta", "-i", "r.fa", "-k", "14", "-a", "2"])
    assert a.inputs == ["a.fa", "b.fa"] and a.outputs == ["o.fa"] and a.two_side
    assert a.corrections == ["one", "gap-size"] and a.confirm == 3 and a.max_search == 9
    assert a.subcommand == "fasta" and a.sub_inputs == ["r.fa"] and a.abundance == 2
    assert cli.fasta_kmer_size(a.kmer_size) == 13          # src/cli.rs:459
    assert cli.fasta_kmer_size(13) == 13
    a = p.parse_args(["fasta", "-i", "r.fa", "-k", "11", "first-minimum"])
    assert a.inputs is None and a.corrections is None and a.abundance is None
    assert a.abundance_selection == "first-minimum"
    a = p.parse_args(["solid", "-i", "x.solid", "-f", "solid"])
    assert a.subcommand == "solid" and a.format == "solid" and a.kmer_size is None
    # spectrum of raw.fasta at k=11: first rise at index 6 -> 7
    assert cli.first_minimum([1436018, 442564, 95498, 19526, 4458, 1221, 460, 494, 100]) == 6
    assert cli.first_minimum([5, 4, 3]) is None
    a = p.parse_args(["fasta", "-i", "r.fa", "-k", "11", "percent-most", "0.25"])
    assert a.abundance_selection == "percent-most" and a.percent == 0.25
    assert cli.METHOD_NAMES == ["one", "two", "graph", "greedy", "gap-size"]   # default order, src/cli.rs:121-131


def test_abundance_threshold_methods():
    """br_amd.spectrum (pcon::spectrum::ThresholdMethod, unpinned restatement) on a spectrum worked by hand:
    index*value = 0,50,40,15,40,150,48,14; running sums 0,50,90,105,145,295,343,357"""
    from br_amd import spectrum as sp
    S = [100, 50, 20, 5, 10, 30, 8, 2]
    assert sp.first_minimum(S) == 3                      # 10 > 5
    assert sp.rarefaction(S, 0.1) == 3                   # 5/105 < 0.1 (100/0 = inf, 50/50, 20/90 are not)
    assert sp.rarefaction(S, 0.001) is None
    assert sp.percent_at_least(S, 0.3) == 4              # 145/357 = 0.406 is the first ratio > 0.3
    assert sp.percent_at_most(S, 0.3) == 3
    assert sp.percent_at_least(S, 1.0) is None
    assert sp.get_threshold(S, "percent-least", 0.3) == 4 and sp.get_threshold(S, "percent-most", 0.3) == 3
    assert sp.get_threshold(S, "rarefaction", 0.1) == 3 and sp.get_threshold(S, "first-minimum") == 3
    Z = [0] * 256                                        # nothing counted: 0/0 ratios compare false
    assert sp.first_minimum(Z) is None and sp.rarefaction(Z, 0.5) is None and sp.percent_at_least(Z, 0.5) is None
    assert sp.percent_at_most(Z, 0.5) is None
    import pytest
    with pytest.raises(ValueError):
        sp.get_threshold(S, "median")


def test_fastq_and_csv_readers():
    """host-side readers behind `solid|large-kmer -f fastq|csv` (the reference's optional features, src/set/pcon.rs:27-45,
    114-181): four-line FASTQ records, the stream ends silently at the first malformed one; CSV: header row skipped,
    first column, empty lines skipped, a field that is not a k-mer is an error here"""
    import io
    import pytest
    from br_amd import fasta
    fq = b"@r1 d\nACGTA\n+\nIIIII\n@r2\r\nGG\r\n+r2\r\nII\r\n@bad\nAC\n+\nI\n@r3\nAAAA\n+\nIIII\n"
    assert list(fasta.read_fastq_sequences(io.BytesIO(fq))) == [b"ACGTA", b"GG"]
    assert list(fasta.read_fastq_sequences(io.BytesIO(b""))) == []
    assert list(fasta.read_fastq_sequences(io.BytesIO(b">r1\nACGT\n"))) == []          # not FASTQ: nothing
    assert list(fasta.read_fastq_sequences(io.BytesIO(b"@r1\nACGT\n+\nIIII"))) == [b"ACGT"]   # no final newline
    assert list(fasta.read_fastq_sequences(io.BytesIO(b"@ r1\nACGT\n+\nIIII\n"))) == []    # empty name
    assert list(fasta.read_csv_kmers(io.BytesIO(b"kmer,count\nACG,3\n\nTTT,1\n"), 3)) == [b"ACG", b"TTT"]
    assert list(fasta.read_csv_kmers(io.BytesIO(b"kmer\n"), 3)) == []
    assert list(fasta.read_csv_kmers(io.BytesIO(b""), 3)) == []
    with pytest.raises(ValueError):
        list(fasta.read_csv_kmers(io.BytesIO(b"kmer\nACGT\n"), 3))


def test_c_host_links_without_python_or_torch():
    """tests/abi_smoke (plain C, gcc) links libbrx.so and gets libamdhip64 through libbrx's own RUNPATH; with no
    GPU it must say so and exit 4 (no CPU fallback), not crash on a missing library"""
    import subprocess
    exe = os.path.join(ROOT, "tests", "abi_smoke")
    if not os.path.exists(exe):
        pytest.skip("tests/abi_smoke not built (__graft_entry__.build())")
    if _lib.device_count() > 0:
        pytest.skip("GPU present: covered by test_gpu_parity.py::test_abi_smoke_c_host")
    env = {k: v for k, v in os.environ.items() if k not in ("LD_LIBRARY_PATH", "PYTHONPATH")}
    r = subprocess.run([exe, "a", "b", "c", "0"], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode == 4 and "no GPU visible" in r.stderr, (r.returncode, r.stderr)


def test_cli_rejects_values_outside_u8():
    """-a / -C / -M are u8 in the reference (src/cli.rs:46,50,196): clap rejects 256, it does not wrap to 0"""
    from br_amd import cli
    for argv in (["-C", "256", "solid", "-i", "x", "-f", "solid"], ["-M", "-1", "solid", "-i", "x", "-f", "solid"],
                 ["fasta", "-i", "x", "-k", "11", "-a", "300"]):
        with pytest.raises(SystemExit):
            cli.parser().parse_args(argv)
    with pytest.raises(ValueError):
        br_amd.set._check_u8(300, "abundance")


# ---------------------------------------------------------------- multi-GPU exchange: host arithmetic ----------
def _plan(tables, world, nb, rank):
    import ctypes as C
    L = _lib.lib()
    T = nb + 1
    tab = np.ascontiguousarray(np.asarray(tables, dtype=np.uint64).reshape(world * T))
    bound = np.zeros(world + 1, np.uint32)
    sc, rc = np.zeros(world, np.uint64), np.zeros(world, np.uint64)
    seg = np.zeros(world * T, np.uint64)
    big = C.c_uint64(0)
    u64p, u32p = C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
    st = L.brx_exchange_plan(tab.ctypes.data_as(u64p), world, nb, rank, bound.ctypes.data_as(u32p), sc.ctypes.data_as(u64p),
                             rc.ctypes.data_as(u64p), seg.ctypes.data_as(u64p), C.byref(big))
    return st, bound, sc, rc, seg.reshape(world, T), int(big.value)


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("nb", [4, 13, 512])
def test_exchange_plan_layout(world, nb):
    """brx_exchange_plan, the pure host arithmetic of brx_exchange_build_partitioned (owner bounds, per-peer counts,
    clamped + rebased segment tables, the job's largest message), against a direct numpy statement on synthetic
    level-1 offset tables -- including empty ranks and ranks whose keys all sit in one digit."""
    if nb < world:
        pytest.skip("fewer buckets than ranks is covered by test_exchange_plan_more_ranks_than_buckets")
    rng = np.random.default_rng(world * 1000 + nb)
    counts = rng.integers(0, 50, size=(world, nb)).astype(np.uint64)
    if world > 1:
        counts[1] = 0                                           # a rank that counted nothing
    if world > 2:
        counts[2] = 0
        counts[2, nb - 1] = 12345                               # everything in the last digit
    tables = np.concatenate([np.zeros((world, 1), np.uint64), np.cumsum(counts, axis=1)], axis=1)
    exp_bound = [r * nb // world for r in range(world + 1)]
    sent_total = np.zeros((world, world), np.uint64)
    for me in range(world):
        st, bound, sc, rc, seg, big = _plan(tables, world, nb, me)
        assert st == 0 and bound.tolist() == exp_bound
        for r in range(world):
            assert sc[r] == counts[me, exp_bound[r]:exp_bound[r + 1]].sum()
            assert rc[r] == counts[r, exp_bound[me]:exp_bound[me + 1]].sum()
            # the segment table of source r: inside the owned range the running count of the received keys, flat outside
            lo, hi = exp_bound[me], exp_bound[me + 1]
            e = np.zeros(nb + 1, np.uint64)
            e[lo:hi + 1] = np.concatenate([[0], np.cumsum(counts[r, lo:hi])])
            e[hi + 1:] = e[hi]
            assert np.array_equal(seg[r], e)
        sent_total[me] = sc
        assert big == max(counts[x, exp_bound[y]:exp_bound[y + 1]].sum() for x in range(world) for y in range(world))
    # what rank x sends to y is what y expects from x, and nothing is lost
    for me in range(world):
        _, _, _, rc, _, _ = _plan(tables, world, nb, me)
        assert np.array_equal(rc, sent_total[:, me])
    assert sent_total.sum() == counts.sum()


def test_exchange_plan_more_ranks_than_buckets():
    """4 first digits over 8 ranks: half of the ranks own an empty digit range and receive nothing"""
    world, nb = 8, 4
    counts = np.arange(1, world * nb + 1, dtype=np.uint64).reshape(world, nb)
    tables = np.concatenate([np.zeros((world, 1), np.uint64), np.cumsum(counts, axis=1)], axis=1)
    owners = 0
    for me in range(world):
        st, bound, sc, rc, seg, _ = _plan(tables, world, nb, me)
        assert st == 0
        if bound[me] == bound[me + 1]:
            assert rc.sum() == 0 and seg.sum() == 0
        else:
            owners += 1
            assert rc.sum() == counts[:, bound[me]:bound[me + 1]].sum()
        assert sc.sum() == counts[me].sum()
    assert owners == nb


def test_exchange_plan_rejects_bad_tables():
    st = _plan([[0, 3, 2]], 1, 2, 0)[0]            # decreasing
    assert st == _lib.BRX_ERR_ARG
    st = _plan([[1, 3, 4]], 1, 2, 0)[0]            # does not start at 0
    assert st == _lib.BRX_ERR_ARG
