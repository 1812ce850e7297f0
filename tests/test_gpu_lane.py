"""correct::One's forward pass in its lane-per-chunk form (br_amd/csrc/brx_onelane.hip): a read is cut into units at
sync points (positions behind R solid original k-mers in a row), every unit is scanned by one lane from the predicted
state (i, original k-mer), the predecessor checks the prediction when it gets there, misses scan on into a second
staging area, two misses hand the read back to the group kernel, and a stitch kernel joins the pieces.  Whatever the
chunk length and the sync rule, the bytes are the sequential scan's (src/correct/mod.rs:53-107), i.e. the oracle's."""
import os

import numpy as np
import pytest

import br_amd
from br_amd import _lib, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture
def lane_env(monkeypatch):
    def set_(chunk=None, sync=None, lane=None, mask=None):
        for key, v in (("BRX_LANE_CHUNK", chunk), ("BRX_LANE_SYNC", sync), ("BRX_LANE", lane), ("BRX_LANE_MASK", mask)):
            if v is None:
                monkeypatch.delenv(key, raising=False)
            else:
                monkeypatch.setenv(key, str(v))
    return set_


@pytest.mark.parametrize("chunk,sync", [(64, 1), (64, 4), (100, 2), (256, 4), (333, 8), (2048, 4), (None, None)])
def test_fixture_reads_any_chunking(raw_reads, solid_fixture_bytes, lane_env, chunk, sync):
    """the reference's fixture set (k = 11) on 80 reads of raw.fasta (3 - 62 kb), One alone and in front of Graph,
    forward only and forward + reverse: byte parity for every chunk length / sync rule, and the units really ran"""
    lane_env(chunk, sync)
    reads = raw_reads[:80]
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    ref = O.Solid.from_bytes(solid_fixture_bytes)
    for names, two_side in ((["one"], True), (["one"], False), (["one", "graph"], False), (["two", "one"], True)):
        chain = br_amd.Chain(gs, [(m, 5, 7) for m in names], two_side=two_side)
        got = chain.correct_reads(reads)
        st = chain.last_stats()
        om = O.build_methods(ref, names, 5, 7)
        for r, g in zip(reads, got):
            assert g == O.correct_record(om, r, two_side)
        assert st["lane_units"] >= len(reads) and st["lane_unwritten_units"] == 0
        if chunk is not None and chunk <= 256:
            assert st["lane_units"] > 20 * len(reads)
        assert st["fixes"] > 0


def test_lane_off_is_the_group_kernel(raw_reads, solid_fixture_bytes, lane_env):
    lane_env(lane=0)
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    chain = br_amd.Chain(gs, [("one", 5, 7)], two_side=False)
    a = chain.correct_reads(raw_reads[:30])
    assert chain.last_stats()["lane_units"] == 0
    lane_env()
    chain2 = br_amd.Chain(gs, [("one", 5, 7)], two_side=False)
    b = chain2.correct_reads(raw_reads[:30])
    assert chain2.last_stats()["lane_units"] > 0 and a == b


@pytest.mark.parametrize("mask", [None, 2])
@pytest.mark.parametrize("k,confirm", [(13, 5), (15, 2), (19, 5), (19, 0), (21, 3), (25, 1)])
def test_synthetic_reads_vs_oracle(lane_env, k, confirm, mask):
    """synthetic ONT-error reads (the bench's error model) against a counted set: dense / lazy / sparse holdings of the set
    behind the same automaton, confirm 0 ... 5, short reads, reads shorter than k, an empty read; mask = 2: One too scans
    the stretches without a fix off the solidity mask of the original k-mers (BRX_LANE_MASK, off for One by default)"""
    lane_env(128, 3, mask=mask)
    cfg = synth.config(genome_len=40_000, read_len=3_000)
    g = synth.genome_host(cfg)
    bases, offs = synth.reads_host(cfg, g, 0, 500)
    reads = [bases[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(500)]
    reads += [b"", b"ACGT", reads[0][:k - 1], reads[1][:k], reads[2][:k + 1], reads[3][:k + 7], reads[4][:130]]
    if k <= 21:
        gs = br_amd.Pcon.from_count(reads, k, 2)
        ref = O.Solid.from_count(k, O.count_reads(k, reads), 2) if k <= 15 else O.Solid.sparse_from_count(k, reads, 2)
    else:
        gs = br_amd.Pcon.from_fasta(reads, k)
        ref = O.Solid.sparse_from_count(k, reads, 0)
    chain = br_amd.Chain(gs, [("one", confirm, 7)], two_side=False)
    got = chain.correct_reads(reads)
    om = O.build_methods(ref, ["one"], confirm, 7)
    bad = [i for i, (r, x) in enumerate(zip(reads, got)) if x != O.correct_record(om, r, False)]
    assert not bad, bad[:10]
    st = chain.last_stats()
    if confirm == 0:            # -C 0: every scenario scores 0 == c; that corner stays with the group kernel
        assert st["lane_units"] == 0
    else:
        assert st["lane_units"] > 5_000 and st["lane_redone_reads"] < 50


def test_confirm_beyond_the_window_takes_the_group_kernel(raw_reads, solid_fixture_bytes, lane_env):
    """-C 9: the look-aheads do not fit the 8-base window of a lane's round, so the pass falls back as a whole"""
    lane_env(100, 2)
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    ref = O.Solid.from_bytes(solid_fixture_bytes)
    chain = br_amd.Chain(gs, [("one", 9, 7)], two_side=True)
    got = chain.correct_reads(raw_reads[:20])
    om = O.build_methods(ref, ["one"], 9, 7)
    assert got == [O.correct_record(om, r, True) for r in raw_reads[:20]]
    assert chain.last_stats()["lane_units"] == 0


def test_growing_reads_and_slot_overflow(lane_env):
    """a set in which every trigger resolves as a deletion makes reads GROW (one.rs:61: offset 0): units outgrow their
    staging regions and reads their slots; both go back through the group kernel / the redo with more slack"""
    lane_env(64, 1)
    rng = np.random.default_rng(5)
    k = 9
    genome = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 4000).tobytes()
    reads = []
    for _ in range(60):
        s = int(rng.integers(0, 3000))
        r = bytearray(genome[s:s + 900])
        for pos in sorted(rng.integers(20, len(r) - 20, size=60).tolist(), reverse=True):
            del r[pos]                                    # deletions only: the corrector puts the bases back
        reads.append(bytes(r))
    gs = br_amd.Pcon.from_fasta([genome], k)
    ref = O.Solid(k)
    ref.set_seq(genome)
    chain = br_amd.Chain(gs, [("one", 2, 7)], two_side=True)
    got = chain.correct_reads(reads)
    om = O.build_methods(ref, ["one"], 2, 7)
    assert got == [O.correct_record(om, r, True) for r in reads]
    assert sum(len(x) for x in got) > sum(len(r) for r in reads)


# ---------------------------------------------------------------- Graph / GapSize in lane form ----------------
@pytest.mark.parametrize("chunk,sync,mask", [(64, 1, None), (100, 4, 0), (333, 2, 2), (None, None, None), (None, None, 0)])
@pytest.mark.parametrize("names", [["graph"], ["gap_size"], ["one", "graph", "gap_size"]])
def test_walking_correctors_any_chunking(raw_reads, solid_fixture_bytes, lane_env, chunk, sync, mask, names):
    """correct::Graph / correct::GapSize forward passes as lane automata (error_len, alt_nucs, the unique-successor walk
    with Brent's detector / the exact visited rule, GapSize's three-way dispatch): raw.fasta against the reference's
    fixture set, forward only and forward + reverse, any chunk length / sync rule; with the solidity mask of the original
    k-mers (the default for these: clean stretches and error_len are read off it), without (mask = 0), and for One too (2)"""
    lane_env(chunk, sync, mask=mask)
    reads = raw_reads[:50]
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    ref = O.Solid.from_bytes(solid_fixture_bytes)
    om = O.build_methods(ref, names, 5, 7)
    for two_side in (True, False):
        chain = br_amd.Chain(gs, [(m, 5, 7) for m in names], two_side=two_side)
        got = chain.correct_reads(reads)
        bad = [i for i, (r, x) in enumerate(zip(reads, got)) if x != O.correct_record(om, r, two_side)]
        assert not bad, bad[:10]
        st = chain.last_stats()
        assert st["lane_units"] >= len(reads) and st["fixes"] > 0


def test_walk_lane_off_is_the_group_kernel(raw_reads, solid_fixture_bytes, monkeypatch):
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    monkeypatch.setenv("BRX_LANE_WALK", "0")
    chain = br_amd.Chain(gs, [("graph", 5, 7), ("gap_size", 5, 7)], two_side=False)
    a = chain.correct_reads(raw_reads[:25])
    assert chain.last_stats()["lane_units"] == 0
    monkeypatch.delenv("BRX_LANE_WALK")
    chain2 = br_amd.Chain(gs, [("graph", 5, 7), ("gap_size", 5, 7)], two_side=False)
    b = chain2.correct_reads(raw_reads[:25])
    assert chain2.last_stats()["lane_units"] > 0 and a == b


@pytest.mark.parametrize("mask", [None, 0])
@pytest.mark.parametrize("k,method", [(13, "graph"), (13, "gap_size"), (19, "graph"), (19, "gap_size"), (21, "graph"), (21, "gap_size")])
def test_walking_correctors_synthetic(lane_env, k, method, mask):
    """the bench's error model against counted sets: bit vector (k = 13), lazy bits + index (19), sparse chained index (21);
    with the solidity mask (default) and without"""
    lane_env(128, 3, mask=mask)
    cfg = synth.config(genome_len=40_000, read_len=3_000)
    g = synth.genome_host(cfg)
    bases, offs = synth.reads_host(cfg, g, 0, 400)
    reads = [bases[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(400)]
    reads += [b"", b"ACGT", reads[0][:k - 1], reads[1][:k], reads[2][:k + 1], reads[3][:k + 7], reads[4][:130]]
    gs = br_amd.Pcon.from_count(reads, k, 2)
    ref = O.Solid.from_count(k, O.count_reads(k, reads), 2) if k <= 15 else O.Solid.sparse_from_count(k, reads, 2)
    chain = br_amd.Chain(gs, [(method, 5, 7)], two_side=False)
    got = chain.correct_reads(reads)
    om = O.build_methods(ref, [method], 5, 7)
    bad = [i for i, (r, x) in enumerate(zip(reads, got)) if x != O.correct_record(om, r, False)]
    assert not bad, bad[:10]
    st = chain.last_stats()
    assert st["lane_units"] > 4_000 and st["fixes"] > 1_000 and st["lane_redone_reads"] < 40


def test_gap_size_walks_over_repeats(lane_env):
    """GapSize's fixed-length walk returns None when it meets a k-mer twice (gap_size.rs:75-81).  The lanes do not look;
    the replay kernel does, for every such fix -- tandem repeats longer than k make walks go round in circles here."""
    lane_env(64, 1)
    rng = np.random.default_rng(11)
    k = 9
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    unit = rng.choice(alpha, 14).tobytes()
    genome = rng.choice(alpha, 300).tobytes() + unit * 30 + rng.choice(alpha, 300).tobytes() + unit * 12 + rng.choice(alpha, 200).tobytes()
    reads = []
    for _ in range(300):
        s = int(rng.integers(0, len(genome) - 400))
        r = bytearray(genome[s:s + 400])
        for pos in sorted(rng.integers(15, len(r) - 15, size=int(rng.integers(2, 14))).tolist(), reverse=True):
            x = rng.random()
            if x < 0.4:
                del r[pos]
            elif x < 0.7:
                r.insert(pos, int(alpha[rng.integers(0, 4)]))
            else:
                r[pos] = int(alpha[rng.integers(0, 4)])
        reads.append(bytes(r))
    gs = br_amd.Pcon.from_fasta([genome], k)
    ref = O.Solid(k)
    ref.set_seq(genome)
    for names in (["gap_size"], ["graph"], ["gap_size", "graph"]):
        chain = br_amd.Chain(gs, [(m, 2, 7) for m in names], two_side=True)
        got = chain.correct_reads(reads)
        om = O.build_methods(ref, names, 2, 7)
        bad = [i for i, (r, x) in enumerate(zip(reads, got)) if x != O.correct_record(om, r, True)]
        assert not bad, (names, bad[:10])
        assert chain.last_stats()["lane_units"] > 0


def test_two_chains_on_two_threads_share_a_fresh_successor_table(lane_env):
    """The successor table of the walking lane forms belongs to the SET and is built by whichever chain walks first; a
    second chain on the same set -- its own stream, another host thread (include/brx.h: only calls on ONE chain are
    serialised) -- must not read it half written.  Two threads start Graph / GapSize chains on a freshly indexed set at
    the same moment, several times over fresh sets; every read equals the oracle's."""
    import threading
    lane_env(128, 2)
    k = 19
    cfg = synth.config(genome_len=60_000, read_len=4_000)
    g = synth.genome_host(cfg)
    bases, offs = synth.reads_host(cfg, g, 0, 600)
    reads = [bases[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(600)]
    ref = O.Solid.sparse_from_count(k, reads, 2)
    want = {m: [O.correct_record(O.build_methods(ref, [m], 5, 7), r, True) for r in reads] for m in ("graph", "gap_size")}
    for rep in range(4):
        gs = br_amd.Pcon.from_count(reads, k, 2)  # a fresh set: no successor table yet
        chains = [br_amd.Chain(gs, [(m, 5, 7)], two_side=True) for m in ("graph", "gap_size")]
        out, errs = [None, None], []
        gate = threading.Barrier(2)

        def work(j):
            try:
                gate.wait()
                out[j] = chains[j].correct_reads(reads)
            except Exception as e:  # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work, args=(j,)) for j in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for j, m in enumerate(("graph", "gap_size")):
            assert chains[j].last_stats()["lane_units"] > 0
            bad = [i for i in range(len(reads)) if out[j][i] != want[m][i]]
            assert not bad, (rep, m, bad[:5])


@pytest.mark.parametrize("rev", ["1", "2", "3"])
def test_reverse_passes_in_lane_form(raw_reads, solid_fixture_bytes, lane_env, monkeypatch, rev):
    """BRX_LANE_REV (bit 0 Graph, bit 1 GapSize): the reverse passes -- reads stored back to front, never copied
    reversed (src/lib.rs:48-55) -- through the lane form as well: the packed copy and the replay read logical base j at
    in[n - 1 - j].  Fixture reads (3 - 62 kb, k = 11: a dense set, so the reversed reads do meet solid k-mers, triggers
    and walks) and synthetic ones at k = 19 against a counted set; chains whose second method sees staged, reversed
    input; every read equals the oracle's and the group-kernel form's."""
    lane_env(100, 2)
    monkeypatch.setenv("BRX_LANE_REV", rev)
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    ref = O.Solid.from_bytes(solid_fixture_bytes)
    reads = raw_reads[:40] + [b"", b"ACGTACGTAC", raw_reads[41][:11], raw_reads[42][:12], raw_reads[43][:27]]
    for names in (["graph"], ["gap_size"], ["graph", "gap_size"], ["one", "gap_size", "graph"]):
        chain = br_amd.Chain(gs, [(m, 5, 7) for m in names], two_side=False)
        got = chain.correct_reads(reads)
        assert chain.last_stats()["lane_unwritten_units"] == 0
        om = O.build_methods(ref, names, 5, 7)
        bad = [i for i, r in enumerate(reads) if got[i] != O.correct_record(om, r, False)]
        assert not bad, (names, bad[:5])
    k = 19
    cfg = synth.config(genome_len=50_000, read_len=3_000)
    g = synth.genome_host(cfg)
    bases, offs = synth.reads_host(cfg, g, 0, 400)
    sreads = [bases[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(400)]
    gs2 = br_amd.Pcon.from_count(sreads, k, 2)
    ref2 = O.Solid.sparse_from_count(k, sreads, 2)
    for names in (["graph"], ["gap_size"], ["gap_size", "graph"]):
        got = br_amd.Chain(gs2, [(m, 5, 7) for m in names], two_side=False).correct_reads(sreads)
        om = O.build_methods(ref2, names, 5, 7)
        bad = [i for i, r in enumerate(sreads) if got[i] != O.correct_record(om, r, False)]
        assert not bad, (names, bad[:5])
        monkeypatch.setenv("BRX_LANE_REV", "0")
        assert br_amd.Chain(gs2, [(m, 5, 7) for m in names], two_side=False).correct_reads(sreads) == got
        monkeypatch.setenv("BRX_LANE_REV", rev)


@pytest.mark.parametrize("lean", ["", "0"])
def test_lean_reverse_passes(raw_reads, solid_fixture_bytes, monkeypatch, lean):
    """The reverse passes of Two / Graph / Greedy / GapSize in lean form (rev_scan_kernel: scan, error_len and alt_nucs
    only; every read that would do anything else is handed back to the group kernel) against the oracle and against the
    64-lane group kernel (BRX_REV_LEAN=0).  Fixture reads at k = 11 -- a dense set, so the reversed reads meet solid
    k-mers and most reads are handed back -- and synthetic reads at k = 19 of which every third is stored back to front,
    so that the REVERSE pass is the one that finds them correctable; reads of length 0, < k, k, k + 1; chains, so the
    pass sees staged input that earlier passes have changed."""
    if lean:
        monkeypatch.setenv("BRX_REV_LEAN", lean)
    else:
        monkeypatch.delenv("BRX_REV_LEAN", raising=False)
    monkeypatch.delenv("BRX_GROUP_REV", raising=False)
    monkeypatch.delenv("BRX_GROUP", raising=False)
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    ref = O.Solid.from_bytes(solid_fixture_bytes)
    reads = raw_reads[:30] + [b"", b"ACGTACGTAC", raw_reads[41][:11], raw_reads[42][:12], raw_reads[43][:27]]
    for names in (["two"], ["graph"], ["greedy"], ["gap_size"], ["one", "two", "gap_size", "graph", "greedy"]):
        chain = br_amd.Chain(gs, [(m, 5, 7) for m in names], two_side=False)
        got = chain.correct_reads(reads)
        om = O.build_methods(ref, names, 5, 7)
        bad = [i for i, r in enumerate(reads) if got[i] != O.correct_record(om, r, False)]
        assert not bad, (names, bad[:5])
    k = 19
    cfg = synth.config(genome_len=50_000, read_len=3_000)
    g = synth.genome_host(cfg)
    bases, offs = synth.reads_host(cfg, g, 0, 300)
    sreads = [bases[int(offs[i]):int(offs[i + 1])].tobytes() for i in range(300)]
    gs2 = br_amd.Pcon.from_count(sreads, k, 2)
    ref2 = O.Solid.sparse_from_count(k, sreads, 2)
    mixed = [r[::-1] if i % 3 == 0 else r for i, r in enumerate(sreads)] + [sreads[0][:k], sreads[1][:k + 1][::-1], sreads[2][:k - 1]]
    for names, c in ((["two"], 2), (["graph"], 5), (["greedy"], 3), (["gap_size"], 5), (["gap_size", "graph", "two"], 4)):
        chain = br_amd.Chain(gs2, [(m, c, 7) for m in names], two_side=False)
        got = chain.correct_reads(mixed)
        om = O.build_methods(ref2, names, c, 7)
        want = [O.correct_record(om, r, False) for r in mixed]
        bad = [i for i, r in enumerate(mixed) if got[i] != want[i]]
        assert not bad, (names, bad[:5])
        # the reversed reads were corrected by the reverse pass (the test would prove little otherwise)
        assert any(want[i] != mixed[i] for i in range(0, 300, 3)), names


@pytest.mark.parametrize("walk_group", ["", "16"])
def test_verify_groups_stay_inside_the_visited_lists_of_a_small_batch(raw_reads, solid_fixture_bytes, monkeypatch, walk_group):
    """Fuzz seed 211 case 5889 / seed 7 case 226 (round 4): a batch of a few reads has the chain's minimum of visited
    lists, and the verify pass of the lean reverse form, launched with 4-lane groups (64 a block), indexed 64 lists where
    the chain had sized 32 -- with three-entry lists (BRX_MAXPATH=3) the groups overwrote each other's walks and reads
    came out wrong.  The grid is now cut from the lists the chain has (sized_path_lists, at least 64)."""
    monkeypatch.setenv("BRX_REV_VERIFY_G", "4")
    monkeypatch.setenv("BRX_MAXPATH", "3")
    monkeypatch.delenv("BRX_REV_LEAN", raising=False)
    monkeypatch.delenv("BRX_GROUP_REV", raising=False)
    monkeypatch.delenv("BRX_GROUP", raising=False)
    if walk_group:
        monkeypatch.setenv("BRX_GROUP_WALK", walk_group)
    else:
        monkeypatch.delenv("BRX_GROUP_WALK", raising=False)
    gs = br_amd.Pcon.from_pcon_solid(solid_fixture_bytes)
    ref = O.Solid.from_bytes(solid_fixture_bytes)
    reads = [r[:300] for r in raw_reads[:18]] + [raw_reads[18][:94], raw_reads[19][:1700]]
    for names, c in ((["gap_size", "graph", "graph"], 0), (["graph", "gap_size", "two"], 5)):
        got = br_amd.Chain(gs, [(m, c, 1) for m in names], two_side=False).correct_reads(reads)
        om = O.build_methods(ref, names, c, 1)
        bad = [i for i, r in enumerate(reads) if got[i] != O.correct_record(om, r, False)]
        assert not bad, (names, bad[:5])
