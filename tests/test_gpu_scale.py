"""Full-size (BASELINE.json configs[1]: 1e5 synthetic 10 kb reads, k=19) checks of the HIP path through
size-independent properties, where the oracle cannot run on everything:
  - the two build strategies (dense u8 table vs partitioned keys) give the same 16 GiB bitset, and
    counting in several batches equals counting in one;
  - reads whose every k-mer is solid come back unchanged (the reference's "no over-correction" asserts);
  - batch-splitting does not change corrected reads (reads are independent units);
  - a sample of reads is compared byte for byte with the oracle, using the set exported from HBM.
"""
import numpy as np
import pytest

import br_amd
from br_amd import _lib, synth
from br_amd import dist as bd
from oracle import oracle as O

pytestmark = pytest.mark.gpu

K, ABUNDANCE, N_READS, READ_LEN = 19, 3, 100_000, 10_000


@pytest.fixture(scope="module")
def job():
    import torch
    cfg = synth.config(genome_len=N_READS * READ_LEN // 50, read_len=READ_LEN)
    stream = torch.cuda.current_stream().cuda_stream
    dg = torch.empty(cfg.genome_len, dtype=torch.uint8, device="cuda")
    synth.genome_device(cfg, 0, dg.data_ptr(), stream)
    cap = int(N_READS * READ_LEN * 1.03) + (1 << 20)
    db = torch.empty(cap, dtype=torch.uint8, device="cuda")
    do = torch.empty(N_READS + 1, dtype=torch.int64, device="cuda")
    total = synth.reads_device(cfg, 0, dg.data_ptr(), 0, N_READS, db.data_ptr(), cap, do.data_ptr(), stream)
    cnt = br_amd.Counter(K, 0, _lib.COUNT_SORTED)
    cnt.add_batch_device(db.data_ptr(), do.data_ptr(), N_READS, total, stream)
    solid = cnt.finish(ABUNDANCE, stream)
    del cnt
    return {"cfg": cfg, "bases": db, "offsets": do, "total": total, "solid": solid, "stream": stream, "genome": dg}


def _bits_tensor(s):
    import torch
    ptr, n = s.device_bits()
    return bd.device_view(ptr, n).view(torch.int64)


def test_build_strategies_and_batching_agree(job):
    import torch
    stream = job["stream"]
    ref = _bits_tensor(job["solid"])
    n_solid = job["solid"].popcount()
    assert 15_000_000 < n_solid < 40_000_000          # ~ genome size (20 Mbp) + a few solid error k-mers

    # dense: the reference's own data structure (2^37-byte u8 table)
    cnt = br_amd.Counter(K, 0, _lib.COUNT_DENSE)
    cnt.add_batch_device(job["bases"].data_ptr(), job["offsets"].data_ptr(), N_READS, job["total"], stream)
    dense = cnt.finish(ABUNDANCE, stream)
    del cnt
    torch.cuda.synchronize()
    assert dense.popcount() == n_solid
    assert torch.equal(_bits_tensor(dense), ref)
    del dense

    # partitioned, three unequal batches (exercises the per-bucket merge of batches)
    off_h = job["offsets"].cpu()
    cuts = [0, 17_000, 60_001, N_READS]
    cnt = br_amd.Counter(K, 0, _lib.COUNT_SORTED)
    keep = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        sub_off = (job["offsets"][a:b + 1]).contiguous()
        keep.append(sub_off)
        nb = int(off_h[b] - off_h[a])
        cnt.add_batch_device(job["bases"].data_ptr(), sub_off.data_ptr(), b - a, nb, stream)
    merged = cnt.finish(ABUNDANCE, stream)
    torch.cuda.synchronize()
    assert torch.equal(_bits_tensor(merged), ref)


def test_all_solid_reads_come_back_unchanged(job):
    """abundance 0 makes every k-mer of the reads solid: nothing can trigger, the forward scan is the
    identity (the `assert_eq!(refe, corrector.correct(refe))` half of every reference unit test)."""
    import torch
    stream = job["stream"]
    n = 30_000
    off = job["offsets"][:n + 1].contiguous()
    nb = int(off[-1].item())
    cnt = br_amd.Counter(K, 0, _lib.COUNT_SORTED)
    cnt.add_batch_device(job["bases"].data_ptr(), off.data_ptr(), n, nb, stream)
    every = cnt.finish(0, stream)
    del cnt
    d_out = torch.empty(nb + (1 << 20), dtype=torch.uint8, device="cuda")
    d_oo = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    for method in ("one", "graph", "gap_size", "two"):
        chain = br_amd.Chain(every, [(method, 5, 7)], two_side=True)
        tot = chain.correct_batch_device(job["bases"].data_ptr(), off.data_ptr(), n, nb, d_out.data_ptr(), d_out.numel(),
                                         d_oo.data_ptr(), stream)
        assert tot == nb
        assert torch.equal(d_oo, off)
        assert torch.equal(d_out[:nb], job["bases"][:nb])
        assert chain.last_stats()["triggers"] == 0


def test_batch_split_invariance(job):
    """reads are independent units: correcting [A|B] in one call == correcting A and B separately."""
    import torch
    stream = job["stream"]
    n = 40_000
    off_h = job["offsets"][:n + 1].cpu()
    nb = int(off_h[-1])
    chain = br_amd.Chain(job["solid"], [("one", 5, 7)], two_side=False)
    out_all = torch.empty(int(nb * 1.05) + (1 << 20), dtype=torch.uint8, device="cuda")
    oo_all = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    off = job["offsets"][:n + 1].contiguous()
    tot_all = chain.correct_batch_device(job["bases"].data_ptr(), off.data_ptr(), n, nb, out_all.data_ptr(), out_all.numel(),
                                         oo_all.data_ptr(), stream)
    pieces = []
    for a, b in ((0, 12_345), (12_345, n)):
        sub = job["offsets"][a:b + 1].contiguous()
        snb = int(off_h[b] - off_h[a])
        o = torch.empty(int(snb * 1.05) + (1 << 20), dtype=torch.uint8, device="cuda")
        oo = torch.empty(b - a + 1, dtype=torch.int64, device="cuda")
        t = chain.correct_batch_device(job["bases"].data_ptr(), sub.data_ptr(), b - a, snb, o.data_ptr(), o.numel(),
                                       oo.data_ptr(), stream)
        pieces.append(o[:t])
    assert torch.equal(torch.cat(pieces), out_all[:tot_all])


def test_sample_of_reads_matches_oracle(job):
    """byte parity on a sample, with the very set the GPU built (exported from HBM, 16 GiB)."""
    import torch
    stream = job["stream"]
    bits = job["solid"].export_bits()
    osolid = O.Solid.wrap(K, bits)
    d_out = torch.empty(int(job["total"] * 1.05) + (1 << 20), dtype=torch.uint8, device="cuda")
    d_oo = torch.empty(N_READS + 1, dtype=torch.int64, device="cuda")
    rng = np.random.default_rng(5)
    sample = sorted(set(rng.integers(0, N_READS, size=60).tolist()) | {0, N_READS - 1})
    off_h = job["offsets"].cpu().numpy()
    for methods in (["one"], ["one", "graph"], ["two"], ["greedy"], ["gap_size"]):
        chain = br_amd.Chain(job["solid"], [(m, 5, 7) for m in methods], two_side=False)
        tot = chain.correct_batch_device(job["bases"].data_ptr(), job["offsets"].data_ptr(), N_READS, job["total"],
                                         d_out.data_ptr(), d_out.numel(), d_oo.data_ptr(), stream)
        oo_h = d_oo.cpu().numpy()
        assert int(oo_h[-1]) == tot
        om = O.build_methods(osolid, methods, 5, 7)
        for r in sample:
            src = job["bases"][int(off_h[r]):int(off_h[r + 1])].cpu().numpy().tobytes()
            got = d_out[int(oo_h[r]):int(oo_h[r + 1])].cpu().numpy().tobytes()
            assert got == O.correct_record(om, src, False), (methods, r)
        st = chain.last_stats()
        assert st["fixes"] > (10_000_000 if methods[0] in ("one", "gap_size") else 500_000)


def _threads():
    import os
    try:
        return max(1, min(len(os.sched_getaffinity(0)), 16))
    except AttributeError:
        return 8


def test_config1_every_read_vs_oracle(job):
    """BASELINE configs[1] as bench.py runs it -- One, forward + reverse, all 100 000 reads -- compared with the oracle
    READ BY READ (16 s on 16 host threads; `O.correct_batch_mt_check` compares inside its workers).  The lane form's
    correctness hangs on rare events (a missed sync point is a handful per million units, a read handed back one in
    100 000) that a sample of 62 reads never meets at this size.  And brx_chain_last_stats' `fixes` are the committed
    ones: equal to the oracle's count (speculative units and stretches scanned twice used to be counted too)."""
    import torch
    stream = job["stream"]
    osolid = O.Solid.wrap(K, job["solid"].export_bits())
    d_out = torch.empty(int(job["total"] * 1.05) + (1 << 20), dtype=torch.uint8, device="cuda")
    d_oo = torch.empty(N_READS + 1, dtype=torch.int64, device="cuda")
    chain = br_amd.Chain(job["solid"], [("one", 5, 7)], two_side=False)
    tot = chain.correct_batch_device(job["bases"].data_ptr(), job["offsets"].data_ptr(), N_READS, job["total"],
                                     d_out.data_ptr(), d_out.numel(), d_oo.data_ptr(), stream)
    st = chain.last_stats()
    assert st["lane_units"] > 0, "the forward pass did not run in lane form"
    assert st["lane_unwritten_units"] == 0
    bases = job["bases"][:job["total"]].cpu().numpy()
    offs = job["offsets"].cpu().numpy().astype(np.uint64)
    got, got_off = d_out[:tot].cpu().numpy(), d_oo.cpu().numpy().astype(np.uint64)
    _, want_total, want_fixes, n_bad, first_bad = O.correct_batch_mt_check(osolid, ["one"], bases, offs, got, got_off, 5, 7,
                                                                           False, _threads())
    assert n_bad == 0, f"{n_bad} of {N_READS} reads differ from the oracle, first: read {first_bad}"
    assert want_total == tot
    assert st["fixes"] == want_fixes, (st["fixes"], want_fixes)


@pytest.mark.parametrize("method", ["graph", "gap_size"])
def test_config1_walking_lane_forms_10000_reads_vs_oracle(job, method):
    """the lane forms of the walking correctors (solidity mask on, the default) on 10 000 full-size reads, every read
    against the oracle, forward + reverse; committed fixes equal the oracle's count"""
    import torch
    stream = job["stream"]
    n = 10_000
    osolid = O.Solid.wrap(K, job["solid"].export_bits())
    off = job["offsets"][:n + 1].contiguous()
    nb = int(off[-1].item())
    d_out = torch.empty(int(nb * 1.05) + (1 << 20), dtype=torch.uint8, device="cuda")
    d_oo = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    chain = br_amd.Chain(job["solid"], [(method, 5, 7)], two_side=False)
    tot = chain.correct_batch_device(job["bases"].data_ptr(), off.data_ptr(), n, nb, d_out.data_ptr(), d_out.numel(),
                                     d_oo.data_ptr(), stream)
    st = chain.last_stats()
    assert st["lane_units"] > 0, "the forward pass did not run in lane form"
    assert st["lane_unwritten_units"] == 0
    bases = job["bases"][:nb].cpu().numpy()
    offs = off.cpu().numpy().astype(np.uint64)
    got, got_off = d_out[:tot].cpu().numpy(), d_oo.cpu().numpy().astype(np.uint64)
    _, want_total, want_fixes, n_bad, first_bad = O.correct_batch_mt_check(osolid, [method], bases, offs, got, got_off, 5, 7,
                                                                           False, _threads())
    assert n_bad == 0, f"{method}: {n_bad} of {n} reads differ from the oracle, first: read {first_bad}"
    assert want_total == tot
    assert st["fixes"] == want_fixes, (st["fixes"], want_fixes)


def test_more_than_2_31_kmers_in_one_counter():
    """2.3 Gbp in one batch: key offsets beyond 2^31 (a sign-extended lane read once sent the final counting
    pass into an endless loop there).  Property check instead of the oracle: at 50x coverage nearly every
    k-mer of the genome is solid, a random 19-mer is not."""
    import torch
    n_reads = 230_000
    cfg = synth.config(genome_len=n_reads * READ_LEN // 50, read_len=READ_LEN)
    stream = torch.cuda.current_stream().cuda_stream
    dg = torch.empty(cfg.genome_len, dtype=torch.uint8, device="cuda")
    synth.genome_device(cfg, 0, dg.data_ptr(), stream)
    cap = int(n_reads * READ_LEN * 1.03) + (1 << 20)
    db = torch.empty(cap, dtype=torch.uint8, device="cuda")
    do = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")
    total = synth.reads_device(cfg, 0, dg.data_ptr(), 0, n_reads, db.data_ptr(), cap, do.data_ptr(), stream)
    assert total > (1 << 31)
    cnt = br_amd.Counter(K, 0, _lib.COUNT_SORTED)
    cnt.add_batch_device(db.data_ptr(), do.data_ptr(), n_reads, total, stream)
    solid = cnt.finish(ABUNDANCE, stream)
    del cnt, db
    n_solid = solid.popcount()
    assert 0.95 * cfg.genome_len < n_solid < 1.2 * cfg.genome_len
    g = dg[:200_000].cpu().numpy()
    code = (g >> 1) & 3
    kmers = np.zeros(len(g) - K + 1, dtype=np.uint64)
    for j in range(K):
        kmers = (kmers << np.uint64(2)) | code[j:j + len(kmers)].astype(np.uint64)
    assert solid.get_many(kmers).mean() > 0.98  # (the first 10 kb of the genome are covered thinly)
    rnd = np.random.default_rng(3).integers(0, 1 << (2 * K), size=100_000, dtype=np.uint64)
    assert solid.get_many(rnd).mean() < 0.01
    kl = solid.keylist_device(stream)
    assert kl is not None and kl[1] == n_solid


@pytest.mark.parametrize("hooks", [
    {},
    {"BRX_HF_RATIO": "0.01"},                                            # "hardly any key is distinct": every table starts too small
    {"BRX_HF_LOG_T": "10", "BRX_HF_MIN_LT": "6"},                        # 1024-slot tables: most buckets need passes
    {"BRX_HF_RATIO": "0.01", "BRX_HF_LOG_T": "8", "BRX_HF_MIN_LT": "4"},  # 256 slots and a wrong guess: the whole ladder
], ids=["default", "ratio_0.01", "tables_1024", "tables_256_ratio_0.01"])
def test_hash_final_skewed_buckets(hooks, monkeypatch):
    """(hooks: the table-sizing test switches of part_finish_impl, read per call -- a wrong guess of the share of distinct
    keys or a small table must only cost passes.)  The LDS hash-count of the partitioned finish on hand-made keys (fed through the exchange entry
    add_partitioned_device): one level-2 bucket with 40 000 DISTINCT hashes that all fall into the first quarter of
    its key range (more than the table holds: the pass is redone on finer ranges, and the solid hashes outnumber the
    workgroup's list buffer), a bucket with one hash repeated 300 000 times (many passes, a saturated count), and a
    sprinkle of ordinary keys.  Checked against the counts computed on the host."""
    import torch
    for name, val in hooks.items():
        monkeypatch.setenv(name, val)
    k = 19
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(11)
    nb1 = 512  # first digit: 9 bits; level-1 keys: 28 bits = 8-bit second digit | 20-bit rest
    def key32(d2, rest):
        return (np.uint32(d2) << np.uint32(20)) | rest.astype(np.uint32)
    parts = {}
    dense = np.arange(40_000, dtype=np.uint32)                      # rest < 2^18: all in the first of 4 ranges
    parts[5] = np.concatenate([key32(7, dense), key32(7, dense[:1000]), key32(7, dense[:1000]),   # 1000 of them 3x
                               key32(9, np.full(300_000, 123, dtype=np.uint32))])
    parts[300] = np.concatenate([key32(d, rng.integers(0, 1 << 20, 2000)) for d in (0, 255)] * 2)  # every key twice
    keys, l1off = [], np.zeros(nb1 + 1, dtype=np.int64)
    for b in range(nb1):
        ks = parts.get(b, np.zeros(0, dtype=np.uint32))
        keys.append(rng.permutation(ks))
        l1off[b + 1] = l1off[b] + len(ks)
    keys = np.concatenate(keys)
    want = {}
    for b, ks in parts.items():
        for v in ks.tolist():
            h = (b << 28) | v
            want[h] = want.get(h, 0) + 1
    dk = torch.from_numpy(keys.view(np.int32).copy()).cuda()
    do = torch.from_numpy(l1off).cuda()
    # a second key set: the same special buckets under a background of 92 M single (a few double) random keys in the
    # other first digits, which lifts the average bucket over the threshold where one workgroup takes a bucket
    per = 181_000
    g = torch.Generator(device="cuda").manual_seed(3)
    segs, off_bg, pos = [], np.zeros(nb1 + 1, dtype=np.int64), 0
    for b in range(nb1):
        if b in parts:
            seg = dk[int(l1off[b]):int(l1off[b + 1])]
        else:
            seg = torch.randint(0, 1 << 28, (per,), generator=g, device="cuda", dtype=torch.int32)
        segs.append(seg)
        pos += seg.numel()
        off_bg[b + 1] = pos
    dk_bg, do_bg = torch.cat(segs), torch.from_numpy(off_bg).cuda()
    for abundance, with_bg in ((0, False), (2, False), (200, False), (2, True), (200, True)):
        cnt = br_amd.Counter(k, 0, _lib.COUNT_SORTED)
        if with_bg:
            cnt.add_partitioned_device(dk_bg.data_ptr(), do_bg.data_ptr(), dk_bg.numel())
        else:
            cnt.add_partitioned_device(dk.data_ptr(), do.data_ptr(), dk.numel())
        gs = br_amd.Pcon.new(k)
        cnt.finish_into(abundance, gs, stream)
        torch.cuda.synchronize()
        assert gs.bits_state() == 1
        solid = sorted(h for h, c in want.items() if min(c, 255) > abundance)
        kl = gs.keylist_device(stream)
        assert kl is not None
        if with_bg:  # background keys seen 3+ times are solid too at abundance 2 (a handful): compare the special buckets
            allk = bd.device_view(kl[0], max(kl[1], 1) * 8).view(torch.int64)[:kl[1]].cpu().numpy()
            got = allk[np.isin(allk >> 28, list(parts))]
            assert np.array_equal(np.sort(got), np.array(solid, dtype=np.int64))
            assert kl[1] - len(got) < (200 if abundance == 2 else 1)
            continue
        assert kl[1] == len(solid)
        got = bd.device_view(kl[0], max(kl[1], 1) * 8).view(torch.int64)[:kl[1]].cpu().numpy()
        assert np.array_equal(np.sort(got), np.array(solid, dtype=np.int64))
        assert gs.popcount() == len(solid)
