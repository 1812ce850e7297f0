"""The set the 8-GPU jobs of BASELINE.json actually correct against, on ONE card.  configs[3] / [4] replicate the union
of all ranks' solid k-mers into every GPU: ~0.8-1 G keys, 2^29 index lines (32 GiB), 16-mer minimizers -- eight times
the per-GPU share the other tests build.  Synthetic: 4 Gbp of 8x coverage of a 500 Mbp genome at `-a 1` (half of round 2's
measurement run, tools/bigset_bench.py, so that the suite stays in minutes; the index is the same 2^29 lines: it is
sized from 0.4 G keys up), then One at k = 19 and Graph + GapSize at k = 21 over a block of the reads:
size-independent properties (batch-split invariance, committed fixes plausible, lane forms ran or not as designed)
and a 64-read sample against the oracle holding the SAME set (bit vector at k = 19, every solid hash at k = 21).
Reference loop being sharded: src/lib.rs:72-139; the set every corrector borrows: src/lib.rs:141-147."""
import numpy as np
import pytest

import br_amd
from br_amd import _lib, synth
from br_amd import dist as bd
from oracle import oracle as O

pytestmark = pytest.mark.gpu

READ_LEN, COVERAGE, ABUNDANCE, N_READS = 10_000, 8, 1, 400_000


@pytest.fixture(scope="module")
def reads():
    import torch
    cfg = synth.config(genome_len=N_READS * READ_LEN // COVERAGE, read_len=READ_LEN)
    stream = torch.cuda.current_stream().cuda_stream
    dg = torch.empty(cfg.genome_len, dtype=torch.uint8, device="cuda")
    synth.genome_device(cfg, 0, dg.data_ptr(), stream)
    cap = int(N_READS * READ_LEN * 1.03) + (1 << 20)
    db = torch.empty(cap, dtype=torch.uint8, device="cuda")
    do = torch.empty(N_READS + 1, dtype=torch.int64, device="cuda")
    total = synth.reads_device(cfg, 0, dg.data_ptr(), 0, N_READS, db.data_ptr(), cap, do.data_ptr(), stream)
    del dg
    return {"cfg": cfg, "bases": db, "offsets": do, "total": total, "stream": stream, "off_h": do.cpu().numpy()}


def _correct(chain, rd, first, n):
    import torch
    off_h = rd["off_h"]
    nb = int(off_h[first + n] - off_h[first])
    sub = (rd["offsets"][first:first + n + 1] - rd["offsets"][first]).contiguous()
    out = torch.empty(int(nb * 1.08) + (1 << 20), dtype=torch.uint8, device="cuda")
    oo = torch.empty(n + 1, dtype=torch.int64, device="cuda")
    tot = chain.correct_batch_device(rd["bases"].data_ptr() + int(off_h[first]), sub.data_ptr(), n, nb, out.data_ptr(), out.numel(),
                                     oo.data_ptr(), rd["stream"])
    return out, oo.cpu().numpy(), tot


def _sample_vs_oracle(rd, om, out, oo_h, first, n, seed):
    rng = np.random.default_rng(seed)
    sample = sorted(set((first + rng.integers(0, n, size=62)).tolist()) | {first, first + n - 1})
    off_h, changed = rd["off_h"], 0
    for r in sample:
        src = rd["bases"][int(off_h[r]):int(off_h[r + 1])].cpu().numpy().tobytes()
        got = out[int(oo_h[r - first]):int(oo_h[r - first + 1])].cpu().numpy().tobytes()
        assert got == O.correct_record(om, src, False), r
        changed += got != src
    assert changed > len(sample) // 2


def test_bigset_k19_one_sample_vs_oracle(reads):
    import torch
    K = 19
    cnt = br_amd.Counter(K, 0, _lib.COUNT_SORTED)
    cnt.add_batch_device(reads["bases"].data_ptr(), reads["offsets"].data_ptr(), N_READS, reads["total"], reads["stream"])
    solid = cnt.finish(ABUNDANCE, reads["stream"])
    del cnt
    n_solid = solid.popcount()
    # 8x coverage, 5 % errors: a genome 19-mer is seen error-free ~3 times; seen twice or more: ~0.8 of them
    assert 0.6 * reads["cfg"].genome_len < n_solid < 1.1 * reads["cfg"].genome_len

    chain = br_amd.Chain(solid, [("one", 5, 7)], two_side=False)
    n = 200_000
    out, oo_h, tot = _correct(chain, reads, 0, n)
    info = solid.index_info()                       # (built by the first chain that needs it)
    assert info["valid"] and info["m"] == 16 and info["log2_lines"] == 29, info  # the 8-GPU jobs' table: 32 GiB
    st = chain.last_stats()
    assert int(oo_h[-1]) == tot and st["fixes"] > 10 * n and st["overflow_retries"] == 0 and st["lane_unwritten_units"] == 0
    # reads are independent units: the second half of the block on its own gives the same bytes
    out2, oo2_h, tot2 = _correct(chain, reads, n // 2, n // 2)
    assert tot2 == tot - int(oo_h[n // 2])
    assert torch.equal(out2[:tot2], out[int(oo_h[n // 2]):tot])
    del out2
    om = O.build_methods(O.Solid.wrap(K, solid.export_bits()), ["one"], 5, 7)
    _sample_vs_oracle(reads, om, out, oo_h, 0, n, 21)


def test_bigset_k21_graph_gap_size_sample_vs_oracle(reads):
    import torch
    K = 21
    cnt = br_amd.Counter(K, 0, _lib.COUNT_SORTED)
    cnt.add_batch_device(reads["bases"].data_ptr(), reads["offsets"].data_ptr(), N_READS, reads["total"], reads["stream"])
    solid = cnt.finish(ABUNDANCE, reads["stream"])
    del cnt
    assert solid.is_sparse()
    n_solid = solid.popcount()
    assert 0.5 * reads["cfg"].genome_len < n_solid < 1.1 * reads["cfg"].genome_len
    kl = solid.keylist_device(reads["stream"])           # before anything rebuilds the index from another list
    assert kl is not None and kl[1] == n_solid
    members = torch.sort(bd.device_view(kl[0], kl[1] * 8).view(torch.int64).clone()).values.cpu().numpy().view(np.uint64)
    assert np.all(members[1:] != members[:-1])

    methods = ["graph", "gap_size"]
    chain = br_amd.Chain(solid, [(m, 5, 7) for m in methods], two_side=False)
    n = 100_000
    out, oo_h, tot = _correct(chain, reads, 50_000, n)
    info = solid.index_info()
    assert info["valid"] and info["log2_lines"] == 29, info
    st = chain.last_stats()
    assert int(oo_h[-1]) == tot and st["fixes"] > 5 * n and st["lane_units"] > 0 and st["lane_unwritten_units"] == 0
    out2, oo2_h, tot2 = _correct(chain, reads, 50_000 + n // 2, n // 2)
    assert tot2 == tot - int(oo_h[n // 2])
    assert torch.equal(out2[:tot2], out[int(oo_h[n // 2]):tot])
    del out2
    osolid = O.Solid(K, _h=O.lib().bro_solid_new_sparse(K, members.ctypes.data, members.size))
    om = O.build_methods(osolid, methods, 5, 7)
    _sample_vs_oracle(reads, om, out, oo_h, 50_000, n, 22)
