#!/usr/bin/env python3
"""Every corrector on the bench data set (1e5 synthetic 10 kb reads, k=19, set built once): time of the
forward + reverse passes and the kernel's own counters.  Not the contract bench (bench.py is); this is
the per-method table of DESIGN.md.  usage: python tools/method_bench.py [reads=100000] [method ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import br_amd
from br_amd import _lib, synth

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
# METHOD_BENCH_BATCH=N: the set is built from all reads, the passes are timed on the first N only (what one
# 8192-record host batch costs on the device against the full-size set)
n_batch = int(os.environ.get("METHOD_BENCH_BATCH", "0")) or n_reads
methods = sys.argv[2:] or ["one", "two", "graph", "gap_size", "greedy"]
k, a, read_len = 19, 3, 10000
cfg = synth.config(genome_len=n_reads * read_len // 50, read_len=read_len)
stream = torch.cuda.current_stream().cuda_stream
dg = torch.empty(cfg.genome_len, dtype=torch.uint8, device="cuda")
synth.genome_device(cfg, 0, dg.data_ptr(), stream)
cap = int(n_reads * read_len * 1.03) + (1 << 20)
db = torch.empty(cap, dtype=torch.uint8, device="cuda")
do = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")
total = synth.reads_device(cfg, 0, dg.data_ptr(), 0, n_reads, db.data_ptr(), cap, do.data_ptr(), stream)
cnt = br_amd.Counter(k, 0, _lib.COUNT_SORTED)
cnt.add_batch_device(db.data_ptr(), do.data_ptr(), n_reads, total, stream)
gs = cnt.finish(a, stream)
del cnt
d_out = torch.empty(int(total * 1.1) + (1 << 20), dtype=torch.uint8, device="cuda")
d_oo = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")
if n_batch < n_reads:
    total = int(do[n_batch].item())
    n_reads = n_batch
_lib.profile_enable(True)
rows = []
for m in methods:
    # METHOD_BENCH_FWD_ONLY=1: the forward pass alone (two_side): the difference to the default run is the reverse pass
    chain = br_amd.Chain(gs, [(m, 5, 7)], two_side=os.environ.get("METHOD_BENCH_FWD_ONLY", "") == "1")
    best = None
    for rep in range(int(os.environ.get("METHOD_BENCH_REPS", "2"))):
        _lib.profile_reset()
        t0 = time.perf_counter()
        out_total = chain.correct_batch_device(db.data_ptr(), do.data_ptr(), n_reads, total, d_out.data_ptr(), d_out.numel(),
                                               d_oo.data_ptr(), stream)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        best = wall if best is None else min(best, wall)
    st = chain.last_stats()
    prof = {kk: round(v["total_ms"], 2) for kk, v in _lib.profile_all().items() if v["launches"]}
    row = {"kernels_ms": prof, "method": m, "wall_ms_fwd_rev": round(best, 2), "gbases_per_s": round(total / best / 1e6, 3), "out_bases": int(out_total),
           **{kk: int(v) for kk, v in st.items()}}
    rows.append(row)
    print(json.dumps(row), flush=True)
    del chain
