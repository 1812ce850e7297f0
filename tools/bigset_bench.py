#!/usr/bin/env python3
"""The correction passes against a set of the size the 8-GPU jobs REPLICATE into every GPU (BASELINE configs[3] / [4]:
the union of all ranks' k-mers, ~0.8-1 G solid k-mers, 2^29 index lines = 32 GiB), on one card: 8 Gbp of 8x coverage
of a 1 Gbp genome, `-a 1`, as DESIGN.md section 6 did in round 2.  One JSON line per method chain: wall time of
forward + reverse over all reads, the kernels' own timers, the set's index.
usage: python tools/bigset_bench.py K [n_reads=800000] [variant ...]
variant = chain[:fwd][@ENV=V[;ENV=V...]]   chain: one | graph,gap_size | ...; :fwd = forward pass only (two_side);
                                           @...: library switches set for this variant only (A/B runs on one set)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import br_amd
from br_amd import _lib, synth

k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 800_000
variants = sys.argv[3:] or (["one"] if k == 19 else ["graph,gap_size"])
coverage, a, read_len = 8, 1, 10_000
cfg = synth.config(genome_len=n_reads * read_len // coverage, read_len=read_len)
stream = torch.cuda.current_stream().cuda_stream
dg = torch.empty(cfg.genome_len, dtype=torch.uint8, device="cuda")
synth.genome_device(cfg, 0, dg.data_ptr(), stream)
cap = int(n_reads * read_len * 1.03) + (1 << 20)
db = torch.empty(cap, dtype=torch.uint8, device="cuda")
do = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")
total = synth.reads_device(cfg, 0, dg.data_ptr(), 0, n_reads, db.data_ptr(), cap, do.data_ptr(), stream)
del dg
torch.cuda.synchronize()
t0 = time.perf_counter()
cnt = br_amd.Counter(k, 0, _lib.COUNT_SORTED)
cnt.add_batch_device(db.data_ptr(), do.data_ptr(), n_reads, total, stream)
gs = cnt.finish(a, stream)
torch.cuda.synchronize()
build_ms = (time.perf_counter() - t0) * 1e3
del cnt
d_out = torch.empty(int(total * 1.08) + (1 << 20), dtype=torch.uint8, device="cuda")
d_oo = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")
_lib.profile_enable(True)
for variant in variants:
    spec, _, envs = variant.partition("@")
    fwd_only = spec.endswith(":fwd")
    names = spec[:-4].split(",") if fwd_only else spec.split(",")
    env_set = dict(e.split("=", 1) for e in envs.split(";") if e)
    saved = {kk: os.environ.get(kk) for kk in env_set}
    os.environ.update(env_set)
    chain = br_amd.Chain(gs, [(m, 5, 7) for m in names], two_side=fwd_only)
    best, prof = None, None
    for rep in range(3):
        _lib.profile_reset()
        t0 = time.perf_counter()
        out_total = chain.correct_batch_device(db.data_ptr(), do.data_ptr(), n_reads, total, d_out.data_ptr(), d_out.numel(),
                                               d_oo.data_ptr(), stream)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        if rep and (best is None or wall < best):   # (the first repetition builds workspaces and the successor table)
            best = wall
            prof = {kk: round(v["total_ms"], 2) for kk, v in _lib.profile_all().items() if v["launches"]}
    st = chain.last_stats()
    passes = (1 if fwd_only else 2) * len(names)
    print(json.dumps({"k": k, "chain": names, "forward_only": fwd_only, "switches": env_set, "reads": n_reads, "bases": int(total), "abundance": a, "coverage": coverage,
                      "solid_kmers": int(gs.popcount()), "index": gs.index_info(), "build_ms_incl_first_touch": round(build_ms, 1),
                      "wall_ms": round(best, 2), "ms_per_pass": round(best / passes, 2),
                      "gbases_per_s_correct_only": round(total / best / 1e6, 3),
                      "roofline_frac_66B": round(66.0 * total * passes / (best * 1e-3) / 8e12, 4),
                      "kernels_ms": prof, "out_bases": int(out_total), **{kk: int(v) for kk, v in st.items()}}), flush=True)
    del chain
    for kk, v in saved.items():
        if v is None:
            os.environ.pop(kk, None)
        else:
            os.environ[kk] = v
