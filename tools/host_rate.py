#!/usr/bin/env python3
"""PCIe-inclusive rate of the HOST-buffer entry points (what a Rust/C caller of the ABI sees when it hands
over pageable host memory, reference batch size 8192 records): brx_set_count_add_batch + finish, then
brx_chain_correct_batch (H2D, kernels, D2H into malloc'd buffers).  Never the bench's `value`.
usage: python tools/host_rate.py [n_batches=4]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import br_amd
from br_amd import _lib, synth

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 4
k, a, read_len, per = 19, 3, 10000, 8192
n_reads = n_batches * per
cfg = synth.config(genome_len=n_reads * read_len // 50, read_len=read_len)
g = synth.genome_host(cfg)
batches = [synth.reads_host(cfg, g, b * per, per) for b in range(n_batches)]
total = sum(int(o[-1]) for _, o in batches)
cnt = br_amd.Counter(k, 0)
t0 = time.perf_counter()
for hb, ho in batches:
    cnt.add_batch(hb, ho)
gs = cnt.finish(a)
t_build = time.perf_counter() - t0
chain = br_amd.Chain(gs, [("one", 5, 7)], two_side=False)
chain.correct_batch(*batches[0])  # warm-up: workspace + probe index
t0 = time.perf_counter()
out_bases = 0
for hb, ho in batches:
    ob, oo = chain.correct_batch(hb, ho)
    out_bases += int(oo[-1])
t_corr = time.perf_counter() - t0
print(json.dumps({"reads": n_reads, "bases": total, "batch_reads": per,
                  "host_build_gbases_per_s": round(total / t_build / 1e9, 3),
                  "host_correct_gbases_per_s": round(total / t_corr / 1e9, 3),
                  "ms_per_batch_correct": round(t_corr / n_batches * 1e3, 2), "out_bases": out_bases}))
