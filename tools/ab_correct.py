#!/usr/bin/env python3
"""A/B harness (one process, interleaved rounds, cdna_hip_programming.md rule 24) for the correction
kernel's tuning switches: BRX_TUNE bit flags and BRX_GROUP.  Prints median/min ms of the forward and
reverse correct_pass launches and the probes issued, per variant.
usage: python tools/ab_correct.py [reads=100000] [rounds=5] variant ...   variant = name:GROUP:TUNE"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import br_amd
from br_amd import _lib, synth

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
variants = [(v.split(":") + ["", ""])[:4] for v in (sys.argv[3:] or ["base:16:0", "legacy:16:7"])]  # name:G:TUNE[:G_REV]
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
chain = br_amd.Chain(gs, [("one", 5, 7)], two_side=False)
d_out = torch.empty(int(total * 1.05) + (1 << 20), dtype=torch.uint8, device="cuda")
d_oo = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")
res = {v[0]: [] for v in variants}
probes = {}
_lib.profile_enable(True)
for r in range(rounds + 1):
    for name, grp, tune, grev in variants:
        os.environ["BRX_GROUP"], os.environ["BRX_TUNE"] = grp, tune
        os.environ["BRX_GROUP_REV"] = grev or grp
        _lib.profile_reset()
        chain.correct_batch_device(db.data_ptr(), do.data_ptr(), n_reads, total, d_out.data_ptr(), d_out.numel(), d_oo.data_ptr(), stream)
        ms, n = _lib.profile_get("correct_pass")
        if r:
            res[name].append(ms)
        probes[name] = chain.last_stats()["probes"]
for name, v in res.items():
    print(f"{name:12s} correct fwd+rev: median {statistics.median(v):7.2f} ms  min {min(v):7.2f} ms  probes {probes[name]/1e9:.3f} G"
          f"  -> {2 * total / statistics.median(v) / 1e6:.2f} Gbase-pass/s")
