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
variants = [(v.split(":") + ["", "", ""])[:5] for v in (sys.argv[3:] or ["base:16:0", "legacy:16:7"])]  # name:G:TUNE[:G_REV[:off|M,LOG2LINES]]
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
rounds_n = {}
info = {}
idx_now = {}
_lib.profile_enable(True)
for r in range(rounds + 1):
    for name, grp, tune, grev, idx in variants:
        os.environ["BRX_GROUP"], os.environ["BRX_TUNE"] = grp, tune
        os.environ["BRX_GROUP_REV"] = grev or grp
        os.environ["BRX_INDEX"] = "0" if idx == "off" else "1"
        if idx != "off" and idx_now.get("v") != idx:
            m_, ll_ = (idx.split(",") + ["0"])[:2] if idx else ("0", "0")
            info[name] = gs.index_build(int(m_), int(ll_), stream)
            idx_now["v"] = idx
        _lib.profile_reset()
        chain.correct_batch_device(db.data_ptr(), do.data_ptr(), n_reads, total, d_out.data_ptr(), d_out.numel(), d_oo.data_ptr(), stream)
        ms, n = _lib.profile_get("correct_pass")
        if r:
            res[name].append(ms)
        print(f"# round {r} {name}: {ms:.2f} ms", flush=True)
        probes[name] = chain.last_stats()["probes"]
        rounds_n[name] = chain.last_stats()["rounds"]
for name, v in res.items():
    print(f"{name:12s} correct fwd+rev: median {statistics.median(v):7.2f} ms  min {min(v):7.2f} ms  probes {probes[name]/1e9:.3f} G"
          f" rounds {rounds_n[name]/1e6:.1f} M  -> {2 * total / statistics.median(v) / 1e6:.2f} Gbase-pass/s  {info.get(name, '')}")
