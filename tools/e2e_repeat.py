#!/usr/bin/env python3
"""The FASTA file -> FASTA file leg of bench.py, several times in one process, every stage timed on its own: where a
slow repetition loses its time (bench.py's `e2e` block lists two runs; one of them is sometimes several times slower in
its build leg).  usage: python tools/e2e_repeat.py [reads=100000] [repetitions=6]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import br_amd
from br_amd import synth
from br_amd.driver import run_correction

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
k, a = 19, 3
cfg = synth.config(genome_len=n_reads * 10000 // 50, read_len=10000)
g = synth.genome_host(cfg)
tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
src, dst = os.path.join(tmp, "brx_e2e_rep_in.fasta"), os.path.join(tmp, "brx_e2e_rep_out.fasta")
total = 0
with open(src, "wb") as f:
    for lo in range(0, n_reads, 5000):
        bases, offs = synth.reads_host(cfg, g, lo, min(5000, n_reads - lo))
        for r in range(len(offs) - 1):
            f.write(b">r%d\n" % (lo + r)); f.write(bases[int(offs[r]):int(offs[r + 1])].tobytes()); f.write(b"\n")
        total += int(offs[-1])
print("input", total, "bases", flush=True)
try:
    for rep in range(reps):
        t = [time.perf_counter()]
        cnt = br_amd.Counter(k, 0); t.append(time.perf_counter())
        with open(src, "rb") as f:
            cnt.count_fasta(f)
        t.append(time.perf_counter())
        gs = cnt.finish(a); t.append(time.perf_counter())
        del cnt; t.append(time.perf_counter())
        if os.path.exists(dst):
            os.remove(dst)  # (outside the timed stages: truncating 1 GB of tmpfs costs 50-100 ms)
        methods = br_amd.build_methods(["one"], gs, 5, 7); t.append(time.perf_counter())
        with open(src, "rb") as fi, open(dst, "wb") as fo:
            st = run_correction([fi], [fo], methods, False, native=True)
        t.append(time.perf_counter())
        del methods, gs; t.append(time.perf_counter())
        names = ["Counter()", "count_fasta", "finish", "del counter (+ rm out)", "build_methods", "run_correction", "del chain+set"]
        print("rep %d: %s | end to end %.2f Gbases/s" % (rep, "  ".join("%s %.3f" % (n_, t[i + 1] - t[i]) for i, n_ in enumerate(names)),
                                                        total / (t[6] - t[0]) / 1e9), flush=True)
finally:
    for p in (src, dst):
        if os.path.exists(p):
            os.remove(p)
