#!/usr/bin/env python3
"""End-to-end rate of the drop-in path on FASTA files (what `python -m br_amd ... fasta -k 19 -a 3` does):
count_fasta (parse + H2D + partition) -> finish -> run_correction (parse -> GPU -> format -> write), files in
/dev/shm so that the disk is not what is measured.  Not the contract bench (bench.py is).
usage: python tools/e2e_cli.py [reads=100000] [python_reads=2000]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import br_amd
from br_amd import synth
from br_amd.driver import run_correction

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n_py = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
k, a, read_len = 19, 3, 10000
tmp = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
src, dst, dst_py = (os.path.join(tmp, f"brx_e2e_{os.getpid()}_{x}.fasta") for x in ("in", "out", "outpy"))
cfg = synth.config(genome_len=n_reads * read_len // 50, read_len=read_len)
stream = torch.cuda.current_stream().cuda_stream
dg = torch.empty(cfg.genome_len, dtype=torch.uint8, device="cuda")
synth.genome_device(cfg, 0, dg.data_ptr(), stream)
cap = int(n_reads * read_len * 1.03) + (1 << 20)
db = torch.empty(cap, dtype=torch.uint8, device="cuda")
do = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")
total = synth.reads_device(cfg, 0, dg.data_ptr(), 0, n_reads, db.data_ptr(), cap, do.data_ptr(), stream)
hb, ho = db[:total].cpu().numpy(), do.cpu().numpy()
del db, dg
try:
    with open(src, "wb") as f:
        for r in range(n_reads):
            f.write(b">r%d\n" % r)
            f.write(hb[int(ho[r]):int(ho[r + 1])].tobytes())
            f.write(b"\n")
    res = {"reads": n_reads, "bases": int(total), "file_bytes": os.path.getsize(src)}

    def throttled():  # CFS bandwidth throttling of this container (cgroup v2), in thread-seconds
        try:
            with open("/sys/fs/cgroup/cpu.stat") as f:
                d = dict(line.split() for line in f)
            return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)) / 1e6
        except OSError:
            return 0, 0.0

    th0 = throttled()
    t0 = time.perf_counter()
    cnt = br_amd.Counter(k, 0)
    with open(src, "rb") as f:
        st_c = cnt.count_fasta(f)
    gs = cnt.finish(a)
    t1 = time.perf_counter()
    del cnt
    methods = br_amd.build_methods(["one"], gs, 5, 7)
    with open(src, "rb") as fi, open(dst, "wb") as fo:
        st = run_correction([fi], [fo], methods, False, native=True)
    t2 = time.perf_counter()
    th1 = throttled()
    res.update({"cpu_throttle_events": th1[0] - th0[0], "cpu_throttled_thread_s": round(th1[1] - th0[1], 3)})
    res.update({"build_s": round(t1 - t0, 3), "build_gbases_per_s": round(total / (t1 - t0) / 1e9, 3),
                "build_parse_s": round(st_c["ns_parse"] / 1e9, 3), "build_gpu_s": round(st_c["ns_gpu"] / 1e9, 3),
                "correct_s": round(t2 - t1, 3), "correct_gbases_per_s": round(total / (t2 - t1) / 1e9, 3),
                "correct_parse_s": round(st["ns_parse"] / 1e9, 3), "correct_gpu_s_2workers": round(st["ns_gpu"] / 1e9, 3),
                "correct_write_s": round(st["ns_write"] / 1e9, 3), "batches": st["batches"],
                "end_to_end_gbases_per_s": round(total / (t2 - t0) / 1e9, 3), "out_bytes": os.path.getsize(dst)})
    # the record-by-record Python driver on the head of the same file, and that both say the same
    head = os.path.join(tmp, f"brx_e2e_{os.getpid()}_head.fasta")
    with open(src, "rb") as fi, open(head, "wb") as fo:
        for _ in range(2 * n_py):
            fo.write(fi.readline())
    t3 = time.perf_counter()
    with open(head, "rb") as fi, open(dst_py, "wb") as fo:
        run_correction([fi], [fo], methods, False, native=False)
    t4 = time.perf_counter()
    nbytes = os.path.getsize(dst_py)
    with open(dst, "rb") as f1, open(dst_py, "rb") as f2:
        same = f1.read(nbytes) == f2.read()
    os.remove(head)
    res.update({"python_driver_gbases_per_s": round(int(ho[n_py]) / (t4 - t3) / 1e9, 4), "python_vs_native_identical": same})
    print(json.dumps(res))
finally:
    for p in (src, dst, dst_py):
        if os.path.exists(p):
            os.remove(p)
