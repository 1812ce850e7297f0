#!/usr/bin/env python3
"""Where a batch handed over in host memory spends its time (BRX_TRACE=2 prints the stages of every
brx_chain_correct_batch): four synchronous 8192-record batches, then twelve through _async / _wait over N chains.
usage: BRX_TRACE=2 python tools/host_async_trace.py [n_chains=3]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import br_amd
from br_amd import _lib, synth

n_ch = int(sys.argv[1]) if len(sys.argv) > 1 else 3
k, a, read_len, per, n_b = 19, 3, 10000, 8192, 4
cfg = synth.config(genome_len=100_000 * read_len // 50, read_len=read_len)
g = synth.genome_host(cfg)
L = _lib.lib()
cnt = br_amd.Counter(k, 0)
batches = []
for b in range(n_b + 1):
    hb, ho = synth.reads_host(cfg, g, b * per, per)
    cnt.add_batch(hb, ho)
    ptr = L.brx_host_alloc(int(ho[-1]))
    C.memmove(ptr, hb.ctypes.data, int(ho[-1]))
    batches.append((ptr, np.ascontiguousarray(ho, dtype=np.uint64), int(ho[-1])))
# (the set of 41 k reads only: smaller than the bench's, same index size class)
gs = cnt.finish(a)
ob, oo = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint64)()
chains = [br_amd.Chain(gs, [("one", 5, 7)], two_side=False) for _ in range(n_ch)]
def sync_call(ch, bf):
    _lib.check(L.brx_chain_correct_batch(ch._h, bf[0], bf[1].ctypes.data, per, C.byref(ob), C.byref(oo)))
    L.brx_buf_free(ob); L.brx_buf_free(oo)
for ch in chains:
    sync_call(ch, batches[n_b])
print("---- synchronous, one chain", file=sys.stderr, flush=True)
t0 = time.perf_counter()
for b in range(n_b):
    sync_call(chains[0], batches[b])
dt_sync = (time.perf_counter() - t0) / n_b * 1e3
print("---- asynchronous, %d chains" % n_ch, file=sys.stderr, flush=True)
seq = [batches[b % n_b] for b in range(12)]
t0 = time.perf_counter()
for j, bf in enumerate(seq):
    ch = chains[j % n_ch]
    if j >= n_ch:
        _lib.check(L.brx_chain_correct_batch_wait(ch._h, C.byref(ob), C.byref(oo)))
        L.brx_buf_free(ob); L.brx_buf_free(oo)
    _lib.check(L.brx_chain_correct_batch_async(ch._h, bf[0], bf[1].ctypes.data, per))
for j in range(len(seq) - n_ch, len(seq)):
    _lib.check(L.brx_chain_correct_batch_wait(chains[j % n_ch]._h, C.byref(ob), C.byref(oo)))
    L.brx_buf_free(ob); L.brx_buf_free(oo)
dt_async = (time.perf_counter() - t0) / len(seq) * 1e3
print(json.dumps({"ms_per_batch_sync": round(dt_sync, 2), "ms_per_batch_async": round(dt_async, 2), "chains": n_ch,
                  "gbases_per_s_sync": round(batches[0][2] / dt_sync / 1e6, 2), "gbases_per_s_async": round(batches[0][2] / dt_async / 1e6, 2)}))
