"""One rank of a world-N run of the SHIPPED multi-GPU exchange (br_amd/csrc/brx_exchange.hip, behind the C ABI) with
all ranks on ONE card: the library's librccl calls are served by tests/libfake_rccl.so (selected through BRX_RCCL_PATH
by the test that starts this process), everything above those ten entry points is the product's code, and every
kernel is the product's.  The communicator id travels through a file: no torch.distributed anywhere.
usage: python abi_exchange_worker.py RANK WORLD K ABUNDANCE N_READS OUT_PREFIX part|dense"""
import os
import pickle
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, k, a, n_reads = (int(x) for x in sys.argv[1:6])
out_prefix, strategy = sys.argv[6], sys.argv[7]

import numpy as np
import br_amd
from br_amd import _lib, fasta
from br_amd import dist as bd

with open(os.path.join(ROOT, "tests", "golden", "raw.fasta"), "rb") as f:
    reads = [seq for _, _, seq in fasta.read_records(f)][:n_reads]
lo, hi = bd.shard_range(len(reads), world, rank)
mine = reads[lo:hi]                                   # may be empty: that rank still joins every collective

id_path = out_prefix + ".id"
if rank == 0:
    ident = bd.AbiExchange.unique_id()
    with open(id_path + ".tmp", "wb") as f:
        f.write(ident)
    os.rename(id_path + ".tmp", id_path)
else:
    t0 = time.time()
    while not os.path.exists(id_path):
        if time.time() - t0 > 120:
            sys.exit("rank %d: no communicator id after 120 s" % rank)
        time.sleep(0.05)
    with open(id_path, "rb") as f:
        ident = f.read()
ex = bd.AbiExchange(world, rank, 0, ident=ident)
result = {"n_mine": len(mine)}
if strategy == "part":
    counter = br_amd.Counter(k, 0, _lib.COUNT_SORTED)
    solid = br_amd.Pcon.new(k)
    for rep in range(2):                              # twice: the bench resets and re-uses the same objects every step
        if mine:
            counter.add_reads(mine)
        ex.build_partitioned(counter, solid, a, None)
        result["stats_%d" % rep] = ex.last_stats()
        chain = br_amd.Chain(solid, [("one", 5, 7), ("graph", 5, 7)], two_side=False)
        result["corrected_%d" % rep] = chain.correct_reads(mine) if mine else []
        del chain
    result["index"] = solid.index_info()
else:
    counter = br_amd.Counter(k, 0, _lib.COUNT_DENSE)
    if mine:
        counter.add_reads(mine)
    ex.reduce_counts(counter, a, None)
    solid = counter.finish(a)
result["popcount"] = solid.popcount()
if k <= 15:
    result["solid_bytes"] = solid.to_solid_bytes()
else:  # 2^(2k-4) bytes per rank is too much to ship: membership of every third k-mer of two reads and of their neighbours
    sample = []
    for r in (reads[0], reads[-1]):
        for j in range(0, len(r) - k + 1, 3):
            km = br_amd.seq2bit(r[j:j + k])
            sample += [km, km ^ 1, km ^ (3 << 10)]
    result["sample"] = sample
    result["members"] = [bool(x) for x in solid.get_many(sample)]
ex.close()
with open("%s.rank%d.pkl" % (out_prefix, rank), "wb") as f:
    pickle.dump(result, f)
