"""One rank of the two-process rehearsal of the multi-GPU set exchange on ONE card (tests/test_gpu_parity.py):
`gloo` between the ranks (RCCL cannot put two ranks on one GPU), the real HIP kernels for everything else.
usage: python dist_gpu_worker.py RANK WORLD PORT K ABUNDANCE N_READS OUT_PREFIX"""
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, port, k, a, n_reads = (int(x) for x in sys.argv[1:7])
out_prefix = sys.argv[7]

import numpy as np
import torch
import torch.distributed as dist
import br_amd
from br_amd import _lib, fasta
from br_amd import dist as bd

with open(os.path.join(ROOT, "tests", "golden", "raw.fasta"), "rb") as f:
    reads = [seq for _, _, seq in fasta.read_records(f)][:n_reads]
lo, hi = bd.shard_range(len(reads), world, rank)
mine = reads[lo:hi]
dist.init_process_group("gloo", rank=rank, world_size=world, init_method=f"tcp://127.0.0.1:{port}")
try:
    stream = torch.cuda.current_stream().cuda_stream
    counter = br_amd.Counter(k, 0, _lib.COUNT_SORTED)
    solid = br_amd.Pcon.new(k)
    bases, offs = br_amd.pack_reads(mine)
    db = torch.from_numpy(bases.copy()).cuda()
    do = torch.from_numpy(offs.astype(np.int64)).cuda()
    result = {}
    for rep in range(2):  # twice: the bench resets and re-uses the same objects every step
        counter.reset(stream)
        counter.add_batch_device(db.data_ptr(), do.data_ptr(), len(mine), int(offs[-1]), stream)
        bd.SetExchange(world, rank).build_partitioned(counter, solid, a, stream)
        torch.cuda.synchronize()
        chain = br_amd.Chain(solid, [("one", 5, 7), ("graph", 5, 7)], two_side=False)
        result["corrected_%d" % rep] = chain.correct_reads(mine)
        del chain
    result["bits_state_after_correct"] = solid.bits_state()
    result["index"] = solid.index_info()
    if k <= 15:
        result["solid_bytes"] = solid.to_solid_bytes()
    else:  # 2^(2k-4) bytes per rank is too much to ship: membership of every k-mer of two reads and of their neighbours
        sample = []
        for r in (reads[0], reads[-1]):
            for j in range(0, len(r) - k + 1, 3):
                km = br_amd.seq2bit(r[j:j + k])
                sample += [km, km ^ 1, km ^ (3 << 10)]
        result["sample"] = sample
        result["members"] = [bool(x) for x in solid.get_many(sample)]
    with open("%s.rank%d.pkl" % (out_prefix, rank), "wb") as f:
        pickle.dump(result, f)
    dist.barrier()
finally:
    dist.destroy_process_group()
