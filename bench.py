#!/usr/bin/env python3
"""bench.py -- corrected Gbases/s at k=19 on synthetic 10 kb ONT-error reads (BASELINE.json).

One STEP = one whole job of the hot path on data already resident in HBM:
    zero the counter -> count canonical k-mers of every read -> (N>1: exchange) -> threshold into
    the solid bitset -> correct every read with correct::One, forward + reverse pass.
value = bases corrected by all ranks / wall time (max over ranks), i.e. set build INCLUDED.
The correction-only and build-only rates are reported next to it in "phases".

Workload at N=1: BASELINE.json configs[1] -- synthetic 1 Gbp (1e5 reads x 10 kb, 50x coverage of a
uniform random genome, 2 % sub / 1.5 % ins / 1.5 % del), k=19, `fasta -a 3` build + `-c one`.
N>1: weak scaling, every rank owns 1e5 reads of one N x 20 Mbp genome; the set is built from ALL
ranks' reads (one exchange step), then replicated; correction needs no communication.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000, help="reads per GPU (10 kb each)")
    ap.add_argument("--read-len", type=int, default=10_000)
    ap.add_argument("--k", type=int, default=19)
    ap.add_argument("--abundance", type=int, default=3)
    ap.add_argument("--confirm", type=int, default=5)
    ap.add_argument("--coverage", type=int, default=50)
    ap.add_argument("--method", default="one", choices=["one", "two", "graph", "greedy", "gap_size"],
                    help="corrector of the step (default: the metric's correct::one; configs[2] is `--reads 1000000 --method greedy`)")
    ap.add_argument("--strategy", default="auto", choices=["auto", "dense", "sorted"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--progress", action="store_true", help="timestamps of the stages on stderr (big configurations)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all on cuda:0 with the gloo backend: checks the multi-rank code path where only one "
                         "card is available (RCCL cannot put two ranks on one GPU); the rate it prints means nothing")
    ap.add_argument("--force-exchange", action="store_true",
                    help="debug: run the multi-GPU exchange step even with one rank (RCCL, world_size 1)")
    ap.add_argument("--cpu-reads", type=int, default=0, help="reads in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="optional PMC-derived HBM bytes per launch, produced by profiles/collect_pmc.sh")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist
    import numpy as np
    import br_amd
    from br_amd import _lib, synth

    t_prog = time.perf_counter()
    dev = 0 if args.rehearse_on_one_gpu else local_rank
    torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream().cuda_stream

    k, a = args.k, args.abundance
    n_reads, read_len = args.reads, args.read_len
    genome_len = max(world * n_reads * read_len // args.coverage, read_len)
    cfg = synth.config(genome_len=genome_len, read_len=read_len)

    # ---- synthetic input, resident in HBM before the timed region --------------------------------
    d_genome = torch.empty(genome_len, dtype=torch.uint8, device="cuda")
    synth.genome_device(cfg, dev, d_genome.data_ptr(), stream)
    cap = int(n_reads * read_len * 1.03) + (1 << 20)
    d_bases = torch.empty(cap, dtype=torch.uint8, device="cuda")
    d_off = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")
    total = synth.reads_device(cfg, dev, d_genome.data_ptr(), rank * n_reads, n_reads, d_bases.data_ptr(), cap,
                               d_off.data_ptr(), stream)
    d_out = torch.empty(int(total * 1.05) + (1 << 20), dtype=torch.uint8, device="cuda")
    d_out_off = torch.empty(n_reads + 1, dtype=torch.int64, device="cuda")

    strategy = {"auto": _lib.COUNT_AUTO, "dense": _lib.COUNT_DENSE, "sorted": _lib.COUNT_SORTED}[args.strategy]
    counter = br_amd.Counter(k, dev, strategy)
    gs = br_amd.Pcon.new(k, dev)
    chain = br_amd.Chain(gs, [(args.method, args.confirm, 7)], two_side=False)

    # The process group is created AFTER the big HBM allocations above (measured: buffers allocated
    # after RCCL's communicator exists stream at a fraction of the bandwidth on this stack).
    if world > 1 or args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from br_amd import dist as brx_dist
    multi = world > 1 or args.force_exchange
    exchanger = brx_dist.SetExchange(world, rank) if multi else None
    partitioned = multi and (args.strategy == "sorted" or (args.strategy == "auto" and k >= 15))

    phase_ms = {"build": 0.0, "correct": 0.0}
    if args.progress and rank == 0:
        print("[bench %8.2f s] input resident: %d reads, %d bases" % (time.perf_counter() - t_prog, n_reads, total), file=sys.stderr, flush=True)

    def note(msg):
        if args.progress and rank == 0:
            print("[bench %8.2f s] %s" % (time.perf_counter() - t_prog, msg), file=sys.stderr, flush=True)

    def step(timed: bool):
        t0 = time.perf_counter()
        counter.reset(stream)
        counter.add_batch_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, total, stream)
        if exchanger is not None and partitioned:
            exchanger.build_partitioned(counter, gs, a, stream)   # keys to their owner, solid set back
        else:
            if exchanger is not None:
                exchanger.reduce_counts(counter, a, stream)              # dense: u8 all-reduce
            counter.finish_into(a, gs, stream)
        if timed:
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        note("set built (%.1f ms)" % ((t1 - t0) * 1e3))
        out_total = chain.correct_batch_device(d_bases.data_ptr(), d_off.data_ptr(), n_reads, total,
                                               d_out.data_ptr(), d_out.numel(), d_out_off.data_ptr(), stream)
        if timed:
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            phase_ms["build"] += (t1 - t0) * 1e3
            phase_ms["correct"] += (t2 - t1) * 1e3
            note("corrected (%.1f ms)" % ((t2 - t1) * 1e3))
        return out_total

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    _lib.profile_enable(True)
    _lib.profile_reset()
    barrier()
    t0 = time.perf_counter()
    out_total = 0
    for _ in range(args.steps):
        out_total = step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tb = torch.tensor([total], dtype=torch.int64, device="cuda")
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        total_all = int(tb.item())
    else:
        total_all = total

    prof = _lib.profile_all()
    stats = chain.last_stats()
    solid_bits = gs.popcount()
    # cheap self-checks of the job (a wrong set shows up here long before anyone diffs FASTA files):
    # every rank must hold the same set, and roughly one solid k-mer per genome position and strand-pair
    checks = {"solid_per_genome_base": round(solid_bits / genome_len, 3)}
    if world > 1:
        lo = torch.tensor([solid_bits], dtype=torch.int64, device="cuda")
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        checks["set_popcount_identical_across_ranks"] = bool(lo.item() == hi.item())
    min_fixes = 0.01 * total if args.method == "one" else 0  # One repairs most isolated errors; Greedy few
    checks["plausible"] = bool(0.9 < solid_bits / genome_len < 1.3 and stats["fixes"] > min_fixes)

    # ---- roofline of the dominant kernel ---------------------------------------------------------
    n_table = 1 << (2 * k - 1)
    alg_bytes = {  # ALGORITHMIC bytes per launch (SURVEY 8(d) per-unit figure x units per launch)
        "correct_pass": 66.0 * total,                 # 1 B in + 1 B out + one 64 B probe per base, per pass
        "correct_pass_two": 66.0 * total, "correct_pass_graph": 66.0 * total, "correct_pass_greedy": 66.0 * total,
        "correct_pass_gap_size": 66.0 * total,        # same figure for every method
        "count_dense": 129.0 * total,                 # 1 B in + 64 B counter line read + 64 B write-back
        "count_keys": 9.0 * total,                    # 1 B in + 8 B key out
        "threshold": float(n_table + n_table // 8),   # stream the u8 table, write the bitset
        "count_zero": float(n_table),
        "compact": 2.0 * total,
    }
    dominant, best = None, -1.0
    for name, v in prof.items():
        if name in alg_bytes and v["launches"] > 0 and v["total_ms"] > best:
            dominant, best = name, v["total_ms"]
    roofline = None
    if dominant:
        avg_ms = prof[dominant]["total_ms"] / prof[dominant]["launches"]
        achieved = alg_bytes[dominant] / (avg_ms * 1e-3) / 1e9
        traffic = None
        try:
            with open(args.traffic_json) as f:
                tj = json.load(f)
            if tj.get("kernel") == dominant and tj.get("bases_per_launch") == total and args.method == "one":
                traffic = tj.get("hbm_bytes_per_launch")
        except Exception:
            pass
        roofline = {"kernel": dominant, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": alg_bytes[dominant],
                    "launches": prof[dominant]["launches"]}

    # ---- CPU baseline: the oracle on a bounded sample, rank 0 at N=1 only ------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, cfg, gs, k, n_reads)

    if rank == 0:
        value = total_all * args.steps / elapsed / 1e9
        line = {
            "metric": "corrected Gbases/sec at k=%d, 10 kb ONT-error reads" % k,
            "value": round(value, 4), "unit": "Gbases/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64 k-mers / 1-bit set",
            "data": "synthetic",
            "config": {"workload": "synthetic %.2f Gbp/GPU ONT-error %d bp reads, k=%d, set build (-a %d) + "
                                   "correct::%s (-C %d) fwd+rev" % (total / 1e9, read_len, k, a, args.method, args.confirm),
                       "reads_per_gpu": n_reads, "bases_per_gpu": total, "genome_len": genome_len,
                       "strategy": args.strategy, "parallelism": "reads sharded x%d" % world,
                       **({"rehearsal": "all ranks on cuda:0 over gloo: code-path check, not a measurement"}
                          if args.rehearse_on_one_gpu else {})},
            "phases": {"build_ms_per_step": round(phase_ms["build"] / args.steps, 3),
                       "correct_ms_per_step": round(phase_ms["correct"] / args.steps, 3),
                       "correct_only_gbases_per_s": round(total * args.steps / (phase_ms["correct"] * 1e-3) / 1e9, 3)
                       if phase_ms["correct"] > 0 else None,
                       "build_only_gbases_per_s": round(total * args.steps / (phase_ms["build"] * 1e-3) / 1e9, 3)
                       if phase_ms["build"] > 0 else None},
            "kernels": {n: {"avg_ms": round(v["total_ms"] / max(v["launches"], 1), 4), "launches": v["launches"]}
                        for n, v in prof.items() if v["launches"]},
            "correct_stats": {**{k_: int(v) for k_, v in stats.items()}, "out_bases": int(out_total),
                              "solid_kmers": int(solid_bits)},
            "probe_index": gs.index_info(),
            "checks": checks,
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if world > 1 or args.force_exchange:
        dist.destroy_process_group()


def cpu_baseline(args, cfg, gs, k, n_reads):
    """Times the CPU oracle (restatement of the reference's scalar path; the Rust reference itself
    cannot be built offline) on a bounded sample: correct::One fwd+rev of the first S reads against
    the same k-mer set (exported from HBM).  Set build is NOT part of this sample (a 2^(2k-1)-byte
    host table at k=19 is 128 GiB); the like-for-like GPU figure is phases.correct_only."""
    import concurrent.futures as cf
    import numpy as np
    from br_amd import synth
    from oracle import oracle as O

    cores = min(os.cpu_count() or 1, 16)
    S = args.cpu_reads or min(n_reads, 4096 * cores)   # ~10-20 s of CPU work on 16 threads
    bits = gs.export_bits()                       # 2^(2k-4) bytes, D2H once
    g = synth.genome_host(cfg)
    bases, offs = synth.reads_host(cfg, g, 0, S)
    solid = O.Solid.wrap(k, bits)
    bounds = np.linspace(0, S, cores + 1).astype(int)

    def work(i):
        lo, hi = int(bounds[i]), int(bounds[i + 1])
        if hi <= lo:
            return 0
        ms = O.build_methods(solid, ["one"], args.confirm, 7)
        sub_off = offs[lo:hi + 1] - offs[lo]
        sub = bases[int(offs[lo]):int(offs[hi])]
        out, oo = O.correct_batch(ms, sub, sub_off, False)
        return int(oo[-1])

    t0 = time.perf_counter()
    with cf.ThreadPoolExecutor(cores) as ex:
        list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    nb = int(offs[S])
    return {"value": round(nb / dt / 1e9, 5), "unit": "Gbases/s", "cores": cores, "kind": "port",
            "sample": "correct::one fwd+rev of the first %d reads (%.1f Mbp) against the GPU-built k=%d set; "
                      "correction phase only, %d threads, %.1f s" % (S, nb / 1e6, k, cores, dt)}


if __name__ == "__main__":
    main()
