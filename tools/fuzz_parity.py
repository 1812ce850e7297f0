#!/usr/bin/env python3
"""Randomised differential test: the HIP correctors against the CPU oracle on small random jobs with odd shapes
(k, confirm, max_search, abundance, read lengths around k, error rates, method chains, group widths, index on/off,
sparse/lazy sets).  Prints one line per case and exits non-zero at the first mismatch.
usage: python tools/fuzz_parity.py [seconds=120] [seed=1] [only_case=0]
(only_case: replay that case of the seed's sequence alone -- the cases before it are generated and skipped)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import br_amd
from br_amd import _lib
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only_case = int(sys.argv[3]) if len(sys.argv) > 3 else 0
repeat = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # (with only_case: run the case this many times -- rare races)
# FUZZ_FOCUS=walklane: every job ends in graph or gap_size with the lane forms on and short chunks (their own random
# stream, drawn after the case's: the cases of a seed stay what they are without it)
focus = os.environ.get("FUZZ_FOCUS", "")
frng = np.random.default_rng(seed + 7919)
rng = np.random.default_rng(seed)
METHODS = ["one", "two", "graph", "greedy", "gap_size"]
ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_reads(glen, n_reads, rl_lo, rl_hi, err):
    g = ALPHA[rng.integers(0, 4, glen)]
    reads = []
    for _ in range(n_reads):
        L = int(rng.integers(rl_lo, rl_hi + 1))
        if L == 0:
            reads.append(b"")
            continue
        s = int(rng.integers(0, max(glen - L, 1)))
        r = g[s:s + L].copy()
        if rng.random() < 0.5:
            r = ALPHA[(np.searchsorted(ALPHA, r) ^ 2)][::-1]  # revcomp in the A0 C1 T2 G3 coding is not xor on ASCII: do it by index
        out = []
        for b in r.tolist():
            x = rng.random()
            if x < err / 3:
                out.append(int(ALPHA[rng.integers(0, 4)]))           # substitution (maybe silent)
            elif x < 2 * err / 3:
                out.append(b); out.append(int(ALPHA[rng.integers(0, 4)]))  # insertion
            elif x < err:
                pass                                                  # deletion
            else:
                out.append(b)
        if rng.random() < 0.05 and out:
            out[int(rng.integers(0, len(out)))] = ord("N")
        reads.append(bytes(out))
    return reads


t_end = time.time() + budget
case = 0
while time.time() < t_end:
    case += 1
    k = int(rng.choice([5, 7, 9, 11, 13, 15, 17, 19, 21, 25]))
    a = int(rng.choice([0, 1, 2, 3]))
    c = int(rng.choice([0, 1, 2, 5, 9]))
    ms = int(rng.choice([1, 3, 7, 12]))
    glen = int(rng.choice([300, 2000, 20000]))
    n_reads = int(rng.integers(1, 120))
    reads = make_reads(glen, n_reads, 0, int(rng.choice([k - 1, 3 * k, 400, 3000])), float(rng.choice([0.0, 0.02, 0.08, 0.2])))
    chain_len = int(rng.choice([1, 1, 2, 3]))
    names = [str(rng.choice(METHODS)) for _ in range(chain_len)]
    if "two" in names and c == 0:
        c = 1  # Two with -C 0 makes the reference panic (two.rs:265: a scenario whose apply() is None scores 0 == c)
    two_side = bool(rng.random() < 0.3)
    env = {"BRX_GROUP": str(rng.choice(["", "4", "8", "16", "32", "64"])), "BRX_INDEX": str(rng.choice(["1", "1", "0"])),
           "BRX_GROUP_WALK": str(rng.choice(["", "4", "8", "16"])), "BRX_LINE_BITS": str(rng.choice(["", "0"])),
           "BRX_INDEX_FWD": str(rng.choice(["", "1", "31"])), "BRX_REDO_MAX": str(rng.choice(["", "", "0"])),
           "BRX_MAXPATH": str(rng.choice(["", "", "3"])),
           "BRX_INDEX_MIN_K": "5", "BRX_FORCE_SPARSE": str(rng.choice(["0", "0", "1"])),
           "BRX_LAZY_BITS": str(rng.choice(["1", "0"])), "BRX_INDEX_LOG_LINES": str(rng.choice(["0", "0", "5"])),
           # the LDS hash count of the partitioned finish: wrong guesses of the share of distinct keys and tiny tables
           "BRX_HF_RATIO": str(rng.choice(["", "", "0.01", "1"])), "BRX_HF_LOG_T": str(rng.choice(["", "", "6", "8"])),
           "BRX_HF_MIN_LT": str(rng.choice(["", "", "4"])), "BRX_WIDE_L2": str(rng.choice(["", "1", "0"])),
           "BRX_HASH_FINAL": str(rng.choice(["", "", "0"])),
           # One's forward pass cut into units (brx_onelane.hip): off, tiny chunks, sync runs from sloppy to strict
           "BRX_LANE": str(rng.choice(["", "", "", "0"])), "BRX_LANE_CHUNK": str(rng.choice(["", "64", "100", "333"])),
           "BRX_LANE_SYNC": str(rng.choice(["", "1", "2", "8"])), "BRX_LANE_WALK": str(rng.choice(["", "", "", "0"])),
           # level 1 of the partitioned build: blocks that take many tiles each even on small inputs
           "BRX_L1_GRID": str(rng.choice(["", "", "1", "5"])),
           # the solidity mask of the original k-mers: off / walking correctors (default) / One as well
           "BRX_LANE_MASK": str(rng.choice(["", "0", "2", "2"]))}
    if focus == "walklane":
        names[-1] = str(frng.choice(["graph", "gap_size"]))
        env.update({"BRX_LANE": "", "BRX_LANE_WALK": "", "BRX_LANE_CHUNK": str(frng.choice(["64", "100"])),
                    "BRX_LANE_SYNC": str(frng.choice(["1", "2", "4"])), "BRX_LANE_MASK": str(frng.choice(["", "0", "2"]))})
        if c > 5 and names[-1] == "gap_size":
            c = 5
    for key, v in env.items():
        if v == "":
            os.environ.pop(key, None)
        else:
            os.environ[key] = v
    os.environ["BRX_GROUP_REV"] = os.environ.get("BRX_GROUP", "") or "64"
    strategy = _lib.COUNT_SORTED if (k >= 17 or (k >= 7 and rng.random() < 0.7)) else _lib.COUNT_DENSE
    if k > 21:
        continue_presence = True  # counting stops at k = 21: larger k are presence-only (large-kmer) sets
    else:
        continue_presence = False
    if strategy == _lib.COUNT_DENSE:
        os.environ["BRX_FORCE_SPARSE"] = "0"
    counted = [r for r in reads]
    batch = int(rng.choice([3, 50, 8192])) if continue_presence else 0
    if only_case and case != only_case:
        if case > only_case:
            break
        continue
    if continue_presence:
        a = 0
        gs = br_amd.Pcon.from_fasta(counted, k, batch=batch)
    else:
        cnt = br_amd.Counter(k, 0, strategy)
        if counted:
            cnt.add_reads(counted)
        gs = cnt.finish(a)
    ref = O.Solid.sparse_from_count(k, counted, a) if k >= 17 else O.Solid.from_count(k, O.count_reads(k, counted), a)
    desc = f"case {case}: k={k} a={a} c={c} ms={ms} reads={n_reads} chain={names} two_side={two_side} strat={strategy} {env}"
    if gs.popcount() != ref.popcount():
        print("SET MISMATCH", desc); sys.exit(1)
    om = O.build_methods(ref, names, c, ms)
    for again in range(repeat - 1 if only_case else 0):
        # the whole job again, set build included (allocations, atomics and timing differ from run to run)
        cnt2 = br_amd.Counter(k, 0, strategy)
        if counted:
            cnt2.add_reads(counted)
        gs2 = cnt2.finish(a) if not continue_presence else br_amd.Pcon.from_fasta(counted, k, batch=batch)
        got2 = br_amd.Chain(gs2, [(m, c, ms) for m in names], two_side=two_side).correct_reads(reads)
        bad = [ri for ri, (r, g_) in enumerate(zip(reads, got2)) if g_ != O.correct_record(om, r, two_side)]
        if gs2.popcount() != ref.popcount() or bad:
            print("REPEAT MISMATCH at repetition", again, "set", gs2.popcount(), ref.popcount(), "reads", bad[:10])
            for ri in bad[:3]:
                print(" read", ri, len(reads[ri]), reads[ri][:600], "\n want", O.correct_record(om, reads[ri], two_side)[:600], "\n got ", got2[ri][:600])
    try:
        got = br_amd.Chain(gs, [(m, c, ms) for m in names], two_side=two_side).correct_reads(reads)
    except _lib.BrxError as e:
        if "does not terminate" in str(e):   # greedy can spin in the reference too: not comparable
            print("skip (non-terminating greedy)", desc); continue
        print("ERROR", e, desc); sys.exit(1)
    for ri, (r, g_) in enumerate(zip(reads, got)):
        want = O.correct_record(om, r, two_side)
        if g_ != want:
            print("READ MISMATCH", desc, "\nread", ri, "of length", len(r), ":", r[:400], "\nwant:", want[:400], "\ngot: ", g_[:400])
            if only_case:
                continue
            sys.exit(1)
    print("ok", desc, flush=True)
print(f"{case} cases, no mismatch")
