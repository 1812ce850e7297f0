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

import json, collections
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only_case = int(sys.argv[3]) if len(sys.argv) > 3 else 0
repeat = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # (with only_case: run the case this many times -- rare races)
# FUZZ_FOCUS=walklane: every job ends in graph or gap_size with the lane forms on and short chunks (their own random
# stream, drawn after the case's: the cases of a seed stay what they are without it)
focus = os.environ.get("FUZZ_FOCUS", "")
frng = np.random.default_rng(seed + 7919)
# switches added after round 3 draw from a stream of their own, every case, so that the cases of a seed stay what they
# were (seed 1, case 3213 is on record: profiles/r4_case3213_audit.md)
xrng = np.random.default_rng(seed + 104729)
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


history = collections.deque(maxlen=9)  # this case's settings and the eight before it (process state carried over?)
REPORT_DIR = os.environ.get("FUZZ_REPORT_DIR", os.path.join(ROOT, "gpurun_out"))


def first_diff(a: bytes, b: bytes) -> int:
    m = min(len(a), len(b))
    for x in range(m):
        if a[x] != b[x]:
            return x
    return m if len(a) != len(b) else -1


def run_chain(gs, names, c, ms, two_side, reads, env_over=None):
    """the same job once more in this process, optionally with some switches changed"""
    saved = {}
    for key, v in (env_over or {}).items():
        saved[key] = os.environ.get(key)
        if v == "":
            os.environ.pop(key, None)
        else:
            os.environ[key] = v
    try:
        return br_amd.Chain(gs, [(m, c, ms) for m in names], two_side=two_side).correct_reads(reads)
    finally:
        for key, v in saved.items():
            if v is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = v


def report_mismatch(desc, case, seed, gs, ref, names, c, ms, two_side, reads, got, bad, rebuild):
    """Everything needed to explain a mismatch without seeing it again, written to a file that is kept: the reads that
    differ (input, want, got, first differing offset), the settings of this case and the eight before it, and what the
    same process answers when the job is asked again: unchanged, with the lane forms off, with a freshly built set, and
    pass by pass (each pass of the chain alone, fed the oracle's intermediate: which pass diverges first)."""
    om = O.build_methods(ref, names, c, ms)
    rep = {"seed": seed, "case": case, "desc": desc, "previous_cases": list(history), "reads": [], "reruns": {}, "passes": []}
    for ri in bad[:8]:
        want = O.correct_record(om, reads[ri], two_side)
        rep["reads"].append({"read": ri, "len": len(reads[ri]), "first_diff": first_diff(want, got[ri]), "input": reads[ri].decode("latin1"),
                             "want": want.decode("latin1"), "got": got[ri].decode("latin1")})
    want_all = [O.correct_record(om, r, two_side) for r in reads]

    def differing(out):
        return [ri for ri in range(len(reads)) if out[ri] != want_all[ri]]
    try:
        rep["reruns"]["same_settings_same_set"] = differing(run_chain(gs, names, c, ms, two_side, reads))
        rep["reruns"]["lane_off"] = differing(run_chain(gs, names, c, ms, two_side, reads, {"BRX_LANE": "0"}))
        rep["reruns"]["lane_mask_off"] = differing(run_chain(gs, names, c, ms, two_side, reads, {"BRX_LANE_MASK": "0"}))
        rep["reruns"]["lane_walk_off"] = differing(run_chain(gs, names, c, ms, two_side, reads, {"BRX_LANE_WALK": "0"}))
        rep["reruns"]["rev_lean_off"] = differing(run_chain(gs, names, c, ms, two_side, reads, {"BRX_REV_LEAN": "0"}))
        gs2 = rebuild()
        rep["reruns"]["fresh_set_popcount"] = [int(gs2.popcount()), int(ref.popcount())]
        rep["reruns"]["same_settings_fresh_set"] = differing(run_chain(gs2, names, c, ms, two_side, reads))
        # pass by pass: forward passes of the chain in order, then (unless two_side) the same over the reversed reads --
        # every pass is the GPU chain [method] alone, forward only, on the ORACLE's intermediate of the reads that differed
        cur = [reads[ri] for ri in bad[:8]]
        for direction in range(1 if two_side else 2):
            if direction == 1:
                cur = [x[::-1] for x in cur]
            for m in names:
                o1 = O.build_methods(ref, [m], c, ms)
                want1 = [O.correct_record(o1, x, True) for x in cur]
                got1 = run_chain(gs, [m], c, ms, True, cur)
                got1_off = run_chain(gs, [m], c, ms, True, cur, {"BRX_LANE": "0"})
                rep["passes"].append({"dir": direction, "method": m,
                                      "differs": [j for j in range(len(cur)) if got1[j] != want1[j]],
                                      "differs_lane_off": [j for j in range(len(cur)) if got1_off[j] != want1[j]]})
                cur = want1
    except Exception as e:  # the report must come out whatever the re-runs do
        rep["rerun_error"] = repr(e)
    os.makedirs(REPORT_DIR, exist_ok=True)
    path = os.path.join(REPORT_DIR, f"fuzz_mismatch_seed{seed}_case{case}.json")
    with open(path, "w") as f:
        json.dump(rep, f, indent=1)
    print("READ MISMATCH", desc, "\nreport:", path, "\nreads:", bad[:8], "first_diff:", [x["first_diff"] for x in rep["reads"]],
          "\nreruns:", rep["reruns"], "\npasses:", rep["passes"], flush=True)


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
    # reverse passes in lane form (round 4): by index size (default: off at these sizes) / Graph / GapSize / both / off
    env["BRX_LANE_REV"] = str(xrng.choice(["", "1", "2", "3", "3", "0"]))
    # graded chunk lengths at the end of a batch (One's lane form): on (default) / off
    env["BRX_LANE_TAIL"] = str(xrng.choice(["", "1", "1"]))
    # reverse passes of Two / Graph / Greedy / GapSize in lean form (rev_scan_kernel): on (default) / off; the form exists
    # for 64-lane reverse groups, which half of the cases now get whatever BRX_GROUP says
    env["BRX_REV_LEAN"] = str(xrng.choice(["", "", "", "0"]))
    rev64 = bool(xrng.random() < 0.5)
    env["BRX_REV_VERIFY_G"] = str(xrng.choice(["", "4"]))  # lanes per open trigger of the verify pass
    env["BRX_REV_LEAN_ONE"] = str(xrng.choice(["", "1"]))  # One's reverse pass in the lean form, too
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
    os.environ["BRX_GROUP_REV"] = "64" if rev64 else (os.environ.get("BRX_GROUP", "") or "64")
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
    history.append(desc)

    def rebuild():
        if continue_presence:
            return br_amd.Pcon.from_fasta(counted, k, batch=batch)
        cnt2 = br_amd.Counter(k, 0, strategy)
        if counted:
            cnt2.add_reads(counted)
        return cnt2.finish(a)
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
        the_chain = br_amd.Chain(gs, [(m, c, ms) for m in names], two_side=two_side)
        got = the_chain.correct_reads(reads)
        if the_chain.last_stats()["lane_unwritten_units"]:
            print("INVARIANT BROKEN: unit records never written by the lane pass", the_chain.last_stats(), desc); sys.exit(1)
        del the_chain
    except _lib.BrxError as e:
        if "does not terminate" in str(e):   # greedy can spin in the reference too: not comparable
            print("skip (non-terminating greedy)", desc); continue
        print("ERROR", e, desc); sys.exit(1)
    bad = [ri for ri, (r, g_) in enumerate(zip(reads, got)) if g_ != O.correct_record(om, r, two_side)]
    if bad:
        report_mismatch(desc, case, seed, gs, ref, names, c, ms, two_side, reads, got, bad, rebuild)
        if not only_case:
            sys.exit(1)
        continue
    print("ok", desc, flush=True)
print(f"{case} cases, no mismatch")
