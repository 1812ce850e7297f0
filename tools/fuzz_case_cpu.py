#!/usr/bin/env python3
"""Rebuild one job of tools/fuzz_parity.py WITHOUT a GPU and run the oracle over it pass by pass.
usage: python tools/fuzz_case_cpu.py SEED CASE [read_prefix]
The fuzzer's generator is deterministic: its own loop is executed up to the first GPU call of the wanted case, then
every read (or the reads starting with `read_prefix`) goes through the chain one pass at a time with the oracle:
length in / out, triggers, fixes and the solidity of the pass's input.  This is how profiles/r4_case3213_audit.md
found that the one read round 3's unexplained mismatch named was the batch's first and had no trigger in any pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
seed, case = sys.argv[1], sys.argv[2]
prefix = sys.argv[3].encode() if len(sys.argv) > 3 else None
src = open(os.path.join(ROOT, "tools", "fuzz_parity.py")).read()
head, rest = src.split("t_end = time.time() + budget\ncase = 0\n", 1)
body = rest.split("    if only_case and case != only_case:\n", 1)[0].replace("while time.time() < t_end:", "while True:")
body += "    if case == only_case:\n        break\n"
sys.argv = ["fuzz_parity.py", "1e9", seed, case]
g = {"__file__": os.path.join(ROOT, "tools", "fuzz_parity.py")}
exec(compile(head, "fuzz_parity.py(head)", "exec"), g)
exec(compile("case = 0\n" + body, "fuzz_parity.py(loop)", "exec"), g)
O = g["O"]
k, a, c, ms, reads, names, two_side = g["k"], g["a"], g["c"], g["ms"], g["reads"], g["names"], g["two_side"]
print(f"case {g['case']}: k={k} a={a} c={c} ms={ms} reads={len(reads)} chain={names} two_side={two_side} env={g['env']}")
ref = O.Solid.sparse_from_count(k, reads, a) if k >= 17 else O.Solid.from_count(k, O.count_reads(k, reads), a)
print("solid k-mers:", ref.popcount())
for ri, r in enumerate(reads):
    if prefix is not None and not r.startswith(prefix):
        continue
    cur, tot_trig, lines = r, 0, []
    for direction in range(1 if two_side else 2):
        if direction == 1:
            cur = cur[::-1]
        for m in names:
            om = O.build_methods(ref, [m], c, ms)
            nxt = O.correct_record(om, cur, True)
            st = om[0].stats()
            msk = ref.mask(cur)
            tot_trig += st["triggers"]
            lines.append(f"   dir {direction} {m:8s} {len(cur):6d} -> {len(nxt):6d}  triggers {st['triggers']:4d} fixes {st['fixes']:4d}  "
                         f"solid k-mers of the input {int(sum(bin(int(x)).count('1') for x in msk))}")
            cur = nxt
    if prefix is not None or tot_trig == 0:
        print(f"read {ri}: {len(r)} bases, {tot_trig} triggers in all passes")
        if prefix is not None:
            print("\n".join(lines))
