#!/usr/bin/env python3
"""Random FASTA-like byte streams through the native host pipeline (brx_run_correction_fd) and through the Python
statement of the same reader / writer rules (br_amd/fasta.py + Chain): the outputs must be identical.
usage: python tools/fuzz_fasta.py [seconds=60] [seed=1]"""
import io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import br_amd
from br_amd.driver import run_correction

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
seqs = [bytes(rng.choice(list(b"ACGT"), size=3000).astype(np.uint8)) for _ in range(3)]
gs = br_amd.Pcon.from_count([s * 3 for s in seqs], 11, 1)
methods = br_amd.build_methods(["one"], gs, 3, 7)
PIECES = [b">", b">r", b" ", b"\t", b"\n", b"\r\n", b"\r", b"desc x", b"ACGT", b"acgtn", b"N", b"", b"\n\n", b">a b\tc \n"]
t_end, case = time.time() + budget, 0
while time.time() < t_end:
    case += 1
    parts = []
    for _ in range(int(rng.integers(0, 40))):
        x = rng.random()
        if x < 0.35:
            s = seqs[int(rng.integers(0, 3))]
            a = int(rng.integers(0, 2900)); b = a + int(rng.integers(0, 400))
            chunk = bytearray(s[a:b])
            for _ in range(int(rng.integers(0, 4))):
                if chunk:
                    chunk[int(rng.integers(0, len(chunk)))] = int(rng.choice(list(b"ACGT")))
            w = int(rng.choice([0, 0, 60, 7]))
            if w:
                chunk = b"\n".join(bytes(chunk[i:i + w]) for i in range(0, len(chunk), w))
            parts.append(bytes(chunk) + bytes(rng.choice([b"\n", b"\r\n", b""])))
        elif x < 0.6:
            parts.append(b">rec%d%s\n" % (case, bytes(rng.choice([b"", b" d", b"  two  words \t", b"\t"]))))
        else:
            parts.append(bytes(PIECES[int(rng.integers(0, len(PIECES)))]))
    text = b"".join(parts)
    ts = bool(rng.random() < 0.5)
    o_native, o_py = io.BytesIO(), io.BytesIO()
    run_correction([io.BytesIO(text)], [o_native], methods, ts, native=True, batch_records=int(rng.choice([0, 1, 3])))
    run_correction([io.BytesIO(text)], [o_py], methods, ts, native=False)
    if o_native.getvalue() != o_py.getvalue():
        print("MISMATCH on", repr(text[:300])); sys.exit(1)
print(f"{case} streams, native == python")
