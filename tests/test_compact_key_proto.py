"""CPU: the compact index entry proposed for the next round (tools/compact_key_proto.py) identifies
a solid k-mer exactly -- (line, code) is a function of the canonical k-mer alone, the same for both strands, and
decodes back to it -- for the parameter sets the shipped index uses, on random k-mers and on the ones where a minimizer
scheme goes wrong first: homopolymers and short periods (the same m-mer in every window: ties), palindromic m-mers
(m = 16: an m-mer that is its own reverse complement), k-mers that differ only in a flank base."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import compact_key_proto as P  # noqa: E402

CODE = {ord("A"): 0, ord("C"): 1, ord("T"): 2, ord("G"): 3}  # brx_kmer.hpp: (c >> 1) & 3


def pack(s):
    v = 0
    for ch in s.encode():
        v = (v << 2) | CODE[ch]
    return v


@pytest.mark.parametrize("k,m,log_lines,bits", [(19, 15, 25, 19), (21, 16, 29, 17), (21, 15, 28, 20), (15, 13, 20, 19), (31, 15, 25, 45)])
def test_code_identifies_the_canonical_kmer(k, m, log_lines, bits):
    assert P.code_bits(k, m, log_lines) == bits
    rng = np.random.default_rng(k * 100 + m)
    fwd = rng.integers(0, 1 << (2 * k), 60_000, dtype=np.uint64)
    special = []
    for unit in ("A", "C", "G", "T", "AC", "AT", "CG", "ACG", "ACGT", "AACC", "ACCGGT"):
        s = (unit * k)[:k]
        special += [pack(s), pack(s[1:] + "A"), pack("G" + s[:-1])]
    pal16 = "ACGTACGTACGTACGT"  # its own reverse complement
    for pad in ("AAAAA", "CATGC", "TTTTT"):
        s = (pad + pal16 + pad[::-1])[:k] if k >= 21 else (pal16 + pad)[:k]
        special.append(pack(s))
    fwd = np.concatenate([fwd, np.array(special, dtype=np.uint64)])
    # neighbours that differ in one flank base only: same minimizer, same line, different code
    fwd = np.concatenate([fwd, fwd[:2000] ^ np.uint64(1), fwd[:2000] ^ (np.uint64(1) << np.uint64(2 * k - 2))])
    cano = P.canonical(fwd, k)
    line, code = P.encode(fwd, k, m, log_lines)
    assert int(code.max()).bit_length() <= bits and int(line.max()) < (1 << log_lines)
    line_rc, code_rc = P.encode(P.revcomp(fwd, k), k, m, log_lines)
    assert np.array_equal(line, line_rc) and np.array_equal(code, code_rc)  # strand-symmetric
    back, ok = P.decode(line, code, k, m, log_lines)
    assert bool(np.all(ok)) and np.array_equal(back, cano)  # exact
    # ... hence injective: as many (line, code) pairs as distinct canonical k-mers
    assert len(set(zip(line.tolist(), code.tolist()))) == len(set(cano.tolist()))


def test_the_line_is_the_shipped_index_s():
    """line = index_line_of(minimizer_of(fwd, rc)) of br_amd/csrc/brx_index.hpp, restated by hand for one k-mer"""
    k, m, log_lines = 19, 15, 25
    s = "ACGTTGCATGCCGATAGCT"
    fwd = pack(s)
    rc = int(P.revcomp(np.array([fwd], dtype=np.uint64), k)[0])
    w, mm = k - m + 1, (1 << (2 * m)) - 1
    best = 0xFFFFFFFF
    for j in range(w):  # minimizer_hash_w: the m-mer at offset 2j of fwd against the one at offset 2(W-1-j) of rc
        f, r = (fwd >> (2 * j)) & mm, (rc >> (2 * (w - 1 - j))) & mm
        best = min(best, (min(f, r) * 0x9E3779B1) & 0xFFFFFFFF)
    want = ((best * 0x85EBCA6B) & 0xFFFFFFFF) >> (32 - log_lines)
    line, _ = P.encode(np.array([fwd], dtype=np.uint64), k, m, log_lines)
    assert int(line[0]) == want
