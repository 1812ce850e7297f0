#!/usr/bin/env python3
"""Prototype (CPU, numpy) of the COMPACT index entry DESIGN.md section 10.1 proposes: once the LINE of the probe index is
known, a solid k-mer is identified by far fewer than 64 bits.

The shipped index (br_amd/csrc/brx_index.hpp) addresses a 64-byte line by the k-mer's strand-symmetric minimizer:
    c    = min over the W = k - m + 1 windows of the canonical m-mer min(f, revcomp(f))   (32-bit value, < 4^m)
    mh   = c * 0x9E3779B1 mod 2^32            (the order the minimum is taken in; a bijection of c)
    g    = mh * 0x85EBCA6B mod 2^32           (index_line_of; another bijection)
    line = g >> (32 - log2(lines))
and stores `canonical >> 1` + 1 in a u64 slot.  Both multipliers are odd, so (line, the low 32 - log2(lines) bits of g)
gives g, hence mh, hence c back.  What is left of the k-mer is where the minimizer sits (its window j < W), which strand
of it the canonical k-mer carries, and the k - m bases around it:
    code = rest | j << R | strand << (R + JB) | flanks << (R + JB + 1)        R = 32 - log2(lines), JB = bits of W - 1
    k = 19, m = 15, 2^25 lines: 7 + 3 + 1 + 8 = 19 bits;  k = 21, m = 16 (2^29 lines): 3 + 3 + 1 + 10 = 17 bits
-- a 32-bit entry with room for the "empty" value and flags, i.e. 32-byte lines with 7 entries, or 64-byte lines with 15.

This file only PINS the encoding: encode() is a function of the canonical k-mer alone (the same for a k-mer and its reverse
complement), and decode(line, code) gives the canonical k-mer back, so (line, code) identifies the key exactly as the
64-bit slot does -- the index stays exact.  tests/test_compact_key_proto.py checks both on random and on adversarial
k-mers (repeated minimizers, palindromic m-mers, ties).  Nothing in the product path uses this yet.

usage: python tools/compact_key_proto.py [k=19] [m=15] [log2_lines=25] [n=1000000]
"""
import sys
import numpy as np

K1 = np.uint64(0x9E3779B1)
K2 = np.uint64(0x85EBCA6B)
M32 = np.uint64(0xFFFFFFFF)


def _inv32(a):
    """inverse of an odd 32-bit number modulo 2^32 (Newton)"""
    x = a
    for _ in range(5):
        x = (x * (2 - a * x)) & 0xFFFFFFFF
    return x


K1_INV = np.uint64(_inv32(0x9E3779B1))
K2_INV = np.uint64(_inv32(0x85EBCA6B))


def revcomp(x, n):
    """reverse complement of n bases packed 2 bits each (A=0, C=1, T=2, G=3: complement = xor 2, as brx_kmer.hpp)"""
    x = np.asarray(x, dtype=np.uint64)
    out = np.zeros_like(x)
    t = x.copy()
    for _ in range(n):
        out = (out << np.uint64(2)) | ((t & np.uint64(3)) ^ np.uint64(2))
        t >>= np.uint64(2)
    return out


def popcount64(x):
    x = np.asarray(x, dtype=np.uint64)
    c = np.zeros(x.shape, dtype=np.uint64)
    t = x.copy()
    for _ in range(64):
        c += t & np.uint64(1)
        t >>= np.uint64(1)
    return c


def canonical(fwd, k):
    """brx_kmer.hpp: canonical(kmer) = popcount odd ? revcomp : kmer (k odd)"""
    fwd = np.asarray(fwd, dtype=np.uint64)
    rc = revcomp(fwd, k)
    return np.where((popcount64(fwd) & np.uint64(1)) == 1, rc, fwd)


def jbits(w):
    return max(1, int(w - 1).bit_length())


def code_bits(k, m, log_lines):
    return (32 - log_lines) + jbits(k - m + 1) + 1 + 2 * (k - m)


def encode(fwd, k, m, log_lines):
    """(line, code) of the k-mers `fwd` (any strand).  Ties between windows go to the lowest window of the CANONICAL k-mer."""
    cano = canonical(fwd, k)
    w = k - m + 1
    mm = np.uint64((1 << (2 * m)) - 1)
    best_h = np.full(cano.shape, 0xFFFFFFFFFF, dtype=np.uint64)
    best_j = np.zeros(cano.shape, dtype=np.uint64)
    best_s = np.zeros(cano.shape, dtype=np.uint64)
    for j in range(w):
        f = (cano >> np.uint64(2 * j)) & mm
        r = revcomp(f, m)
        c = np.minimum(f, r)
        h = (c * K1) & M32
        better = h < best_h  # strict: the lowest window wins a tie
        best_h = np.where(better, h, best_h)
        best_j = np.where(better, np.uint64(j), best_j)
        best_s = np.where(better, (r < f).astype(np.uint64), best_s)
    g = (best_h * K2) & M32
    R = 32 - log_lines
    line = g >> np.uint64(R)
    rest = g & np.uint64((1 << R) - 1)
    # flanks: the k - m bases of the canonical k-mer outside the minimizer's window, low part then high part
    lo = cano & ((np.uint64(1) << (np.uint64(2) * best_j)) - np.uint64(1))
    hi = cano >> (np.uint64(2) * (best_j + np.uint64(m)))
    flanks = (hi << (np.uint64(2) * best_j)) | lo
    JB = jbits(w)
    code = rest | (best_j << np.uint64(R)) | (best_s << np.uint64(R + JB)) | (flanks << np.uint64(R + JB + 1))
    return line, code


def decode(line, code, k, m, log_lines):
    """canonical k-mer of (line, code); also whether the pair is well-formed (the minimizer fits m bases, j < W)"""
    line = np.asarray(line, dtype=np.uint64)
    code = np.asarray(code, dtype=np.uint64)
    w = k - m + 1
    R = 32 - log_lines
    JB = jbits(w)
    rest = code & np.uint64((1 << R) - 1)
    j = (code >> np.uint64(R)) & np.uint64((1 << JB) - 1)
    s = (code >> np.uint64(R + JB)) & np.uint64(1)
    flanks = code >> np.uint64(R + JB + 1)
    g = (line << np.uint64(R)) | rest
    mh = (g * K2_INV) & M32
    c = (mh * K1_INV) & M32
    ok = (c < np.uint64(1 << (2 * m))) & (j < np.uint64(w)) if m < 16 else (j < np.uint64(w))
    mmer = np.where(s == 1, revcomp(c, m), c)
    lo = flanks & ((np.uint64(1) << (np.uint64(2) * j)) - np.uint64(1))
    hi = flanks >> (np.uint64(2) * j)
    cano = (hi << (np.uint64(2) * (j + np.uint64(m)))) | (mmer << (np.uint64(2) * j)) | lo
    return cano, ok


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
    m = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    log_lines = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 1_000_000
    rng = np.random.default_rng(1)
    fwd = rng.integers(0, 1 << (2 * k), n, dtype=np.uint64)
    line, code = encode(fwd, k, m, log_lines)
    line2, code2 = encode(revcomp(fwd, k), k, m, log_lines)
    back, ok = decode(line, code, k, m, log_lines)
    print(f"k={k} m={m} lines=2^{log_lines}: code {code_bits(k, m, log_lines)} bits (max seen {int(code.max()).bit_length()}); "
          f"strand-symmetric {bool(np.all(line == line2) and np.all(code == code2))}; "
          f"decode(encode) == canonical {bool(np.all(back == canonical(fwd, k)) and np.all(ok))}; "
          f"distinct (line, code) pairs {len(set(zip(line.tolist(), code.tolist())))} of {len(set(canonical(fwd, k).tolist()))} distinct k-mers")


if __name__ == "__main__":
    main()
