#!/usr/bin/env python3
"""What a line format of the probe index would hold: keys per line, overflow, and line requests per read position, on a
random genome of the bench's size (CPU, numpy; no GPU).  For DESIGN.md section 10 (compact index): the shipped format is
64-byte lines of 7 u64 entries addressed by 15-mer minimizers at k = 19.

usage: python tools/compact_index_sim.py [genome_bases=20000000] [k=19]
"""
import sys
import numpy as np

K1, K2 = np.uint64(0x9E3779B1), np.uint64(0x85EBCA6B)
M32 = np.uint64(0xFFFFFFFF)


def mmer_hashes(g, m):
    """hash of the canonical m-mer starting at every position of g (2-bit codes, complement = xor 2)"""
    n = len(g) - m + 1
    f = np.zeros(n, dtype=np.uint64)
    r = np.zeros(n, dtype=np.uint64)
    for j in range(m):
        b = g[j:j + n].astype(np.uint64)
        f = (f << np.uint64(2)) | b
        r |= (b ^ np.uint64(2)) << np.uint64(2 * j)
    return (np.minimum(f, r) * K1) & M32


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 19
    rng = np.random.default_rng(7)
    g = rng.integers(0, 4, n, dtype=np.uint8)
    nk = n - k + 1
    print(f"genome {n} bases, k = {k}: {nk} k-mer positions (a random genome: practically all distinct)")
    print("m  w  lines  line_bytes slots | index GiB | keys/line mean | lines empty | keys overflowing | line changes per position")
    for m in (15, 13, 11):
        h = mmer_hashes(g, m)
        w = k - m + 1
        mh = h[:nk].copy()
        for j in range(1, w):
            mh = np.minimum(mh, h[j:j + nk])
        gl = (mh * K2) & M32
        changes = float(np.count_nonzero(mh[1:] != mh[:-1])) / (nk - 1)
        for log_lines, line_bytes, slots in ((25, 64, 7), (25, 64, 15), (25, 32, 7), (26, 32, 7), (24, 64, 15)):
            line = (gl >> np.uint64(32 - log_lines)).astype(np.int64)
            cnt = np.bincount(line, minlength=1 << log_lines)
            over = float(np.maximum(cnt - slots, 0).sum()) / nk
            empty = float(np.count_nonzero(cnt == 0)) / (1 << log_lines)
            print(f"{m:2d} {w:2d}  2^{log_lines}  {line_bytes:3d}        {slots:2d}    | {(line_bytes << log_lines) / 2**30:5.2f}     | {nk / (1 << log_lines):5.2f}          | {empty:5.3f}       | {over:7.4f}          | {changes:5.3f}")


if __name__ == "__main__":
    main()
