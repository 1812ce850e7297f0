"""ctypes front-end of the CPU oracle (oracle/br_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importers allowed: tests/, __graft_entry__.smoke(),
bench.py's cpu_baseline leg.  br_amd/ never imports this module.

Mirrors the reference's seams so the parity tests read like the reference's own
tests: Solid (pcon::solid::Solid as used in src/correct/*/tests), Corrector
(src/correct/mod.rs:44-108) built per method like build_methods
(src/lib.rs:141-164), correct_batch = run_correction's per-record body
(src/lib.rs:42-55).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbr_oracle.so")

METHODS = {"one": 0, "two": 1, "graph": 2, "greedy": 3, "gap_size": 4}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "br_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libbr_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    u8p, u64p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.c_void_p
    sig = {
        "bro_nuc2bit": (C.c_uint64, [C.c_uint8]),
        "bro_bit2nuc": (C.c_uint8, [C.c_uint64]),
        "bro_seq2bit": (C.c_uint64, [C.c_char_p, C.c_size_t]),
        "bro_revcomp": (C.c_uint64, [C.c_uint64, C.c_int]),
        "bro_canonical": (C.c_uint64, [C.c_uint64, C.c_int]),
        "bro_hash": (C.c_uint64, [C.c_uint64, C.c_int]),
        "bro_mask": (C.c_uint64, [C.c_int]),
        "bro_solid_nbytes": (C.c_uint64, [C.c_int]),
        "bro_solid_new": (vp, [C.c_int]),
        "bro_solid_free": (None, [vp]),
        "bro_solid_k": (C.c_int, [vp]),
        "bro_solid_bits": (u8p, [vp]),
        "bro_solid_set": (None, [vp, C.c_uint64, C.c_int]),
        "bro_solid_get": (C.c_int, [vp, C.c_uint64]),
        "bro_solid_popcount": (C.c_uint64, [vp]),
        "bro_solid_set_seq": (None, [vp, C.c_char_p, C.c_size_t]),
        "bro_solid_from_count": (vp, [C.c_int, vp, C.c_uint8]),
        "bro_solid_new_sparse": (vp, [C.c_int, vp, C.c_size_t]),
        "bro_solid_from_bytes": (vp, [C.c_char_p, C.c_size_t]),
        "bro_solid_wrap": (vp, [C.c_int, vp]),
        "bro_solid_unwrap": (None, [vp]),
        "bro_solid_extend": (None, [vp, vp]),
        "bro_count_nbytes": (C.c_uint64, [C.c_int]),
        "bro_count_seq": (None, [vp, C.c_int, vp, C.c_size_t]),
        "bro_hashes_seq": (C.c_size_t, [C.c_int, vp, C.c_size_t, vp]),
        "bro_corrector_new": (vp, [vp, C.c_int, C.c_int, C.c_int]),
        "bro_corrector_free": (None, [vp]),
        "bro_corrector_stats": (None, [vp, u64p]),
        "bro_alt_nucs": (C.c_int, [vp, C.c_uint64, u64p]),
        "bro_correct": (vp, [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
        "bro_correct_record": (vp, [C.POINTER(vp), C.c_int, C.c_int, C.c_char_p, C.c_size_t,
                                    C.POINTER(C.c_size_t)]),
        "bro_correct_batch": (vp, [C.POINTER(vp), C.c_int, C.c_int, vp, vp, C.c_uint32, vp]),
        "bro_free": (None, [vp]),
        "bro_solid_mask": (None, [vp, vp, C.c_size_t, vp]),
        "bro_bio_global": (C.c_size_t, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, vp]),
        "bro_greedy_audit_enable": (None, [C.c_int]),
        "bro_greedy_audit_get": (None, [u64p]),
        "bro_greedy_tie_check": (C.c_uint64, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t, vp, C.c_size_t,
                                              C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "bro_correct_batch_mt": (C.c_uint64, [vp, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp,
                                              C.c_uint32, C.c_int, vp, u64p]),
        "bro_correct_batch_mt_check": (C.c_uint64, [vp, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp,
                                                    C.c_uint32, C.c_int, vp, u64p, vp, vp, u64p, C.POINTER(C.c_uint32)]),
        "bro_count_batch_mt": (None, [vp, C.c_int, vp, vp, C.c_uint32, C.c_int]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def seq2bit(s: bytes) -> int:
    return lib().bro_seq2bit(s, len(s))


def revcomp(kmer: int, k: int) -> int:
    return lib().bro_revcomp(kmer, k)


def canonical(kmer: int, k: int) -> int:
    return lib().bro_canonical(kmer, k)


def khash(kmer: int, k: int) -> int:
    return lib().bro_hash(kmer, k)


class Solid:
    """pcon::solid::Solid restated (bitset of canonical k-mers, Lsb0)."""

    def __init__(self, k: int, _h=None, _keep=None):
        self._L = lib()
        self._h = _h if _h is not None else self._L.bro_solid_new(k)
        if not self._h:
            raise MemoryError("solid alloc failed")
        self._wrapped = _keep is not None
        self._keep = _keep

    def __del__(self):
        try:
            if self._h:
                if self._wrapped:
                    self._L.bro_solid_unwrap(self._h)
                else:
                    self._L.bro_solid_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def k(self) -> int:
        return self._L.bro_solid_k(self._h)

    def set(self, kmer: int, val: bool = True) -> None:
        self._L.bro_solid_set(self._h, kmer, 1 if val else 0)

    def get(self, kmer: int) -> bool:
        return bool(self._L.bro_solid_get(self._h, kmer))

    def set_seq(self, seq: bytes) -> None:
        """set every forward k-mer of seq (Tokenizer loop of the reference tests)."""
        self._L.bro_solid_set_seq(self._h, seq, len(seq))

    def popcount(self) -> int:
        return self._L.bro_solid_popcount(self._h)

    def bits(self) -> np.ndarray:
        n = self._L.bro_solid_nbytes(self.k)
        p = self._L.bro_solid_bits(self._h)
        return np.ctypeslib.as_array(p, shape=(n,))

    def to_bytes(self) -> bytes:
        return bytes([self.k]) + self.bits().tobytes()

    @classmethod
    def from_bytes(cls, buf: bytes) -> "Solid":
        L = lib()
        h = L.bro_solid_from_bytes(buf, len(buf))
        if not h:
            raise ValueError("bad .solid stream")
        return cls(buf[0], _h=h)

    @classmethod
    def from_count(cls, k: int, counts: np.ndarray, abundance: int) -> "Solid":
        L = lib()
        counts = np.ascontiguousarray(counts, dtype=np.uint8)
        assert counts.size == L.bro_count_nbytes(k)
        h = L.bro_solid_from_count(k, counts.ctypes.data, abundance)
        return cls(k, _h=h)

    @classmethod
    def sparse_from_count(cls, k: int, reads, abundance: int) -> "Solid":
        """the same set as from_count(k, count_reads(k, reads), abundance), held as a sorted array of canonical
        hashes instead of 2^(2k-1) bits (k = 21 would need 256 GiB): count every canonical k-mer of every read
        (saturating at 255 like the u8 counter), keep those with count > abundance."""
        parts = [hashes(k, r) for r in reads if len(r) >= k]
        allh = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint64)
        uniq, cnt = np.unique(allh, return_counts=True)
        keep = np.ascontiguousarray(uniq[np.minimum(cnt, 255) > abundance], dtype=np.uint64)
        h = lib().bro_solid_new_sparse(k, keep.ctypes.data, keep.size)
        return cls(k, _h=h)

    @classmethod
    def wrap(cls, k: int, bits: np.ndarray) -> "Solid":
        """zero-copy view over an existing Lsb0 bit array (e.g. a 16 GiB GPU export)."""
        L = lib()
        assert bits.dtype == np.uint8 and bits.flags.c_contiguous
        assert bits.size == L.bro_solid_nbytes(k)
        h = L.bro_solid_wrap(k, bits.ctypes.data)
        return cls(k, _h=h, _keep=bits)

    def extend(self, other: "Solid") -> None:
        self._L.bro_solid_extend(self._h, other._h)

    def mask(self, seq: bytes) -> np.ndarray:
        n = max(len(seq) - self.k + 1, 0)
        out = np.zeros((n + 7) // 8, dtype=np.uint8)
        buf = np.frombuffer(seq, dtype=np.uint8)
        if n:
            self._L.bro_solid_mask(self._h, buf.ctypes.data, len(seq), out.ctypes.data)
        return out


def count_reads(k: int, reads: Iterable[bytes]) -> np.ndarray:
    """pcon Counter<u8>::count_fasta restated: dense u8 table, saturating."""
    L = lib()
    counts = np.zeros(L.bro_count_nbytes(k), dtype=np.uint8)
    for r in reads:
        b = np.frombuffer(r, dtype=np.uint8)
        L.bro_count_seq(counts.ctypes.data, k, b.ctypes.data, len(r))
    return counts


def hashes(k: int, read: bytes) -> np.ndarray:
    L = lib()
    out = np.zeros(max(len(read) - k + 1, 0), dtype=np.uint64)
    if out.size:
        b = np.frombuffer(read, dtype=np.uint8)
        L.bro_hashes_seq(k, b.ctypes.data, len(read), out.ctypes.data)
    return out


class Corrector:
    """One corrector of build_methods (src/lib.rs:141-164)."""

    def __init__(self, solid: Solid, method: str, confirm: int = 5, max_search: int = 7):
        self._L = lib()
        self.solid = solid
        self.method = method
        self._h = self._L.bro_corrector_new(solid._h, METHODS[method], confirm, max_search)

    def __del__(self):
        try:
            if self._h:
                self._L.bro_corrector_free(self._h)
                self._h = None
        except Exception:
            pass

    def correct(self, seq: bytes) -> bytes:
        n = C.c_size_t(0)
        p = self._L.bro_correct(self._h, seq, len(seq), C.byref(n))
        out = C.string_at(p, n.value)
        self._L.bro_free(p)
        return out

    def stats(self) -> dict:
        a = (C.c_uint64 * 9)()
        self._L.bro_corrector_stats(self._h, a)
        names = ["positions", "triggers", "fixes", "fix_i", "fix_s", "fix_d", "rej_alts", "rej_noscen",
                 "rej_multi"]
        return dict(zip(names, [int(v) for v in a]))


def build_methods(solid: Solid, methods: Sequence[str], confirm: int = 5, max_search: int = 7) -> List[Corrector]:
    return [Corrector(solid, m, confirm, max_search) for m in methods]


def correct_record(methods: Sequence[Corrector], seq: bytes, two_side: bool = False) -> bytes:
    L = lib()
    arr = (C.c_void_p * len(methods))(*[m._h for m in methods])
    n = C.c_size_t(0)
    p = L.bro_correct_record(arr, len(methods), 1 if two_side else 0, seq, len(seq), C.byref(n))
    out = C.string_at(p, n.value)
    L.bro_free(p)
    return out


def correct_batch(methods: Sequence[Corrector], bases: np.ndarray, offsets: np.ndarray,
                  two_side: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """bases: uint8[total]; offsets: uint64[n+1].  Returns (out_bases, out_offsets)."""
    L = lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    arr = (C.c_void_p * len(methods))(*[m._h for m in methods])
    out_off = np.zeros(n + 1, dtype=np.uint64)
    p = L.bro_correct_batch(arr, len(methods), 1 if two_side else 0, bases.ctypes.data, offsets.ctypes.data,
                            n, out_off.ctypes.data)
    total = int(out_off[-1])
    out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(total, 1),))[:total].copy()
    L.bro_free(p)
    return out, out_off


def correct_batch_mt(solid: Solid, methods: Sequence[str], bases: np.ndarray, offsets: np.ndarray, confirm: int = 5,
                     max_search: int = 7, two_side: bool = False, threads: int = 1):
    """the chain over every read on `threads` pthreads (bench.py's CPU baseline): returns (corrected lengths per
    read, total corrected bytes, fixes); the corrected bytes themselves are dropped."""
    lens, total, fixes, _, _ = correct_batch_mt_check(solid, methods, bases, offsets, None, None, confirm, max_search, two_side,
                                                      threads)
    return lens, total, fixes


def correct_batch_mt_check(solid: Solid, methods: Sequence[str], bases: np.ndarray, offsets: np.ndarray,
                           expect: Optional[np.ndarray], expect_off: Optional[np.ndarray], confirm: int = 5,
                           max_search: int = 7, two_side: bool = False, threads: int = 1):
    """correct_batch_mt, and every corrected read is compared on the way with somebody else's answer (expect: uint8,
    expect_off: uint64[n+1]): returns (lens, total, fixes, reads that differ, the lowest of them or None)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = offsets.size - 1
    ids = (C.c_int * len(methods))(*[METHODS[m] for m in methods])
    lens = np.zeros(max(n, 1), dtype=np.uint64)
    fixes = C.c_uint64(0)
    bad = C.c_uint64(0)
    first = C.c_uint32(0xffffffff)
    ep = eo = None
    if expect_off is not None:
        expect = np.ascontiguousarray(expect, dtype=np.uint8)
        expect_off = np.ascontiguousarray(expect_off, dtype=np.uint64)
        assert expect_off.size == n + 1 and int(expect_off[-1]) <= expect.size
        ep, eo = expect.ctypes.data, expect_off.ctypes.data
    total = lib().bro_correct_batch_mt_check(solid._h, ids, len(methods), confirm, max_search, 1 if two_side else 0,
                                             bases.ctypes.data, offsets.ctypes.data, n, threads, lens.ctypes.data,
                                             C.byref(fixes), ep, eo, C.byref(bad), C.byref(first))
    return lens[:n], int(total), int(fixes.value), int(bad.value), (None if first.value == 0xffffffff else int(first.value))


def count_reads_mt(k: int, bases: np.ndarray, offsets: np.ndarray, threads: int = 1) -> np.ndarray:
    """count_reads on `threads` pthreads (saturating atomic u8 cells): same table"""
    L = lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    counts = np.zeros(L.bro_count_nbytes(k), dtype=np.uint8)
    L.bro_count_batch_mt(counts.ctypes.data, k, bases.ctypes.data, offsets.ctypes.data, offsets.size - 1, threads)
    return counts


AUDIT_FIELDS = ["calls", "multi", "ambiguous", "capped", "restated_not_optimal", "triggers", "triggers_amb", "fixes",
                "fixes_amb", "max_sequences"]


def greedy_audit(enable: bool) -> None:
    """switch the tie-break audit of Greedy's alignments on (counters zeroed) or off; single-threaded use only"""
    lib().bro_greedy_audit_enable(1 if enable else 0)


def greedy_audit_counters() -> dict:
    a = (C.c_uint64 * 10)()
    lib().bro_greedy_audit_get(a)
    return dict(zip(AUDIT_FIELDS, [int(v) for v in a]))


def greedy_tie_check(x: bytes, y: bytes, nb: int):
    """(number of optimal operation sequences of global(x, y), their greedy.rs:66-86 results differ, the restated
    traceback is one of them)"""
    ops = np.zeros(len(x) + len(y) + 2, dtype=np.uint8)
    n = lib().bro_bio_global(x, len(x), y, len(y), ops.ctypes.data)
    d, f = C.c_int(0), C.c_int(0)
    cnt = lib().bro_greedy_tie_check(x, len(x), y, len(y), nb, ops.ctypes.data, n, C.byref(d), C.byref(f))
    return int(cnt), bool(d.value), bool(f.value)


def alt_nucs(solid: Solid, kmer: int) -> List[int]:
    a = (C.c_uint64 * 4)()
    n = lib().bro_alt_nucs(solid._h, kmer, a)
    return [int(a[i]) for i in range(n)]


def bio_global(x: bytes, y: bytes) -> str:
    """rust-bio global alignment ops as a string of M/X/D/I (Match/Subst/Del/Ins)."""
    ops = np.zeros(len(x) + len(y) + 2, dtype=np.uint8)
    n = lib().bro_bio_global(x, len(x), y, len(y), ops.ctypes.data)
    return "".join("MXDI"[int(o)] for o in ops[:n])
