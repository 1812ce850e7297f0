"""br::set mirror: KmerSet trait and the Pcon bitset set, backed by libbrx (HIP).

Reference: src/set.rs:17-23 (trait KmerSet {get, k}, BoxKmerSet), src/set/pcon.rs:13-196
(Pcon::{new, from_pcon_solid, from_fasta}, get/k), src/main.rs:72-115 (build by counting).
"""
from __future__ import annotations

import ctypes as C
import gzip
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np

from . import _lib


def _check_u8(v: int, what: str) -> None:
    if not 0 <= int(v) <= 255:  # u8 in the reference (src/cli.rs:196); ctypes would wrap it silently
        raise ValueError(f"{what}={v} does not fit the reference's u8")


def pack_reads(reads: Sequence[bytes]) -> Tuple[np.ndarray, np.ndarray]:
    """records -> (bases uint8[total], offsets uint64[n+1]): the batch layout of include/brx.h."""
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        offs[1:] = np.cumsum([len(r) for r in reads], dtype=np.uint64)
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8) if reads else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(bases), offs


def seq2bit(seq: bytes) -> int:
    """cocktail::kmer::seq2bit: (ascii >> 1) & 3 per base, first base most significant."""
    k = 0
    for c in seq:
        k = (k << 2) | ((c >> 1) & 3)
    return k


class KmerSet:
    """trait KmerSet (src/set.rs:17-21)."""

    def get(self, kmer: int) -> bool:  # pragma: no cover - interface
        raise NotImplementedError

    def k(self) -> int:  # pragma: no cover - interface
        raise NotImplementedError


class Pcon(KmerSet):
    """set::Pcon: canonical-k-mer bitset resident in HBM (src/set/pcon.rs:13-15)."""

    def __init__(self, handle: int, device: int):
        self._h = C.c_void_p(handle)
        self.device = device

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _lib.lib().brx_set_free(self._h)
                self._h = C.c_void_p(None)
        except Exception:
            pass

    # ---- constructors -----------------------------------------------------------------
    @classmethod
    def new(cls, k: int, device: int = 0) -> "Pcon":
        """Pcon::new(pcon::solid::Solid::new(k)) (src/set/pcon.rs:183-185)."""
        h = C.c_void_p()
        _lib.check(_lib.lib().brx_set_new(k, device, C.byref(h)))
        return cls(h.value, device)

    @classmethod
    def from_pcon_solid(cls, data: bytes, device: int = 0) -> "Pcon":
        """Pcon::from_pcon_solid (src/set/pcon.rs:18-25).  `data` may be gzip (niffler sniffs
        compression at src/cli.rs:415) or the raw [k][bits] stream."""
        if data[:2] == b"\x1f\x8b":
            data = gzip.decompress(data)
        buf = np.frombuffer(data, dtype=np.uint8)
        h = C.c_void_p()
        _lib.check(_lib.lib().brx_set_new_from_solid_bytes(buf.ctypes.data, buf.size, device, C.byref(h)))
        return cls(h.value, device)

    @classmethod
    def from_fasta(cls, reads: Iterable[bytes], k: int, device: int = 0, batch: int = 8192) -> "Pcon":
        """Pcon::from_fasta (src/set/pcon.rs:47-112): presence-only set of every canonical k-mer."""
        s = cls.new(k, device)
        chunk = []
        for r in reads:
            chunk.append(r)
            if len(chunk) == batch:  # populate_buffer(.., 8192), src/set/pcon.rs:82
                s.insert_reads(chunk)
                chunk = []
        if chunk:
            s.insert_reads(chunk)
        return s

    @classmethod
    def from_fasta_file(cls, f, k: int, device: int = 0) -> "Pcon":
        """from_fasta over a FASTA stream (binary file object) through the native host pipeline"""
        from . import hostio
        s = cls.new(k, device)
        st = (C.c_uint64 * 8)()
        with hostio.input_fd(f) as fd:
            _lib.check(_lib.lib().brx_set_insert_fasta_fd(s._h, fd, 0, st))
        return s

    @classmethod
    def from_count(cls, reads: Iterable[bytes], k: int, abundance: int, device: int = 0, batch: int = 8192,
                   strategy: int = _lib.COUNT_AUTO) -> "Pcon":
        """`br fasta -k K -a A` (src/main.rs:72-115): count canonical k-mers (u8, saturating),
        solid iff count > abundance."""
        c = Counter(k, device, strategy)
        chunk = []
        for r in reads:
            chunk.append(r)
            if len(chunk) == batch:  # count_fasta(reader, 8192), src/main.rs:74
                c.add_reads(chunk)
                chunk = []
        if chunk:
            c.add_reads(chunk)
        return c.finish(abundance)

    # ---- KmerSet ----------------------------------------------------------------------
    def get(self, kmer: int) -> bool:
        return bool(_lib.lib().brx_set_get(self._h, kmer))

    def k(self) -> int:
        return int(_lib.lib().brx_set_k(self._h))

    def get_many(self, kmers: Sequence[int]) -> np.ndarray:
        km = np.ascontiguousarray(np.asarray(kmers, dtype=np.uint64))
        out = np.zeros(km.size, dtype=np.uint8)
        _lib.check(_lib.lib().brx_set_get_batch(self._h, km.ctypes.data, km.size, out.ctypes.data))
        return out.astype(bool)

    # ---- Solid-level helpers ----------------------------------------------------------
    def set(self, kmer: int, value: bool = True) -> None:
        _lib.check(_lib.lib().brx_set_set(self._h, kmer, value))

    def insert_reads(self, reads: Sequence[bytes]) -> None:
        bases, offs = pack_reads(reads)
        _lib.check(_lib.lib().brx_set_insert_batch(self._h, bases.ctypes.data, offs.ctypes.data, len(reads)))

    def to_solid_bytes(self) -> bytes:
        n = C.c_size_t(0)
        L = _lib.lib()
        L.brx_set_export_solid_bytes(self._h, None, 0, C.byref(n))
        buf = np.zeros(n.value, dtype=np.uint8)
        _lib.check(L.brx_set_export_solid_bytes(self._h, buf.ctypes.data, buf.size, C.byref(n)))
        return buf.tobytes()

    def export_bits(self) -> np.ndarray:
        """the packed Lsb0 bit array (without the k byte) as uint8[]."""
        n = C.c_size_t(0)
        L = _lib.lib()
        L.brx_set_export_solid_bytes(self._h, None, 0, C.byref(n))
        buf = np.zeros(n.value, dtype=np.uint8)
        _lib.check(L.brx_set_export_solid_bytes(self._h, buf.ctypes.data, buf.size, C.byref(n)))
        return buf[1:]

    def is_sparse(self) -> bool:
        """k >= 21: no bit vector, the solid k-mers live in the key list / probe index only (include/brx.h)"""
        return bool(_lib.lib().brx_set_sparse(self._h))

    def bits_state(self) -> int:
        """0 bit vector current, 1 lazy (materialised on demand), 2 sparse (include/brx.h brx_set_bits_state)"""
        return int(_lib.lib().brx_set_bits_state(self._h))

    def popcount(self) -> int:
        n = C.c_uint64(0)
        _lib.check(_lib.lib().brx_set_popcount(self._h, C.byref(n)))
        return n.value

    def device_bits(self) -> Tuple[int, int]:
        p, n = C.c_void_p(), C.c_uint64(0)
        _lib.check(_lib.lib().brx_set_device_bits(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def n_hashes(self) -> int:
        return max(1 << (2 * self.k() - 1), 32)

    def extract_keys_device(self, first_hash: int, n_hashes: int, d_out: int, cap: int, stream: Optional[int] = None) -> int:
        n = C.c_uint64(0)
        _lib.check(_lib.lib().brx_set_extract_keys_device(self._h, first_hash, n_hashes, d_out, cap, C.byref(n), stream))
        return n.value

    def or_keys_device(self, d_keys: int, n: int, stream: Optional[int] = None) -> None:
        _lib.check(_lib.lib().brx_set_or_keys_device(self._h, d_keys, n, stream))

    # ---- probe index (include/brx.h: an HBM-locality copy of the set, no reference counterpart) ----
    def index_build(self, m: int = 0, log2_lines: int = 0, stream: Optional[int] = None) -> dict:
        _lib.check(_lib.lib().brx_set_index_build(self._h, m, log2_lines, stream))
        return self.index_info()

    def index_build_from_keys_device(self, d_keys: int, n: int, m: int = 0, log2_lines: int = 0,
                                     stream: Optional[int] = None) -> dict:
        _lib.check(_lib.lib().brx_set_index_build_from_keys_device(self._h, d_keys, n, m, log2_lines, stream))
        return self.index_info()

    def keylist_device(self, stream: Optional[int] = None):
        """(device pointer, n) of the solid-hash list a partitioned finish left with the set, or None"""
        ptr, n = C.c_void_p(), C.c_uint64(0)
        _lib.check(_lib.lib().brx_set_keylist_device(self._h, C.byref(ptr), C.byref(n), stream))
        return (ptr.value, n.value) if ptr.value else None

    def fingerprint(self, stream: Optional[int] = None) -> Tuple[int, int, int]:
        """(members, sum of their hashes mod 2^64, sum of the squares mod 2^64): the same for the same set whatever holds
        it -- what the ranks of a multi-GPU job compare after the exchange"""
        out = (C.c_uint64 * 3)()
        _lib.check(_lib.lib().brx_set_fingerprint(self._h, out, stream))
        return int(out[0]), int(out[1]), int(out[2])

    def index_drop(self) -> None:
        _lib.check(_lib.lib().brx_set_index_drop(self._h))

    def index_info(self) -> dict:
        v = (C.c_uint64 * 8)()
        _lib.check(_lib.lib().brx_set_index_info(self._h, v))
        return {"valid": bool(v[0]), "m": int(v[1]), "log2_lines": int(v[2]), "keys": int(v[3]),
                "overflow_keys": int(v[4]), "bytes": int(v[5]), "wanted": bool(v[6]), "keylist": bool(v[7])}

    def get_batch_indexed(self, kmers):
        """(answers, n_fallback): `get_batch` through the probe index."""
        ks = np.ascontiguousarray(kmers, dtype=np.uint64)
        out = np.empty(len(ks), dtype=np.uint8)
        fb = C.c_uint64(0)
        _lib.check(_lib.lib().brx_set_get_batch_indexed(self._h, ks.ctypes.data, len(ks), out.ctypes.data, C.byref(fb)))
        return out.astype(bool), fb.value


class Counter:
    """pcon::counter::Counter<u8> as used by `br fasta` (src/main.rs:73-78)."""

    def __init__(self, k: int, device: int = 0, strategy: int = _lib.COUNT_AUTO):
        self._h = C.c_void_p()
        self.k = k
        self.device = device
        _lib.check(_lib.lib().brx_set_count_begin(k, device, strategy, C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _lib.lib().brx_counter_free(self._h)
                self._h = C.c_void_p(None)
        except Exception:
            pass

    def add_reads(self, reads: Sequence[bytes]) -> None:
        bases, offs = pack_reads(reads)
        self.add_batch(bases, offs)

    def add_batch(self, bases: np.ndarray, offsets: np.ndarray) -> None:
        _lib.check(_lib.lib().brx_set_count_add_batch(self._h, bases.ctypes.data, offsets.ctypes.data,
                                                      offsets.size - 1))

    def count_fasta(self, f) -> dict:
        """Counter::count_fasta(reader, record_buffer) (src/main.rs:73-78) through the native host pipeline
        (brx_count_fasta_fd): every record of the FASTA stream `f` (a binary file object)."""
        from . import hostio
        st = (C.c_uint64 * 8)()
        with hostio.input_fd(f) as fd:
            _lib.check(_lib.lib().brx_count_fasta_fd(self._h, fd, 0, st))
        return {"records": int(st[0]), "bases_in": int(st[1]), "batches": int(st[3]), "ns_parse": int(st[4]),
                "ns_gpu": int(st[5]), "ns_wall": int(st[7])}

    def add_batch_device(self, d_bases: int, d_offsets: int, n_reads: int, total_bases: int,
                         stream: Optional[int] = None) -> None:
        _lib.check(_lib.lib().brx_set_count_add_batch_device(self._h, d_bases, d_offsets, n_reads, total_bases,
                                                             stream))

    def clamp(self, cap: int, stream: Optional[int] = None) -> None:
        _lib.check(_lib.lib().brx_counter_clamp(self._h, cap, stream))

    def device_counts(self) -> Tuple[int, int]:
        p, n = C.c_void_p(), C.c_uint64(0)
        _lib.check(_lib.lib().brx_counter_device_counts(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def load_counts(self, first: int, counts: bytes) -> None:
        """overwrite counters [first, first + len(counts)) of a dense counter (`br count`: a table counted elsewhere)"""
        _lib.check(_lib.lib().brx_counter_load_counts(self._h, first, bytes(counts), len(counts)))

    @classmethod
    def from_count_stream(cls, f, device: int = 0, chunk: int = 64 << 20) -> "Counter":
        """pcon::counter::Counter::<u8>::from_stream (src/main.rs:59-61): a count table written by `pcon count`.
        Layout as restated from pcon's published serializer -- one byte k, then the 2^(2k-1) u8 counters of the
        canonical hashes in order (the reference builds pcon with `count_u8`, Cargo.toml:23); the reference holds no
        count-file fixture, so the layout is UNPINNED.  A short stream is an error, like read_exact's."""
        head = f.read(1)
        if len(head) != 1:
            raise _lib.BrxError(-5, "count stream: empty")
        k = head[0]
        if not 1 <= k <= 19:
            raise _lib.BrxError(-1, f"count stream: k={k} (a dense u8 table fits one GPU for k <= 19)")
        cnt = cls(k, device, _lib.COUNT_DENSE)
        total, pos = 1 << (2 * k - 1), 0
        while pos < total:
            buf = f.read(min(chunk, total - pos))
            if not buf:
                raise _lib.BrxError(-5, f"count stream: ends after {pos} of {total} counters")
            cnt.load_counts(pos, buf)
            pos += len(buf)
        return cnt

    def spectrum(self, stream: Optional[int] = None) -> np.ndarray:
        """pcon::spectrum::Spectrum::from_count: uint64[256] histogram of the counts (255 = 255 or more).
        Either strategy; the counter is left as it was, so `finish(threshold)` can follow."""
        h = np.zeros(256, dtype=np.uint64)
        _lib.check(_lib.lib().brx_counter_spectrum(self._h, h.ctypes.data_as(C.POINTER(C.c_uint64)), stream))
        return h

    def l1_view(self) -> Tuple[int, int, int, int]:
        """(d_keys, d_l1off, n_buckets, n_keys) of the single counted batch (partitioned strategy)."""
        pk, po, nb, nk = C.c_void_p(), C.c_void_p(), C.c_uint32(0), C.c_uint64(0)
        _lib.check(_lib.lib().brx_counter_l1_view(self._h, C.byref(pk), C.byref(po), C.byref(nb), C.byref(nk)))
        return pk.value or 0, po.value or 0, nb.value, nk.value

    def add_partitioned_device(self, d_keys: int, d_l1off: int, n_keys: int) -> None:
        _lib.check(_lib.lib().brx_counter_add_partitioned_device(self._h, d_keys, d_l1off, n_keys))

    def reset(self, stream: Optional[int] = None) -> None:
        _lib.check(_lib.lib().brx_counter_reset(self._h, stream))

    def finish_into(self, abundance: int, dst: "Pcon", stream: Optional[int] = None) -> None:
        _check_u8(abundance, "abundance")
        _lib.check(_lib.lib().brx_set_count_finish_into(self._h, abundance, stream, dst._h))

    def finish(self, abundance: int, stream: Optional[int] = None) -> Pcon:
        """Solid::from_count(k, counts, abundance): solid iff count > abundance."""
        _check_u8(abundance, "abundance")
        h = C.c_void_p()
        _lib.check(_lib.lib().brx_set_count_finish(self._h, abundance, stream, C.byref(h)))
        return Pcon(h.value, self.device)
