"""br::correct mirror: trait Corrector and the five correctors, executed by libbrx (HIP).

Reference: src/correct/mod.rs:44-108 (trait Corrector), src/correct/exist/one.rs (One),
exist/two.rs (Two), graph.rs (Graph), greedy.rs (Greedy), gap_size.rs (GapSize);
build_methods at src/lib.rs:141-164.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .set import Pcon, pack_reads


class Chain:
    """The Vec<Box<dyn Corrector>> of build_methods plus the two_side flag, i.e. everything
    run_correction's per-record closure captures (src/lib.rs:93-128)."""

    def __init__(self, solid: Pcon, methods: Sequence[Tuple[str, int, int]], two_side: bool = False):
        self.solid = solid  # keep the set alive (the reference borrows it: src/lib.rs:143)
        arr = (_lib.Method * max(len(methods), 1))()
        for i, (name, confirm, max_search) in enumerate(methods):
            if not (0 <= int(confirm) <= 255 and 0 <= int(max_search) <= 255):
                raise ValueError(f"confirm={confirm} / max_search={max_search}: u8 in the reference (src/cli.rs:46,50)")
            arr[i].method = _lib.METHOD_IDS[name]
            arr[i].confirm = confirm
            arr[i].max_search = max_search
        self._h = C.c_void_p()
        _lib.check(_lib.lib().brx_chain_new(solid._h, arr, len(methods), two_side, C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                _lib.lib().brx_chain_free(self._h)
                self._h = C.c_void_p(None)
        except Exception:
            pass

    def correct_batch(self, bases: np.ndarray, offsets: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        L = _lib.lib()
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        ob, oo = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint64)()
        _lib.check(L.brx_chain_correct_batch(self._h, bases.ctypes.data, offsets.ctypes.data, n, C.byref(ob),
                                             C.byref(oo)))
        try:
            out_off = np.ctypeslib.as_array(oo, shape=(n + 1,)).copy()
            total = int(out_off[-1])
            out = np.ctypeslib.as_array(ob, shape=(max(total, 1),))[:total].copy()
        finally:
            L.brx_buf_free(ob)
            L.brx_buf_free(oo)
        return out, out_off

    def correct_batch_async(self, bases: np.ndarray, offsets: np.ndarray) -> None:
        """brx_chain_correct_batch_async: starts the batch and returns; `correct_batch_wait` collects it.  One batch in
        flight per chain -- a host overlaps batches by rotating over two or three chains of the same set."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._inflight = (bases, offsets)  # the library reads them until _wait returns
        _lib.check(_lib.lib().brx_chain_correct_batch_async(self._h, bases.ctypes.data, offsets.ctypes.data, offsets.size - 1))

    def correct_batch_wait(self) -> Tuple[np.ndarray, np.ndarray]:
        L = _lib.lib()
        ob, oo = C.POINTER(C.c_uint8)(), C.POINTER(C.c_uint64)()
        try:
            _lib.check(L.brx_chain_correct_batch_wait(self._h, C.byref(ob), C.byref(oo)))
            n = self._inflight[1].size - 1
            out_off = np.ctypeslib.as_array(oo, shape=(n + 1,)).copy()
            total = int(out_off[-1])
            out = np.ctypeslib.as_array(ob, shape=(max(total, 1),))[:total].copy()
        finally:
            self._inflight = None
            L.brx_buf_free(ob)
            L.brx_buf_free(oo)
        return out, out_off

    def correct_reads(self, reads: Sequence[bytes]) -> List[bytes]:
        bases, offs = pack_reads(reads)
        out, oo = self.correct_batch(bases, offs)
        return [out[int(oo[i]):int(oo[i + 1])].tobytes() for i in range(len(reads))]

    def correct_batch_device(self, d_bases: int, d_offsets: int, n_reads: int, total_bases: int, d_out: int,
                             out_cap: int, d_out_offsets: int, stream: Optional[int] = None) -> int:
        tot = C.c_uint64(0)
        _lib.check(_lib.lib().brx_chain_correct_batch_device(self._h, d_bases, d_offsets, n_reads, total_bases,
                                                             d_out, out_cap, d_out_offsets, C.byref(tot), stream))
        return tot.value

    def last_stats(self) -> dict:
        a = (C.c_uint64 * 8)()
        _lib.check(_lib.lib().brx_chain_last_stats(self._h, a))
        return {"rounds": a[0], "probes": a[1], "triggers": a[2], "fixes": a[3], "overflow_retries": a[4],
                "slot_overflow_reads": a[5], "walk_list_overflows": a[6], "lane_units": a[7] & 0xffffffff,
                "lane_redone_reads": (a[7] >> 32) & 0xffffff, "lane_unwritten_units": a[7] >> 56}


class Corrector:
    """trait Corrector (src/correct/mod.rs:44-108): correct(seq) = ONE forward scan."""

    method = ""

    def __init__(self, valid_kmer: Pcon, confirm: int = 0, max_search: int = 0):
        self._valid = valid_kmer
        self.confirm = confirm
        self.max_search = max_search
        self._chain = Chain(valid_kmer, [(self.method, confirm, max_search)], two_side=True)

    def valid_kmer(self) -> Pcon:
        return self._valid

    def k(self) -> int:
        return self._valid.k()

    def correct(self, seq: bytes) -> bytes:
        return self._chain.correct_reads([bytes(seq)])[0]

    def spec(self) -> Tuple[str, int, int]:
        return (self.method, self.confirm, self.max_search)


class One(Corrector):
    """One::new(&set, c) (src/correct/exist/mod.rs:89-95, one.rs:74)."""
    method = "one"

    def __init__(self, valid_kmer: Pcon, c: int):
        super().__init__(valid_kmer, confirm=c)


class Two(Corrector):
    """Two::new(&set, c) (src/correct/exist/two.rs:328)."""
    method = "two"

    def __init__(self, valid_kmer: Pcon, c: int):
        super().__init__(valid_kmer, confirm=c)


class Graph(Corrector):
    """Graph::new(&set) (src/correct/graph.rs:34-36)."""
    method = "graph"

    def __init__(self, valid_kmer: Pcon):
        super().__init__(valid_kmer)


class Greedy(Corrector):
    """Greedy::new(&set, max_search, nb_validate) (src/correct/greedy.rs:48-54)."""
    method = "greedy"

    def __init__(self, valid_kmer: Pcon, max_search: int, nb_validate: int):
        super().__init__(valid_kmer, confirm=nb_validate, max_search=max_search)


class GapSize(Corrector):
    """GapSize::new(&set, c) (src/correct/gap_size.rs:36-42)."""
    method = "gap_size"

    def __init__(self, valid_kmer: Pcon, c: int):
        super().__init__(valid_kmer, confirm=c)


def build_methods(params: Sequence[str], solid: Pcon, confirm: int, max_search: int) -> List[Corrector]:
    """br::build_methods (src/lib.rs:141-164): order kept, duplicates allowed."""
    out: List[Corrector] = []
    for m in params:
        m = m.replace("-", "_").lower()
        if m == "one":
            out.append(One(solid, confirm))
        elif m == "two":
            out.append(Two(solid, confirm))
        elif m == "graph":
            out.append(Graph(solid))
        elif m == "greedy":
            out.append(Greedy(solid, max_search, confirm))
        elif m == "gap_size":
            out.append(GapSize(solid, confirm))
        else:
            raise ValueError(f"unknown correction method {m!r}")
    return out
