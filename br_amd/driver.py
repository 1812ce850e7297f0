"""br::run_correction mirror (src/lib.rs:22-139).

Batches of 8192 records (populate_buffer's hard-coded size, src/lib.rs:90 -- `record_buffer_len`
only sizes a Vec there, :84), every method in order, the reverse pass unless two_side
(src/lib.rs:48,110), records written in input order (the serial path's order, src/lib.rs:29-66).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import BinaryIO, Dict, Optional, Sequence

from . import _lib, fasta, hostio
from .correct import Chain, Corrector

RECORD_BATCH = 8192


def run_correction(inputs: Sequence[BinaryIO], outputs: Sequence[BinaryIO], methods: Sequence[Corrector],
                   two_side: bool, record_buffer_len: int = 8192, native: Optional[bool] = None,
                   batch_records: int = 0) -> Dict[str, int]:
    """native (default; BRX_HOST_PIPELINE=0 selects the other): the C++ pipeline of libbrx (brx_run_correction_fd:
    parse / GPU / format on their own threads, batches that fill the GPU).  native=False: the same job record by
    record in Python (fasta.py + Chain.correct_reads), kept as the readable statement of the behaviour and
    compared byte for byte with the native path by the tests.  batch_records (native): records per GPU batch,
    0 = the library's default (32768 or 256 MB; the reference's 8192 is too small to fill the GPU)."""
    if native is None:
        native = os.environ.get("BRX_HOST_PIPELINE", "1") != "0"
    totals = {"records": 0, "bases_in": 0, "bases_out": 0, "batches": 0, "ns_parse": 0, "ns_gpu": 0, "ns_write": 0, "ns_wall": 0}
    if not methods:
        raise ValueError("empty method list")
    solid = methods[0].valid_kmer()
    if native:
        specs = (_lib.Method * len(methods))(*[_lib.Method(*_spec_codes(m)) for m in methods])
        for inp, out in zip(inputs, outputs):
            st = (C.c_uint64 * 8)()
            with hostio.input_fd(inp) as ifd, hostio.output_fd(out) as ofd:
                _lib.check(_lib.lib().brx_run_correction_fd(solid._h, specs, len(methods), two_side, ifd, ofd, batch_records, st))
            for key, v in zip(totals, st):
                totals[key] += int(v)
        return totals
    chain = Chain(solid, [m.spec() for m in methods], two_side=two_side)
    for inp, out in zip(inputs, outputs):
        batch = []
        for rec in fasta.read_records(inp):
            batch.append(rec)
            if len(batch) == RECORD_BATCH:
                _flush(chain, batch, out, totals)
                batch = []
        if batch:
            _flush(chain, batch, out, totals)
    return totals


def _spec_codes(m: Corrector):
    name, confirm, max_search = m.spec()
    return _lib.METHOD_IDS[name], confirm, max_search


def _flush(chain, batch, out, totals) -> None:
    seqs = [r[2] for r in batch]
    corrected = chain.correct_reads(seqs)
    for (name, desc, _), seq in zip(batch, corrected):
        fasta.write_record(out, name, desc, seq)
        totals["bases_out"] += len(seq)
    totals["records"] += len(batch)
    totals["bases_in"] += sum(len(s) for s in seqs)
    totals["batches"] += 1
