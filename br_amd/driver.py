"""br::run_correction mirror (src/lib.rs:22-139).

Batches of 8192 records (populate_buffer's hard-coded size, src/lib.rs:90 -- `record_buffer_len`
only sizes a Vec there, :84), every method in order, the reverse pass unless two_side
(src/lib.rs:48,110), records written in input order (the serial path's order, src/lib.rs:29-66).
"""
from __future__ import annotations

from typing import BinaryIO, Sequence

from . import fasta
from .correct import Chain, Corrector

RECORD_BATCH = 8192


def run_correction(inputs: Sequence[BinaryIO], outputs: Sequence[BinaryIO], methods: Sequence[Corrector],
                   two_side: bool, record_buffer_len: int = 8192) -> None:
    if not methods:
        chain = None
    else:
        solid = methods[0].valid_kmer()
        chain = Chain(solid, [m.spec() for m in methods], two_side=two_side)
    for inp, out in zip(inputs, outputs):
        batch = []
        for rec in fasta.read_records(inp):
            batch.append(rec)
            if len(batch) == RECORD_BATCH:
                _flush(chain, batch, out)
                batch = []
        if batch:
            _flush(chain, batch, out)


def _flush(chain, batch, out) -> None:
    seqs = [r[2] for r in batch]
    corrected = chain.correct_reads(seqs) if chain is not None else seqs
    for (name, desc, _), seq in zip(batch, corrected):
        fasta.write_record(out, name, desc, seq)
