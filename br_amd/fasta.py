"""Minimal FASTA reader/writer with the behaviour run_correction relies on.

Reference: noodles::fasta::{Reader, Writer} as used at src/lib.rs:30-31,57-60,80-81,123-131
(noodles 0.74 / noodles-fasta 0.38, not vendored).  Restated from the crate's documented
behaviour; UNPINNED by the reference's tests (tests/br.rs never reads corr.fasta):
  reader: '>' definition line, name = up to the first ASCII whitespace, the rest (trimmed) is the
          description; sequence = following lines up to the next '>' with line ends removed;
          a malformed record ends the stream silently (`while let Some(Ok(record))`, src/lib.rs:35).
  writer: '>name[ description]\\n' then the sequence wrapped at 80 columns.
"""
from __future__ import annotations

from typing import BinaryIO, Iterator, List, Optional, Tuple

LINE_BASES = 80  # noodles fasta::Writer default line_base_count

Record = Tuple[bytes, Optional[bytes], bytes]  # (name, description, sequence)


def read_records(f: BinaryIO) -> Iterator[Record]:
    name = desc = None
    seq: List[bytes] = []
    started = False
    for raw in f:
        line = raw.rstrip(b"\r\n")
        if line.startswith(b">"):
            if started:
                yield name, desc, b"".join(seq)
            body = line[1:]
            parts = body.split(None, 1)
            if not parts or body[:1].isspace():
                return  # missing name: parse error -> stream ends (src/lib.rs:35)
            name = parts[0]
            desc = parts[1].strip() if len(parts) > 1 else None
            seq = []
            started = True
        else:
            if not started:
                return  # data before the first definition: parse error
            seq.append(line)
    if started:
        yield name, desc, b"".join(seq)


def write_record(f: BinaryIO, name: bytes, desc: Optional[bytes], seq: bytes) -> None:
    f.write(b">" + name + ((b" " + desc) if desc else b"") + b"\n")
    for i in range(0, len(seq), LINE_BASES):
        f.write(seq[i:i + LINE_BASES] + b"\n")


def read_fastq_sequences(f: BinaryIO) -> Iterator[bytes]:
    """Sequences of a FASTQ stream for the set builders (`solid -f fastq`, `large-kmer -f fastq`: src/set/pcon.rs:114-181,
    src/set/hash.rs `from_fastq`; noodles::fastq::Reader, optional `fastq` feature, unpinned by any test): records of
    four lines -- '@' + non-empty name, sequence, '+' line, qualities of the sequence's length.  As with FASTA the
    reference's `while let Some(Ok(record))` ends the stream silently at the first malformed record."""
    while True:
        head = f.readline()
        if not head:
            return
        seq, plus, qual = f.readline(), f.readline(), f.readline()
        head, seq, plus, qual = (x.rstrip(b"\r\n") for x in (head, seq, plus, qual))
        if not head.startswith(b"@") or len(head) < 2 or head[1:2].isspace():
            return
        if not plus.startswith(b"+") or len(qual) != len(seq):
            return
        yield seq


def read_csv_kmers(f: BinaryIO, k: int) -> Iterator[bytes]:
    """First column of every row after the header row (csv::Reader::from_reader skips it), as the set builders take it
    (`solid -f csv`, `large-kmer -f csv`: src/set/pcon.rs:27-45, src/set/hash.rs:21-38; optional `csv` feature).  The
    reference passes the field to seq2bit whatever its length; here a field that is not exactly k bases is an error."""
    import csv
    import io
    rows = csv.reader(io.TextIOWrapper(f, encoding="latin-1", newline=""))
    next(rows, None)
    for n, row in enumerate(rows, start=2):
        if not row:
            continue  # the csv crate skips empty lines
        field = row[0].encode("latin-1")
        if len(field) != k:
            raise ValueError(f"csv line {n}: first column {row[0]!r} is not a {k}-mer")
        yield field
