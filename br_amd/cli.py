"""Command-line surface of `br` (src/cli.rs:19-76, 172-185; dispatch src/main.rs:17-58), over the HIP path.

    python -m br_amd -i reads.fa -o corr.fa [-s] [-c one -c graph ...] [-C 5] [-M 7] \
        fasta -i reads.fa -k 19 -a 3                       # count -> threshold -> correct (src/main.rs:72-85)
        solid -i set.solid -f solid                        # load a pcon .solid set     (src/main.rs:117-120)
        count -i table.pcon -a 3                           # load a pcon count table    (src/main.rs:59-70)
        solid -i reads.fa -f fasta -k 19                   # presence-only set          (src/set/pcon.rs:47-112)
        solid -i reads.fq -f fastq -k 19 | solid -i kmers.csv -f csv -k 19   # the reference's optional features

Same flags, defaults and quirks as the reference: `-s/--two-side` DISABLES the reverse pass
(src/lib.rs:48,110); `fasta -k` is forced odd (src/cli.rs:277-279); without `-a` an abundance method
sub-command is needed (src/main.rs:95-110): `first-minimum`, `rarefaction P`, `percent-most P`, `percent-least P`
pick the threshold from the count spectrum (br_amd/spectrum.py: pcon's published formulas, unpinned by the
reference's tests).  `count -i table` loads a pcon count table ([k][2^(2k-1) u8 counters], layout unpinned: the
reference holds no such fixture) and thresholds it the same way; `large-kmer -f fasta` (N4) builds a sparse set for odd k <= 31.  `-t` (rayon
pool size) is accepted and ignored: the GPU is the pool.
"""
from __future__ import annotations

import argparse
import bz2
import gzip
import io
import lzma
import sys
from typing import BinaryIO, List, Optional

from . import _lib, fasta, spectrum
from .correct import build_methods
from .driver import run_correction
from .set import Counter, Pcon

METHOD_NAMES = ["one", "two", "graph", "greedy", "gap-size"]  # clap ValueEnum kebab-case of CorrectionMethod



def u8(text: str) -> int:
    """clap parses -a / -C / -M into u8 fields (src/cli.rs:46,50,196): out-of-range values are rejected, not wrapped"""
    try:
        v = int(text)
    except ValueError:
        raise argparse.ArgumentTypeError("invalid digit found in string: %r" % text)
    if not 0 <= v <= 255:
        raise argparse.ArgumentTypeError("%r is not in 0..=255" % text)
    return v


def open_input(path: str) -> BinaryIO:
    """niffler::get_reader: sniff gz / bz2 / xz by magic bytes (src/cli.rs:209,269,404,415)."""
    raw = open(path, "rb")
    head = raw.peek(6)[:6] if hasattr(raw, "peek") else b""
    # (peek does not move the logical position: a plain file can still be handed to the native pipeline as a descriptor)
    if head[:2] == b"\x1f\x8b":
        return gzip.open(raw, "rb")
    if head[:3] == b"BZh":
        return bz2.open(raw, "rb")
    if head[:6] == b"\xfd7zXZ\x00":
        return lzma.open(raw, "rb")
    return raw


def parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="br", description="Br: Brutal rewrite a simple long read corrector based on kmer "
                                                       "spectrum methode (MI355X-native hot path)")
    # clap derive, Option<Vec<T>>: ONE value per occurrence, repeat the flag for more (-i a -i b)
    p.add_argument("-i", "--inputs", action="append", default=None, help="Path to inputs, default read stdin")
    p.add_argument("-o", "--outputs", action="append", default=None, help="Path to output, default stdout")
    p.add_argument("-s", "--two-side", action="store_true", help="Correct in two side (sic: disables the reverse pass)")
    p.add_argument("-c", "--corrections", action="append", choices=METHOD_NAMES, default=None,
                   help="Correction method")
    p.add_argument("-C", "--confirm", type=u8, default=None, help="Number of kmer required to validate correction")
    p.add_argument("-M", "--max-search", type=u8, default=None, help="Number of base we use to try correct error")
    p.add_argument("-b", "--record_buffer", type=int, default=None, help="Number of sequence record load in buffer")
    p.add_argument("-t", "--threads", type=int, default=None, help="accepted for compatibility, ignored")
    p.add_argument("-q", "--quiet", action="store_true")
    p.add_argument("-v", "--verbosity", action="count", default=0)
    p.add_argument("-T", "--timestamp", default=None)
    p.add_argument("--device", type=int, default=0, help="GPU index (not in the reference)")
    sub = p.add_subparsers(dest="subcommand", required=True)

    def abundance_methods(sp):
        ssub = sp.add_subparsers(dest="abundance_selection")
        ssub.add_parser("first-minimum")
        for name in ("rarefaction", "percent-most", "percent-least"):
            ssub.add_parser(name).add_argument("percent", type=float)

    c = sub.add_parser("count", help="With Count")
    c.add_argument("-i", "--inputs", dest="sub_inputs", required=True)
    c.add_argument("-a", "--abundance", type=u8, default=None)
    abundance_methods(c)
    f = sub.add_parser("fasta", help="With Fasta")
    f.add_argument("-i", "--inputs", dest="sub_inputs", action="append", required=True)
    f.add_argument("-k", "--kmer-size", type=int, required=True)
    f.add_argument("-a", "--abundance", type=u8, default=None)
    abundance_methods(f)
    s = sub.add_parser("solid", help="With Solid")
    s.add_argument("-i", "--input", dest="sub_input", required=True)
    s.add_argument("-f", "--format", choices=["solid", "fasta", "fastq", "csv"], required=True)
    s.add_argument("-k", "--kmer-size", type=int, default=None)
    lk = sub.add_parser("large-kmer", help="Large Kmer mode")
    lk.add_argument("-i", "--input", dest="sub_input", required=True)
    lk.add_argument("-f", "--format", choices=["fasta", "fastq", "csv"], required=True)
    lk.add_argument("-k", "--kmer-size", type=int, required=True)
    return p


def fasta_kmer_size(k: int) -> int:
    """Fasta::kmer_size, src/cli.rs:277-279: even k becomes k-1 (asserted 14 -> 13 at src/cli.rs:459)."""
    return k - ((~(k & 1)) & 1)


first_minimum = spectrum.first_minimum


def _records(paths: List[str]):
    for path in paths:
        with open_input(path) as f:
            for _, _, seq in fasta.read_records(f):
                yield seq


def threshold_and_finish(cnt: Counter, args) -> Pcon:
    """count2solid, src/main.rs:86-115: `-a N` wins; otherwise the counts are histogrammed on the GPU (no u8 table
    for the partitioned counter: its keys are binned bucket by bucket), the threshold is picked on the host and the
    same counter is finished with it"""
    if args.abundance is not None:
        return cnt.finish(args.abundance)
    if args.abundance_selection is None:
        raise SystemExit("Error: You must provide an abundance method or an abundance threshold")      # main.rs:109
    thr = spectrum.get_threshold(cnt.spectrum(), args.abundance_selection, getattr(args, "percent", 0.0))
    if thr is None:
        raise SystemExit("Error: Can't compute minimal abundance")                  # error.rs ComputeAbundanceThreshold
    return cnt.finish(thr)


def build_set(args) -> Pcon:
    dev = args.device
    if args.subcommand == "fasta":
        k = fasta_kmer_size(args.kmer_size)
        if args.abundance is None and args.abundance_selection is None:
            raise SystemExit("Error: You must provide an abundance method or an abundance threshold")  # main.rs:109
        cnt = Counter(k, dev)
        for path in args.sub_inputs:
            with open_input(path) as f:
                cnt.count_fasta(f)
        return threshold_and_finish(cnt, args)
    if args.subcommand == "solid":
        if args.format == "solid":
            with open_input(args.sub_input) as f:
                return Pcon.from_pcon_solid(f.read(), dev)
        if args.kmer_size is None:
            raise SystemExit("Error: Solid input fasta require kmer size")            # error.rs SolidRequireKmerSize
        return presence_set(args.sub_input, args.format, args.kmer_size, dev)
    if args.subcommand == "count":
        # src/main.rs:59-70: Counter::from_stream, then the same threshold -> Solid::from_count as `fasta`
        with open_input(args.sub_inputs) as f:
            cnt = Counter.from_count_stream(f, dev)
        return threshold_and_finish(cnt, args)
    # large-kmer -f fasta: set::Hash::from_fasta (src/set/hash.rs:40-60, src/main.rs:147-163) = every canonical k-mer of
    # every record, no counting.  Same membership as a presence-only Pcon; for k >= 21 the set is sparse (a chained
    # hash table in HBM instead of the bit vector).  Odd k only: cocktail's parity-canonical form is not a function
    # of the {k-mer, revcomp} pair for even k.
    if args.kmer_size % 2 == 0 or not 1 <= args.kmer_size <= 31:
        raise SystemExit("Error: large-kmer mode needs an odd k <= 31 on the HIP path (k=%d)" % args.kmer_size)
    return presence_set(args.sub_input, args.format, args.kmer_size, dev)


def presence_set(path: str, fmt: str, k: int, dev: int) -> Pcon:
    """Pcon::from_fasta / from_fastq / from_csv and their set::Hash twins: every canonical k-mer of every record
    (FASTA through the native pipeline; FASTQ and CSV -- optional features of the reference -- parsed on the host)"""
    with open_input(path) as f:
        if fmt == "fasta":
            return Pcon.from_fasta_file(f, k, dev)
        if fmt == "fastq":
            return Pcon.from_fasta(fasta.read_fastq_sequences(f), k, dev)
        try:
            return Pcon.from_fasta(fasta.read_csv_kmers(f, k), k, dev)
        except ValueError as e:
            raise SystemExit(f"Error: {e}")


def main(argv: Optional[List[str]] = None) -> int:
    args = parser().parse_args(argv)
    kmer_set = build_set(args)
    names = args.corrections or METHOD_NAMES                          # src/cli.rs:121-131: all five by default
    confirm = 5 if args.confirm is None else args.confirm            # src/cli.rs:135-137
    max_search = 7 if args.max_search is None else args.max_search   # src/cli.rs:140-142
    methods = build_methods(names, kmer_set, confirm, max_search)
    inputs = [open_input(p) for p in args.inputs] if args.inputs else [sys.stdin.buffer]
    outputs = [open(p, "wb") for p in args.outputs] if args.outputs else [io.BufferedWriter(sys.stdout.buffer)]
    try:
        run_correction(inputs, outputs, methods, args.two_side, args.record_buffer or 8192)
    finally:
        for f in outputs:
            f.flush()
        for f in inputs + outputs:
            if f not in (sys.stdin.buffer,):
                try:
                    f.close()
                except Exception:
                    pass
    return 0


if __name__ == "__main__":
    sys.exit(main())
