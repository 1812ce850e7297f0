"""br_amd -- MI355X-native (gfx950, HIP) replacement for the hot path of natir/br:
solid-k-mer set build and the per-read correction scan, behind the reference's own
KmerSet / Corrector / build_methods / run_correction surface.

Everything that computes goes through libbrx.so (br_amd/csrc, C ABI in include/brx.h).
"""
from . import _lib  # noqa: F401
from .set import KmerSet, Pcon, Counter, pack_reads, seq2bit  # noqa: F401
from .correct import (Chain, Corrector, One, Two, Graph, Greedy, GapSize, build_methods)  # noqa: F401
from .driver import run_correction  # noqa: F401

__all__ = ["KmerSet", "Pcon", "Counter", "Chain", "Corrector", "One", "Two", "Graph", "Greedy", "GapSize",
           "build_methods", "run_correction", "pack_reads", "seq2bit"]
