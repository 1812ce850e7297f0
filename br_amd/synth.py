"""Deterministic synthetic ONT-like reads.

Uniform random genome, fixed-length reference windows, random strand, i.i.d. per-reference-base
errors (2 % substitution, 1.5 % insertion, 1.5 % deletion by default).  The generator lives in
libbrx (br_amd/csrc/brx_synth.hip) and is bit-identical on host and device.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib

DEFAULT_SEED = 0xB12


def config(genome_len: int, read_len: int = 10_000, seed: int = DEFAULT_SEED, sub: float = 0.02, ins: float = 0.015,
           dele: float = 0.015) -> _lib.Synth:
    return _lib.Synth(seed, genome_len, read_len, int(round(sub * 1e4)), int(round(ins * 1e4)),
                      int(round(dele * 1e4)))


def genome_host(cfg: _lib.Synth) -> np.ndarray:
    g = np.zeros(cfg.genome_len, dtype=np.uint8)
    _lib.check(_lib.lib().brx_synth_genome_host(C.byref(cfg), g.ctypes.data))
    return g


def reads_host(cfg: _lib.Synth, genome: np.ndarray, first_read: int, n_reads: int) -> Tuple[np.ndarray, np.ndarray]:
    cap = int(n_reads) * (2 * cfg.read_len + 8)
    bases = np.zeros(cap, dtype=np.uint8)
    offs = np.zeros(n_reads + 1, dtype=np.uint64)
    tot = C.c_uint64(0)
    _lib.check(_lib.lib().brx_synth_reads_host(C.byref(cfg), genome.ctypes.data, first_read, n_reads,
                                               bases.ctypes.data, cap, offs.ctypes.data, C.byref(tot)))
    return bases[:tot.value].copy(), offs


def genome_device(cfg: _lib.Synth, device: int, d_genome: int, stream: Optional[int] = None) -> None:
    _lib.check(_lib.lib().brx_synth_genome_device(C.byref(cfg), device, d_genome, stream))


def reads_device(cfg: _lib.Synth, device: int, d_genome: int, first_read: int, n_reads: int, d_bases: int,
                 bases_cap: int, d_offsets: int, stream: Optional[int] = None) -> int:
    tot = C.c_uint64(0)
    _lib.check(_lib.lib().brx_synth_reads_device(C.byref(cfg), device, d_genome, first_read, n_reads, d_bases,
                                                 bases_cap, d_offsets, C.byref(tot), stream))
    return tot.value
