"""Abundance-threshold selection from the count spectrum.

The reference picks the threshold with `pcon::spectrum::Spectrum::get_threshold` (src/main.rs:93-110); pcon
(git 0184ae7, Cargo.lock:1106-1124) is not vendored under the reference, so these are restatements of its
published `src/spectrum.rs`, and the reference's own tests pin none of them (tests/br.rs:9-33 runs
`first-minimum` and checks no output): PARITY UNPINNED.  The threshold then feeds `Solid::from_count`
(`count > threshold`).  The spectrum itself (`Counter.spectrum()`, a HIP kernel) is exact by construction and
is tested against a histogram of the oracle's counts.

The arithmetic is what Rust does: u64 sums, f64 ratios, IEEE comparisons (a NaN or inf ratio compares false).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

METHODS = ("first-minimum", "rarefaction", "percent-most", "percent-least")   # src/cli.rs:228-241


def _ratio(a: int, b: int) -> float:
    if b == 0:
        return math.nan if a == 0 else math.inf
    return float(a) / float(b)


def first_minimum(spectrum: Sequence[int]) -> Optional[int]:
    """ThresholdMethod::FirstMinimum: first i with spectrum[i+1] > spectrum[i] (the Pareto/Gaussian crossing)."""
    for i in range(len(spectrum) - 1):
        if int(spectrum[i + 1]) > int(spectrum[i]):
            return i
    return None


def rarefaction(spectrum: Sequence[int], limit: float) -> Optional[int]:
    """ThresholdMethod::Rarefaction: first abundance whose k-mers are less than `limit` of the k-mer
    occurrences seen up to and including it."""
    cumulative = 0
    for index, value in enumerate(spectrum):
        cumulative += index * int(value)
        if _ratio(int(value), cumulative) < limit:
            return index
    return None


def percent_at_least(spectrum: Sequence[int], percent: float) -> Optional[int]:
    """ThresholdMethod::PercentAtLeast: first abundance at which more than `percent` of all k-mer occurrences
    are at or below it (removes at least that share)."""
    total = sum(index * int(value) for index, value in enumerate(spectrum))
    cumulative = 0
    for index, value in enumerate(spectrum):
        cumulative += index * int(value)
        if _ratio(cumulative, total) > percent:
            return index
    return None


def percent_at_most(spectrum: Sequence[int], percent: float) -> Optional[int]:
    """ThresholdMethod::PercentAtMost: one below PercentAtLeast (u8 arithmetic: 0 - 1 wraps to 255 in release)."""
    t = percent_at_least(spectrum, percent)
    return None if t is None else (t - 1) & 0xFF


def get_threshold(spectrum: Sequence[int], method: str, percent: float = 0.0) -> Optional[int]:
    """Spectrum::get_threshold, dispatched on `br`'s sub-command names (src/main.rs:95-108)."""
    if method == "first-minimum":
        return first_minimum(spectrum)
    if method == "rarefaction":
        return rarefaction(spectrum, percent)
    if method == "percent-least":
        return percent_at_least(spectrum, percent)
    if method == "percent-most":
        return percent_at_most(spectrum, percent)
    raise ValueError(f"unknown abundance method {method!r}")
