"""numpy mirror of include/cs_synth.h — the counter-based synthetic data generator.

Bit-identical to the C / HIP versions (integer arithmetic, one exact int->float
conversion, exact power-of-two scaling).  Used by tests and bench.py to re-derive on the
host any slice of a corpus that libcsgpu generated in place on the GPU.
"""
from __future__ import annotations

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def synth_int(seed: int, idx: np.ndarray) -> np.ndarray:
    """cs_synth_int: int32 in [-131070, 131070] for each flat index."""
    with np.errstate(over="ignore"):
        h = _mix64(np.uint64(seed) + idx.astype(np.uint64) * GOLDEN)
    m = np.uint64(0xFFFF)
    v = (
        (h & m).astype(np.int64)
        + ((h >> np.uint64(16)) & m).astype(np.int64)
        + ((h >> np.uint64(32)) & m).astype(np.int64)
        + (h >> np.uint64(48)).astype(np.int64)
    )
    return (v - 131070).astype(np.int32)


def synth_rows(seed: int, first_row: int, n: int, dim: int) -> np.ndarray:
    """cs_synth_value for rows [first_row, first_row+n) -> float32 [n, dim]."""
    out = np.empty((n, dim), dtype=np.float32)
    step = max(1, (1 << 22) // dim)
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        idx = np.arange((first_row + lo) * dim, (first_row + hi) * dim, dtype=np.uint64)
        out[lo:hi] = (synth_int(seed, idx).astype(np.float32) * np.float32(1.0 / 65536.0)).reshape(hi - lo, dim)
    return out


def synth_weight(seed: int, first_idx: int, count: int, shift: int) -> np.ndarray:
    """cs_synth_weight for flat indices [first_idx, first_idx+count)."""
    idx = np.arange(first_idx, first_idx + count, dtype=np.uint64)
    v = synth_int(seed, idx).astype(np.float32) * np.float32(1.0 / 65536.0)
    return v * np.float32(1.0 / float(1 << shift))


def synth_below(seed: int, idx: np.ndarray, n: int) -> np.ndarray:
    """cs_synth_below: uniform integers in [0, n)."""
    with np.errstate(over="ignore"):
        h = _mix64(np.uint64(seed) + idx.astype(np.uint64) * GOLDEN)
    return (((h >> np.uint64(32)) * np.uint64(n)) >> np.uint64(32)).astype(np.uint32)


def synth_planted(seed_c: int, seed_q: int, rows, dim: int) -> np.ndarray:
    """cs_synth_planted: query i = corpus row rows[i] + 0.5 * noise(seed_q, i)."""
    rows = np.asarray(rows, dtype=np.uint64)
    out = np.empty((len(rows), dim), dtype=np.float32)
    for i, r in enumerate(rows):
        base = synth_rows(seed_c, int(r), 1, dim)[0]
        noise = synth_rows(seed_q, i, 1, dim)[0]
        out[i] = base + np.float32(0.5) * noise
    return out
