"""Small matrices shared by the golden-vector generator and the parity tests.

Each case is (name, rows, columns, coo) with coo a structured array in the
reference's `struct mtx` layout (reference src/spmv.h:26), NOT necessarily
sorted.  They cover the edge cases a reference test-suite would need (the
reference has none): row counts that are not multiples of the slice heights,
empty rows, an all-empty matrix, 1x1, one dense row, skewed row lengths,
rectangular shape and an unsorted input.
"""
from __future__ import annotations

import numpy as np

from spmv_amd import generators as G
from spmv_amd.matrix import COO_DTYPE, make_coo


def _from_gen(gen):
    rows, cols, rp, cj, va = gen
    return rows, cols, G.csr_to_coo_numpy(rows, rp, cj, va)


def _dense_row_rect():
    rows, cols = 3, 200
    i = np.concatenate([np.zeros(1, np.int32), np.full(cols, 1, np.int32), np.full(2, 2, np.int32)])
    j = np.concatenate([[5], np.arange(cols), [0, 199]]).astype(np.int32)
    a = np.linspace(-1.0, 1.0, len(i)) + 0.125
    return rows, cols, make_coo(i, j, a)


def _shuffled(case, seed=11):
    rows, cols, coo = case
    rng = np.random.default_rng(seed)
    return rows, cols, np.ascontiguousarray(coo[rng.permutation(len(coo))])


def _diag(n):
    i = np.arange(n, dtype=np.int32)
    return n, n, make_coo(i, i, 1.0 + i.astype(np.float64) / n)


def all_cases():
    cases = {
        "lap2d_37": _from_gen(G.laplacian2d(37)),
        "lap2d_64": _from_gen(G.laplacian2d(64)),
        "stencil27_9x10x11": _from_gen(G.stencil27(9, 10, 11)),
        "band_500_5": _from_gen(G.band(500, 5)),
        "arrow_300_4": _from_gen(G.arrow(300, 4)),
        "powerlaw_700": _from_gen(G.power_law_rows(700)),
        "rmat_s9": _from_gen(G.rmat(9, 8)),
        "dense_row_rect": _dense_row_rect(),
        "one_by_one": (1, 1, make_coo([0], [0], [2.5])),
        "all_empty_130": (130, 130, np.zeros(0, dtype=COO_DTYPE)),
        "diag_64": _diag(64),
    }
    cases["lap2d_37_shuffled"] = _shuffled(cases["lap2d_37"])
    return cases


def x_for(columns: int, seed=5) -> np.ndarray:
    """Deterministic start vector, L2-normalised like the driver's (driver_spmv.c:272-273)."""
    rng = np.random.default_rng(seed)
    x = rng.random(columns)
    return x / np.sqrt(np.dot(x, x)) if columns else x


def has_empty_rows(rows, coo) -> bool:
    return len(np.unique(coo["i"])) != rows


SELL_SIGMAS = (64, 1024, 16384)
MERGE_THREADS = (1, 3, 8)
BAL_THREADS = (1, 3, 8)
