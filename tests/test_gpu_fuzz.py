"""Randomised parity: matrices with adversarial row-length patterns (rows spanning many tiles,
runs of empty rows, single-entry rows, rectangular shapes, sizes off every block boundary) through
every format of the C ABI against the CPU oracle's mult_csr_base (reference csr.c:4-13).
Tolerances as in test_gpu_parity.py."""
import os

import numpy as np
import pytest

import spmv_amd as S
from spmv_amd.matrix import Matrix
from tests.util import bits_equal

pytestmark = pytest.mark.gpu

SEEDS = list(range(int(os.environ.get("SPMV_FUZZ_SEEDS", "12"))))   # more seeds for an occasional longer run


def _random_csr(seed):
    rng = np.random.default_rng(1000 + seed)
    rows = int(rng.choice([1, 63, 64, 65, 257, 1000, 4097, 20011]))
    cols = int(rng.choice([1, 7, 64, 513, rows, 30000]))
    shape = seed % 4
    if shape == 0:                                   # mostly short rows, a few very long ones
        lens = rng.integers(0, 6, rows)
        lens[rng.integers(0, rows, max(rows // 200, 1))] = rng.integers(600, 5000, max(rows // 200, 1))
    elif shape == 1:                                 # long runs of empty rows between dense bursts
        lens = np.where((np.arange(rows) // 97) % 3 == 0, rng.integers(20, 90, rows), 0)
    elif shape == 2:                                 # exactly one entry per row
        lens = np.ones(rows, np.int64)
    else:                                            # geometric lengths
        lens = rng.geometric(0.05, rows) - 1
    lens = np.minimum(lens, cols).astype(np.int64)
    rp = np.zeros(rows + 1, np.int32)
    rp[1:] = np.cumsum(lens)
    cj = np.empty(int(rp[-1]), np.int32)
    for r in np.nonzero(lens)[0]:
        cj[rp[r]:rp[r + 1]] = np.sort(rng.choice(cols, size=lens[r], replace=False))
    va = rng.standard_normal(len(cj))
    x = rng.standard_normal(cols)
    return rows, cols, rp, cj, va, x


@pytest.mark.parametrize("seed", SEEDS)
def test_every_format_on_random_shapes(orc, seed):
    rows, cols, rp, cj, va, x = _random_csr(seed)
    nnz = len(cj)
    want = orc.mult_csr_base(rp, cj, va, x)
    x32 = x.astype(np.float32)
    want32 = orc.mult_csr_base_f32(rp, cj, va.astype(np.float32), x32)
    va32 = va.astype(np.float32).astype(np.float64)
    exact32 = orc.mult_csr_base(rp, cj, va32, x32.astype(np.float64))            # fp64 product of the fp32 data
    mag32 = orc.mult_csr_base(rp, cj, np.abs(va32), np.abs(x32).astype(np.float64))
    bound32 = (np.diff(rp) + 1) * 2.0 ** -24 * mag32 * 1.01 + 1e-45
    coo = S.make_coo(np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp)), cj, va)
    builders = {
        "csr_scalar": lambda dt: Matrix.csr(rows, cols, coo.copy(), 1, dt, S.CSR_SCALAR),
        "csr_vector": lambda dt: Matrix.csr(rows, cols, coo.copy(), 4, dt, S.CSR_VECTOR),
        "csr_merge_256": lambda dt: Matrix.csr_merge(rows, cols, coo.copy(), dt, 256),
        "csr_merge_2048": lambda dt: Matrix.csr_merge(rows, cols, coo.copy(), dt, 2048),
        "csr_bal_256": lambda dt: Matrix.csr_bal(rows, cols, coo.copy(), dt, 256),
        "csr_bal": lambda dt: Matrix.csr_bal(rows, cols, coo.copy(), dt),
        "sell_64": lambda dt: Matrix.sell(rows, cols, rp, cj, va, dt, 64, 64),
        "sell_1024": lambda dt: Matrix.sell(rows, cols, rp, cj, va, dt, 64, 1024),
        "sell_dfc": lambda dt: Matrix.sell_dfc(rows, cols, rp, cj, va, dt, 64, 1024),
        "sellp": lambda dt: Matrix.sellp(rows, cols, rp, cj, va, dt),
        "sellp_8": lambda dt: Matrix.sellp(rows, cols, rp, cj, va, dt, 8),
        "ell": lambda dt: Matrix.ell(rows, cols, coo.copy(), dt),
        "auto": lambda dt: Matrix.auto(rows, cols, coo.copy(), dt, 1),
    }
    exact = {"csr_scalar", "sell_64", "sell_1024", "sell_dfc", "sellp", "sellp_8", "ell"}
    for name, build in builders.items():
        m = build(S.F64)
        assert m.info.nnz == nnz
        y = m.mult(x)
        if name in exact:
            assert bits_equal(y, want), (name, seed)
        else:
            assert orc.rel_l2(y, want) <= 1e-13, (name, seed)
        m.analyze(1)                                 # whatever variant wins must still be right
        assert orc.rel_l2(m.mult(x), want) <= 1e-13, (name, seed, m.info.variant)
        if name.startswith(("csr_merge", "csr_bal")) and nnz:     # patterned-tile kernel
            m.set_option("variant", 2)
            assert orc.rel_l2(m.mult(x), want) <= 1e-13, (name, seed, 2)
        if name == "csr_scalar":                                   # patterned-block kernels
            for v in (16, 22):
                m.set_option("variant", v)
                assert bits_equal(m.mult(x), want), (name, seed, v)
        if name == "sell_dfc" and nnz:
            for v in (9, 12):
                m.set_option("variant", v)
                assert bits_equal(m.mult(x), want), (name, seed, v)
        if name in ("sell_64", "sell_1024", "sellp", "ell"):      # diagonal-column kernels (clamped padding lanes)
            for v in (16, 22):
                m.set_option("variant", v)
                assert bits_equal(m.mult(x), want), (name, seed, v)
        m.free()
        m = build(S.F32)
        y = m.mult(x32.astype(np.float64))
        if name in exact:
            assert bits_equal(y.astype(np.float32), want32), (name, seed)
        else:
            # Any summation order of n fp32 products is within (n+1)·u·Σ|a·x| of the exact sum
            # (u = 2^-24); the in-order fp32 result of the oracle carries the same error, so it
            # is not the yardstick for rows of thousands of entries.
            assert np.all(np.abs(y - exact32) <= bound32), (name, seed)
            assert orc.rel_l2(y, want32.astype(np.float64)) <= 1e-5, (name, seed)
        m.free()
