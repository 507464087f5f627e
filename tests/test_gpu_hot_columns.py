"""Option "hot_columns": columns relabeled by reference count, x permuted per product.  With value 1
the entries keep their places in the rows, so y must be bit-identical to the product of the same
handle without the option (and thereby keeps its parity with the oracle's mult_csr_base,
csr.c:4-13); with value 2 (columns dealt over the XCDs) every row is the sum of eight partial
sums and parity is by the fp64 / fp32 tolerances of the reduction formats."""
import numpy as np
import pytest

import spmv_amd as S
from spmv_amd import generators as G
from spmv_amd.matrix import Matrix
from tests.test_gpu_fuzz import _random_csr
from tests.util import bits_equal

pytestmark = pytest.mark.gpu


def _np(t, dt):
    return t.numpy().astype(dt)


def _matrices():
    out = {}
    for name, gen in (("rmat13", lambda: G.rmat(13, 8, seed=3)), ("powerlaw", lambda: G.power_law_rows(3000, seed=4))):
        rows, cols, rp, cj, va = gen()
        out[name] = (rows, cols, _np(rp, np.int32), _np(cj, np.int32), _np(va, np.float64))
    for seed in (0, 1, 3):
        rows, cols, rp, cj, va, _ = _random_csr(seed)
        if len(cj):
            out[f"fuzz{seed}"] = (rows, cols, rp, cj, va)
    return out


MATRICES = _matrices()


def _builders(rows, cols, rp, cj, va):
    coo = S.make_coo(np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp)), cj, va)
    return {
        "csr_scalar": lambda dt: Matrix.csr(rows, cols, coo.copy(), 1, dt, S.CSR_SCALAR),
        "csr_vector_pad4": lambda dt: Matrix.csr(rows, cols, coo.copy(), 4, dt, S.CSR_VECTOR),
        "csr_merge": lambda dt: Matrix.csr_merge(rows, cols, coo.copy(), dt),
        "csr_bal": lambda dt: Matrix.csr_bal(rows, cols, coo.copy(), dt),
    }


@pytest.mark.parametrize("name", sorted(MATRICES))
def test_hot_columns_product_is_bit_identical(orc, name):
    rows, cols, rp, cj, va = MATRICES[name]
    x = np.random.default_rng(3).standard_normal(cols)
    want = orc.mult_csr_base(rp, cj, va, x)
    for fmt, build in _builders(rows, cols, rp, cj, va).items():
        for dt in (S.F64, S.F32):
            m = build(dt)
            plain = m.mult(x)
            m.set_option("hot_columns", 0)                 # off already; also refreshes the cached info
            bytes_plain = m.info.device_bytes
            assert m.info.hot_columns == 0
            m.set_option("hot_columns", 1)
            assert m.info.hot_columns == 1 and m.info.device_bytes > bytes_plain
            hot = m.mult(x)
            assert bits_equal(hot, plain), (name, fmt, dt)
            if dt == S.F64:
                assert orc.rel_l2(hot, want) <= 1e-13
            m.analyze(1)                                   # variants are timed with the permutation in place
            assert orc.rel_l2(m.mult(x), plain) <= (1e-13 if dt == S.F64 else 2e-6)
            m.set_option("hot_columns", 1)                 # already on: nothing to do
            # 2: the columns dealt over the XCDs -- eight partial sums per row, so parity by tolerance
            m.set_option("hot_columns", 2)
            assert m.info.hot_columns == 2
            split = m.mult(x)
            assert orc.rel_l2(split, plain) <= (1e-13 if dt == S.F64 else 2e-6), (name, fmt, dt)
            if dt == S.F64:
                mag = orc.mult_csr_base(rp, cj, np.abs(va), np.abs(x))
                assert np.all(np.abs(split - want) <= 1e-12 * mag + 1e-300)
            # the tile kernels of the stacked matrix: 3 and 4 keep each column group's hottest labels in LDS
            # and otherwise are kernel 1, bit for bit
            m.set_option("hot_variant", 1)
            tiles = m.mult(x)
            assert orc.rel_l2(tiles, plain) <= (1e-13 if dt == S.F64 else 2e-6)
            for hv in (3, 4):
                m.set_option("hot_variant", hv)
                assert bits_equal(m.mult(x), tiles), (name, fmt, dt, hv)
            m.analyze(1)
            assert orc.rel_l2(m.mult(x), plain) <= (1e-13 if dt == S.F64 else 2e-6)
            m.set_option("hot_columns", 0)
            assert m.info.hot_columns == 0 and m.info.device_bytes == bytes_plain
            with pytest.raises(S.SpmvHipError, match="hot_columns 2"):
                m.set_option("hot_variant", 3)
            m.free()


def test_hot_columns_downloads_still_show_the_callers_columns():
    rows, cols, rp, cj, va = MATRICES["rmat13"]
    m = _builders(rows, cols, rp, cj, va)["csr_merge"](S.F64)
    m.set_option("hot_columns", 1)
    assert np.array_equal(m.download(S.ARR_COL), cj)
    m.free()


def test_hot_columns_refused_where_it_does_not_apply():
    rows, cols, rp, cj, va = MATRICES["powerlaw"]
    m = Matrix.sell(rows, cols, rp, cj, va, S.F64, 64, 1024)
    with pytest.raises(S.SpmvHipError, match="CSR-based"):
        m.set_option("hot_columns", 1)
    m.free()
    coo = S.make_coo(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0))
    m = Matrix.csr_merge(4, 4, coo, S.F64)
    with pytest.raises(S.SpmvHipError, match="empty matrix"):
        m.set_option("hot_columns", 1)
    assert np.all(m.mult(np.ones(4)) == 0.0)
    m.free()


def test_auto_may_relabel_the_columns_of_a_graph_matrix(orc):
    """create_auto tries the option when x outgrows one L2; whatever it decides, y stays right."""
    rows, cols, rp, cj, va = G.rmat(20, 16, seed=11)
    rp, cj, va = _np(rp, np.int32), _np(cj, np.int32), _np(va, np.float64)
    coo = S.make_coo(np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp)), cj, va)
    x = np.random.default_rng(6).standard_normal(cols)
    m = Matrix.auto(rows, cols, coo, S.F64, 2)
    assert m.info.hot_columns in (0, 1)
    assert m.info.hot_columns == 0 or m.info.format in (S.CSR_SCALAR, S.CSR_VECTOR, S.CSR_MERGE, S.CSR_BAL)
    assert orc.rel_l2(m.mult(x), orc.mult_csr_base(rp, cj, va, x)) <= 1e-13
    print("auto on rmat20:", S.FORMAT_NAMES[m.info.format], "variant", m.info.variant, "hot_columns", m.info.hot_columns)
    m.free()


@pytest.mark.parametrize("dt", [S.F64, S.F32])
def test_hot_labels_in_lds_on_a_graph_matrix(orc, dt):
    """R-MAT scale 18 (x of 2 MiB, 32768 labels per column group, more than the LDS holds): the persistent
    tile kernels read part of x from LDS and the rest from memory, and give the bits of the plain tile
    kernel; against the oracle by the tolerance of the eight partial sums."""
    rows, cols, rp, cj, va = G.rmat(18, 16, seed=5)
    rp, cj, va = _np(rp, np.int32), _np(cj, np.int32), _np(va, np.float64)
    coo = S.make_coo(np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp)), cj, va)
    x = np.random.default_rng(8).standard_normal(cols)
    want = orc.mult_csr_base(rp, cj, va, x)
    m = Matrix.csr_bal(rows, cols, coo, dt)
    m.set_option("hot_columns", 2)
    m.set_option("hot_variant", 1)
    tiles = m.mult(x)
    assert orc.rel_l2(tiles, want) <= (1e-13 if dt == S.F64 else 2e-6)
    with pytest.raises(S.SpmvHipError, match="not a tile kernel"):
        m.set_option("hot_variant", 7)
    for hv in (3, 4):
        m.set_option("hot_variant", hv)
        for _ in range(2):                                  # the second product finds the LDS of the first one's workgroups gone
            assert bits_equal(m.mult(x), tiles), hv
    m.free()
