"""Live comparison of the oracle restatement with the COMPILED REFERENCE
(oracle/_ref/libspmv_ref.so) on matrices beyond the committed fixtures.
Skipped where the compiled reference is absent (it is built by
`make -C oracle all` only where /root/reference exists).
"""
import numpy as np
import pytest

import oracle
from spmv_amd import generators as G

pytestmark = pytest.mark.skipif(not oracle.ref_available(), reason="oracle/_ref not built")

SEEDS = [1, 2, 3, 4]


def _random_case(seed):
    rng = np.random.default_rng(seed)
    kind = seed % 4
    if kind == 0:
        rows, cols, rp, cj, va = G.stencil27(int(rng.integers(3, 9)), int(rng.integers(3, 9)), int(rng.integers(3, 9)))
    elif kind == 1:
        rows, cols, rp, cj, va = G.band(int(rng.integers(100, 900)), int(rng.integers(1, 9)), seed=seed)
    elif kind == 2:
        rows, cols, rp, cj, va = G.rmat(int(rng.integers(6, 10)), int(rng.integers(2, 12)), seed=seed)
    else:
        rows, cols, rp, cj, va = G.power_law_rows(int(rng.integers(200, 900)), seed=seed)
    coo = G.csr_to_coo_numpy(rows, rp, cj, va)
    x = oracle.aligned_empty(cols)
    x[:] = rng.random(cols)
    return rows, cols, np.ascontiguousarray(coo[rng.permutation(len(coo))]), x


@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_equals_compiled_reference(seed):
    R = oracle.ref()
    rows, cols, coo_in, x = _random_case(seed)
    nnz = len(coo_in)
    coo = coo_in.copy()
    R.ref_sort_mtx(nnz, coo)
    assert oracle.sort_coo(coo_in).tobytes() == coo.tobytes()

    c = R.ref_create_csr(rows, cols, nnz, 1, coo)
    rp, cj, va = oracle.csr_from_coo(rows, coo, 1)
    assert oracle.as_np(c.contents.i, rows + 1, np.int32).tobytes() == rp.tobytes()
    assert oracle.as_np(c.contents.j, nnz, np.int32).tobytes() == cj.tobytes()
    y = oracle.aligned_empty(rows + 64, fill=np.nan)
    R.ref_mult_csr_base(c, x, y)
    y_base = y[:rows].copy()
    assert oracle.mult_csr_base(rp, cj, va, np.ascontiguousarray(x)).tobytes() == y_base.tobytes()

    for nt in (2, 5):
        m = R.ref_create_csr_merge(nt, rows, cols, nnz, coo)
        y[:] = np.nan
        R.ref_mult_csr_merge(m, x, y)
        yo, rc, vc = oracle.mult_csr_merge(rp, cj, va, np.ascontiguousarray(x), nt)
        assert yo.tobytes() == y[:rows].tobytes()
        assert rc.tobytes() == oracle.as_np(m.contents.row_carry, nt, np.int32).tobytes()

    for sigma in (64, 256):
        s = R.ref_create_sell(c, 64, sigma)
        S = oracle.sell_from_csr(rp, cj, va, 64, sigma)
        ns = R.ref_sell_nslices(s)
        assert oracle.as_np(R.ref_sell_row_order(s), rows, np.int32).tobytes() == S["row_order"].tobytes()
        sp = oracle.as_np(R.ref_sell_slice_pointers(s), ns + 1, np.int32).astype(np.int64)
        assert sp.tobytes() == S["slice_ptr"].tobytes()
        assert oracle.as_np(R.ref_sell_column_indices(s), int(sp[-1]), np.int32).tobytes() == S["col"].tobytes()
        assert oracle.as_np(R.ref_sell_values(s), int(sp[-1]), np.float64).tobytes() == S["val"].tobytes()
        assert oracle.mult_sell(S, np.ascontiguousarray(x)).tobytes() == y_base.tobytes()
        # the same conversion with enable_dfc = 1 (sellcs_format.c:53-136): no padding, active lanes per column
        d = R.ref_create_sell_dfc(c, 64, sigma)
        D = oracle.sell_dfc_from_csr(rp, cj, va, 64, sigma)
        vops = int(D["vop_ptr"][-1])
        assert oracle.as_np(R.ref_sell_row_order(d), rows, np.int32).tobytes() == D["row_order"].tobytes()
        assert oracle.as_np(R.ref_sell_vop_pointers(d), ns + 1, np.uint64).astype(np.int64).tobytes() == D["vop_ptr"].tobytes()
        assert oracle.as_np(R.ref_sell_vop_lengths(d), vops, np.uint8).tobytes() == D["vop_len"].tobytes()
        assert oracle.as_np(R.ref_sell_slice_pointers(d), ns + 1, np.int32).astype(np.int64).tobytes() == D["slice_ptr"].tobytes()
        assert oracle.as_np(R.ref_sell_column_indices(d), nnz, np.int32).tobytes() == D["col"].tobytes()
        assert oracle.as_np(R.ref_sell_values(d), nnz, np.float64).tobytes() == D["val"].tobytes()
        assert oracle.mult_sell_dfc(D, np.ascontiguousarray(x)).tobytes() == y_base.tobytes()

    sp_ = R.ref_create_sellp(rows, cols, nnz, rp, cj, va)
    E = oracle.sliced_ell_from_csr(rp, cj, va, R.ref_block())
    nb = sp_.contents.num_blocks
    bp = oracle.as_np(sp_.contents.block_ptr, nb + 1, np.int32).astype(np.int64)
    assert bp.tobytes() == E["block_ptr"].tobytes()
    assert oracle.as_np(sp_.contents.j, int(bp[-1]), np.int32).tobytes() == E["j"].tobytes()
    y[:] = np.nan
    R.ref_mult_sellp_autovec(sp_, x, y)
    assert oracle.mult_sliced_ell(E, np.ascontiguousarray(x)).tobytes() == y[:rows].tobytes()


def test_driver_init_and_error_metric_match_reference_formulas():
    # driver_spmv.c:267-273 uses unseeded glibc rand(); the oracle calls the same libc
    x, y = oracle.driver_init_xy(1000)
    assert abs(np.linalg.norm(x) - 1.0) < 1e-12 and (y >= 0).all() and (y <= 1).all()
    a = np.arange(1.0, 11.0)
    b = a.copy(); b[3] += 1e-6
    want = np.sqrt(np.sum((a - b) ** 2)) / np.sqrt(np.sum(b ** 2))       # driver_spmv.c:278-289
    assert abs(oracle.rel_l2(a, b) - want) <= 1e-18
