"""The checker for products with a block of vectors: orc_mult_mv_csr must be n products of
mult_csr_base (reference csr.c:4-13) on vectors spaced ldx / ldy apart, and must agree with
scipy's CSR product (an independent implementation) to rounding."""
import ctypes as C

import numpy as np
import scipy.sparse as sp

from tests.test_gpu_fuzz import _random_csr


def test_mult_mv_csr_is_n_single_products(orc):
    for seed in (0, 1, 3, 32):
        rows, cols, rp, cj, va, _ = _random_csr(seed)
        X = np.random.default_rng(seed).standard_normal((5, cols))
        Y = orc.mult_mv_csr(rp, cj, va, X)
        for v in range(5):
            assert Y[v].tobytes() == orc.mult_csr_base(rp, cj, va, np.ascontiguousarray(X[v])).tobytes()
        A = sp.csr_matrix((va, cj, rp), shape=(rows, cols))
        ref = (A @ X.T).T
        mag = (abs(A) @ np.abs(X).T).T
        assert np.all(np.abs(Y - ref) <= 1e-12 * mag + 1e-300)


def test_mult_mv_csr_strides_leave_gaps_alone(orc):
    rows, cols, rp, cj, va, _ = _random_csr(3)
    n, ldx, ldy = 3, cols + 5, rows + 9
    X = np.random.default_rng(1).standard_normal((n, ldx))
    Y = np.full((n, ldy), -3.0)
    orc.lib().orc_mult_mv_csr(rows, rp, cj, va, n, C.c_void_p(X.ctypes.data), ldx, C.c_void_p(Y.ctypes.data), ldy)
    want = orc.mult_mv_csr(rp, cj, va, np.ascontiguousarray(X[:, :cols]))
    assert np.ascontiguousarray(Y[:, :rows]).tobytes() == want.tobytes()
    assert np.all(Y[:, rows:] == -3.0)
