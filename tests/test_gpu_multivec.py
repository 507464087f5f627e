"""Products with a block of vectors through the C ABI (spmv_hip_mult_mv / _mult_mv_dev), the
operation the reference declares as mult_mv_csr (spmv.h:54) and never defines -- parity is therefore
unpinned by the reference and anchored on its single-vector kernel: column v of the result must be
bit-identical to the oracle's mult_csr_base (csr.c:4-13) of vector v, for a handle of any format,
in fp64 and (against the fp32 restatement) in fp32."""
import numpy as np
import pytest
import torch

import spmv_amd as S
from spmv_amd import generators as G
from spmv_amd.matrix import Matrix
from tests.test_gpu_fuzz import _random_csr
from tests.util import bits_equal

pytestmark = pytest.mark.gpu


def _builders(rows, cols, rp, cj, va):
    coo = S.make_coo(np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp)), cj, va)
    return {
        "csr_scalar": lambda dt: Matrix.csr(rows, cols, coo.copy(), 1, dt, S.CSR_SCALAR),
        "csr_vector_pad4": lambda dt: Matrix.csr(rows, cols, coo.copy(), 4, dt, S.CSR_VECTOR),
        "csr_merge": lambda dt: Matrix.csr_merge(rows, cols, coo.copy(), dt),
        "csr_bal": lambda dt: Matrix.csr_bal(rows, cols, coo.copy(), dt),
        "sell_1024": lambda dt: Matrix.sell(rows, cols, rp, cj, va, dt, 64, 1024),
        "sell_dfc": lambda dt: Matrix.sell_dfc(rows, cols, rp, cj, va, dt, 64, 1024),
        "sellp": lambda dt: Matrix.sellp(rows, cols, rp, cj, va, dt),
        "sellp_8": lambda dt: Matrix.sellp(rows, cols, rp, cj, va, dt, 8),
        "ell": lambda dt: Matrix.ell(rows, cols, coo.copy(), dt),
    }


def _cases():
    out = {}
    for seed in (0, 1, 2, 3, 4, 9, 32):
        rows, cols, rp, cj, va, _ = _random_csr(seed)
        out[f"fuzz{seed}"] = (rows, cols, rp, cj, va)
    for name, gen in (("stencil27_9", lambda: G.stencil27(9, 9, 9)), ("powerlaw", lambda: G.power_law_rows(3000, seed=4)),
                      ("rmat12", lambda: G.rmat(12, 8, seed=3))):
        rows, cols, rp, cj, va = gen()
        out[name] = (rows, cols, rp.numpy().astype(np.int32), cj.numpy().astype(np.int32), va.numpy().astype(np.float64))
    return out


CASES = _cases()


@pytest.mark.parametrize("case", sorted(CASES))
def test_mult_mv_every_format_bitwise(orc, case):
    rows, cols, rp, cj, va = CASES[case]
    rng = np.random.default_rng(5)
    for name, build in _builders(rows, cols, rp, cj, va).items():
        for n in (1, 2, 3, 7, 13):
            X = rng.standard_normal((n, cols))
            m = build(S.F64)
            Y = m.mult_mv(X)
            assert Y.shape == (n, rows)
            assert bits_equal(Y, orc.mult_mv_csr(rp, cj, va, X)), (case, name, n)
            m.free()
        m = build(S.F32)
        X32 = rng.standard_normal((5, cols)).astype(np.float32)
        Y = m.mult_mv(X32.astype(np.float64)).astype(np.float32)
        for v in range(5):
            assert bits_equal(Y[v], orc.mult_csr_base_f32(rp, cj, va.astype(np.float32), X32[v])), (case, name, v)
        m.free()


def test_mult_mv_dev_strided_blocks_and_untouched_gaps(orc):
    rows, cols, rp, cj, va = CASES["powerlaw"]
    m = _builders(rows, cols, rp, cj, va)["sell_1024"](S.F64)
    n, ldx, ldy = 6, cols + 37, rows + 11
    rng = np.random.default_rng(8)
    xb = torch.from_numpy(rng.standard_normal((n, ldx))).cuda()
    yb = torch.full((n, ldy), -7.0, dtype=torch.float64, device="cuda")
    m.mult_mv_dev(n, xb.data_ptr(), ldx, yb.data_ptr(), ldy, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = yb.cpu().numpy()
    want = orc.mult_mv_csr(rp, cj, va, np.ascontiguousarray(xb.cpu().numpy()[:, :cols]))
    assert bits_equal(np.ascontiguousarray(got[:, :rows]), want)
    assert np.all(got[:, rows:] == -7.0)                      # the slack between result vectors is not written
    m.free()


def test_mult_mv_matches_repeated_mult_after_analyze(orc):
    """The block product does not depend on the kernel variant picked for single vectors."""
    rows, cols, rp, cj, va = CASES["stencil27_9"]
    m = _builders(rows, cols, rp, cj, va)["sellp"](S.F64)
    m.analyze(1)
    X = np.random.default_rng(2).standard_normal((4, cols))
    assert bits_equal(m.mult_mv(X), orc.mult_mv_csr(rp, cj, va, X))
    m.free()


def test_mult_mv_argument_errors():
    rows, cols, rp, cj, va = CASES["fuzz1"]
    m = _builders(rows, cols, rp, cj, va)["csr_scalar"](S.F64)
    x = torch.zeros(2 * cols, dtype=torch.float64, device="cuda")
    y = torch.zeros(2 * rows, dtype=torch.float64, device="cuda")
    with pytest.raises(S.SpmvHipError, match="shorter than the vectors"):
        m.mult_mv_dev(2, x.data_ptr(), cols - 1 if cols > 1 else 0, y.data_ptr(), rows, 0)
    with pytest.raises(S.SpmvHipError, match="negative vector count"):
        m.mult_mv_dev(-1, x.data_ptr(), cols, y.data_ptr(), rows, 0)
    m.mult_mv_dev(0, 0, cols, 0, rows, 0)                     # no vectors: nothing to do
    with pytest.raises(ValueError):
        m.mult_mv(np.zeros((2, cols + 1)))
    m.free()


def test_mult_mv_empty_matrix(orc):
    rp = np.zeros(6, np.int32)
    cj = np.zeros(0, np.int32)
    va = np.zeros(0)
    for name, build in _builders(5, 3, rp, cj, va).items():
        m = build(S.F64)
        Y = m.mult_mv(np.ones((3, 3)))
        assert Y.shape == (3, 5) and np.all(Y == 0.0), name
        m.free()
