"""GPU tests of behaviour at the edges of the plugin contract: non-finite x in the padded formats,
overlapping vector norms, and the "exact_order" tuning restriction.

The reference stores a padding entry of SELL-C-sigma / SELLP / ELL as (column 0, value 0)
(reference src/sellcs_format.c:167,171, src/sellp.c:440-452, src/ell.c:99-108) and multiplies it
like any other entry, so a non-finite x[0] turns every padded row into NaN.  Every kernel variant
of these formats must put the NaNs where the oracle's restatement of sellcs_mv_d_autovector /
mult_sellp puts them.
"""
import numpy as np
import pytest
import torch

import spmv_amd as S
from spmv_amd import generators as G
from spmv_amd.matrix import Matrix, make_coo, coo_to_csr_host

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

EXACT_SLICED = [0, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31]
REASSOC_SLICED = [1, 2, 3, 4, 9, 10, 11, 12]


def _same_with_nans(y, want):
    return np.array_equal(np.isnan(y), np.isnan(want)) and np.array_equal(y[~np.isnan(want)], want[~np.isnan(want)])


def _cases():
    out = {}
    for name, gen in (("stencil27_9x10x11", G.stencil27(9, 10, 11)), ("lap2d_37", G.laplacian2d(37)), ("band_500_5", G.band(500, 5)),
                      ("powerlaw_700", G.power_law_rows(700))):
        rows, cols, rp, cj, va = gen
        out[name] = (rows, cols, rp.numpy(), cj.numpy(), va.numpy().copy())
    # a band with stored zeros among its real entries: x[col] * 0 must stay the operand, not x[0] * 0
    rows, cols, rp, cj, va = out["band_500_5"]
    va = va.copy()
    va[::7] = 0.0
    out["band_500_5_stored_zeros"] = (rows, cols, rp, cj, va)
    return out


CASES = _cases()


def _nonfinite_x(cols, where):
    rng = np.random.default_rng(17)
    x = rng.random(cols)
    for k, v in where:
        x[k % cols] = v
    return x


@pytest.mark.parametrize("name", sorted(CASES))
@pytest.mark.parametrize("poison", ["nan_at_0", "inf_at_0", "inf_inside", "nan_and_inf"])
def test_nonfinite_x_lands_where_the_reference_puts_it(orc, name, poison):
    rows, cols, rp, cj, va = CASES[name]
    where = {"nan_at_0": [(0, np.nan)], "inf_at_0": [(0, np.inf)], "inf_inside": [(cols // 2, np.inf), (cols - 1, -np.inf)],
             "nan_and_inf": [(0, np.inf), (3, np.nan), (cols // 3, -np.inf)]}[poison]
    x = _nonfinite_x(cols, where)
    # SELL-C-sigma: the oracle's layout + its restatement of the reference kernel on a zeroed y
    for sigma in (64, 1024):
        want = orc.mult_sell(orc.sell_from_csr(rp, cj, va, 64, sigma), x)[:rows]
        m = Matrix.sell(rows, cols, rp, cj, va, S.F64, 64, sigma)
        for v in EXACT_SLICED:
            m.set_option("variant", v)
            assert _same_with_nans(m.mult(x), want), (name, poison, sigma, v)
        for v in REASSOC_SLICED:                       # other summation order, same operands: same NaN rows
            m.set_option("variant", v)
            y = m.mult(x)
            assert np.array_equal(np.isnan(y), np.isnan(want)), (name, poison, sigma, v)
        m.free()
    # SELLP / ELL at the wave's slice height
    want = orc.mult_sliced_ell(orc.sliced_ell_from_csr(rp, cj, va, 64), x)[:rows]
    for make in (lambda: Matrix.sellp(rows, cols, rp, cj, va, S.F64, 64),):
        m = make()
        for v in EXACT_SLICED:
            m.set_option("variant", v)
            assert _same_with_nans(m.mult(x), want), (name, poison, "sellp", v)
        m.free()


def test_nonfinite_x_multivec_sliced(orc):
    rows, cols, rp, cj, va = CASES["stencil27_9x10x11"]
    X = np.stack([_nonfinite_x(cols, [(0, np.nan)]), _nonfinite_x(cols, [(0, np.inf), (5, np.inf)]),
                  _nonfinite_x(cols, [])])
    S_ = orc.sell_from_csr(rp, cj, va, 64, 1024)
    want = np.stack([orc.mult_sell(S_, X[k])[:rows] for k in range(3)])
    m = Matrix.sell(rows, cols, rp, cj, va, S.F64, 64, 1024)
    for v in (0, 16, 22):
        m.set_option("variant", v)
        Y = m.mult_mv(X)
        for k in range(3):
            assert _same_with_nans(Y[k], want[k]), (v, k)
    m.free()


def test_stored_zero_columns_get_no_diagonal_descriptor():
    """A slice column that holds a real entry of value zero must keep reading its indices."""
    rows, cols, rp, cj, va = CASES["band_500_5"]
    a = Matrix.sell(rows, cols, rp, cj, va, S.F64, 64, 1024)
    rows, cols, rp, cj, vz = CASES["band_500_5_stored_zeros"]
    b = Matrix.sell(rows, cols, rp, cj, vz, S.F64, 64, 1024)
    assert a.info.diag_columns > 0
    assert b.info.diag_columns < a.info.diag_columns
    a.free(); b.free()


def test_norms_on_two_streams_do_not_share_scratch():
    L = S.lib()
    n = 3_000_001
    g = torch.Generator(device=DEV).manual_seed(9)
    a = torch.rand(n, dtype=torch.float64, device=DEV, generator=g)
    b = torch.rand(n // 3, dtype=torch.float64, device=DEV, generator=g) * 7.0
    outs = torch.zeros(4, dtype=torch.float64, device=DEV)
    assert L.spmv_hip_norm2_dev(n, a.data_ptr(), S.F64, outs[0:].data_ptr(), 0) == 0
    assert L.spmv_hip_norm2_dev(n // 3, b.data_ptr(), S.F64, outs[1:].data_ptr(), 0) == 0
    torch.cuda.synchronize()
    want_a, want_b = float(outs[0]), float(outs[1])
    assert abs(want_a - float(torch.linalg.vector_norm(a))) <= 1e-12 * want_a
    s1, s2 = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    for _ in range(50):
        outs[2:].zero_()
        torch.cuda.synchronize()
        assert L.spmv_hip_norm2_dev(n, a.data_ptr(), S.F64, outs[2:].data_ptr(), s1.cuda_stream) == 0
        assert L.spmv_hip_norm2_dev(n // 3, b.data_ptr(), S.F64, outs[3:].data_ptr(), s2.cuda_stream) == 0
        torch.cuda.synchronize()
        assert float(outs[2]) == want_a and float(outs[3]) == want_b
    # a program that creates and destroys streams gives the per-stream scratch back: free memory does not
    # shrink over many short-lived streams, and a released stream can be used again
    assert L.spmv_hip_release_stream(s1.cuda_stream) == 0 and L.spmv_hip_release_stream(s2.cuda_stream) == 0
    assert L.spmv_hip_release_stream(s1.cuda_stream) == 0            # nothing held: not an error
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info(DEV)[0]
    for _ in range(600):                                             # 600 x 8 KiB would not show; 600 x a 2 MiB page would
        s = torch.cuda.Stream(device=DEV)
        assert L.spmv_hip_norm2_dev(n // 3, b.data_ptr(), S.F64, outs[3:].data_ptr(), s.cuda_stream) == 0
        assert L.spmv_hip_release_stream(s.cuda_stream) == 0
    assert float(outs[3]) == want_b
    assert free_before - torch.cuda.mem_get_info(DEV)[0] < 64 << 20
    assert L.spmv_hip_norm2_dev(n, a.data_ptr(), S.F64, outs[2:].data_ptr(), s1.cuda_stream) == 0
    torch.cuda.synchronize()
    assert float(outs[2]) == want_a


def test_exact_order_keeps_the_bits_whatever_the_trial_picks(orc):
    rows, cols, rp, cj, va = G.stencil27(40, 41, 42)
    rp, cj, va = rp.numpy(), cj.numpy(), va.numpy()
    x = np.random.default_rng(2).random(cols)
    want = orc.mult_csr_base(rp, cj, va, x)
    for make in (lambda: Matrix.sell(rows, cols, rp, cj, va, S.F64, 64, 1024),
                 lambda: Matrix.sellp(rows, cols, rp, cj, va, S.F64, 64)):
        m = make()
        m.set_option("variant", 12)
        m.set_option("exact_order", 1)
        assert m.info.variant == 0                     # off the reassociating variant
        for _ in range(3):
            assert m.analyze(2) in EXACT_SLICED
            assert m.mult(x).tobytes() == want.tobytes()
        m.set_option("exact_order", 0)
        m.free()
    # vector CSR: lanes per row and kernel family stay, so the bits stay
    coo = G.csr_to_coo_numpy(rows, torch.from_numpy(rp), torch.from_numpy(cj), torch.from_numpy(va))
    m = Matrix.csr(rows, cols, coo, 1, S.F64, S.CSR_VECTOR, 0)
    lanes, variant = m.info.lanes_per_row, m.info.variant
    y0 = m.mult(x)
    m.set_option("exact_order", 1)
    for _ in range(3):
        m.analyze(2)
        assert m.info.lanes_per_row == lanes and (m.info.variant >= 5) == (variant >= 5)
        assert m.mult(x).tobytes() == y0.tobytes()
    m.free()
