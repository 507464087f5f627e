"""Parity at the sizes of the benchmark configurations, against the CPU oracle.

  C1  2-D Laplacian 1000^2 (1 000 000 rows, 4 996 000 nnz): every format against
      the oracle's mult_csr_base.
  C2  27-point stencil 252^3 (16 003 008 rows, 428 661 064 nnz): the CSR is copied
      to the host and multiplied by the oracle (mult_csr_base with the rows dealt
      over the host cores: same bits); SELL-64-1024 variant 0, the index-skipping
      variants and whatever the kernel trial picks, lane-per-row CSR and its
      patterned-block variants must equal it BIT FOR BIT; the reassociating kernels
      (16-byte-load SELL, vector CSR, merge-path / balanced tiles with and without
      tile patterns) within rel-L2 1e-13 and the component-wise bound
      |y - y_ref|_i <= 1e-12 (|A||x|)_i.  Plus the size-independent properties:
      the analytic product A*1, the checksum identity sum(A x) = (A 1) . x for the
      symmetric A, linearity.
  C3  R-MAT scale 26 (67 108 864 rows, ~1.06e9 nnz): the same against the oracle --
      lane-per-row CSR bit for bit; merge-path, balanced (plain, columns relabeled
      by reference count, columns dealt over the XCDs), vector CSR within the
      tolerances above.
  C3' R-MAT scale 22, scale 24 (automatic choice).
Reference bar: rel-L2 < 1e-10 (reference src/driver_spmv.c:278-289,313); oracle:
mult_csr_base (reference src/csr.c:4-13).
"""
import numpy as np
import pytest
import torch

import spmv_amd as S
from spmv_amd import generators as G
from spmv_amd.matrix import Matrix

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _rel(a, b):
    return float(torch.linalg.vector_norm(a - b) / torch.linalg.vector_norm(b))


def _dev_matrix(fmt, rows, cols, rp, cj, va, p0=0, p1=0, dtype=S.F64):
    st = torch.cuda.current_stream().cuda_stream
    return Matrix.from_device_csr(fmt, rows, cols, int(rp[-1]), rp.data_ptr(), cj.data_ptr(), va.data_ptr(), dtype, p0, p1,
                                  st, keepalive=(rp, cj, va))


def _mult(m, x, rows):
    y = torch.full((rows,), float("nan"), dtype=x.dtype, device=DEV)
    m.mult_dev(x.data_ptr(), y.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return y


def test_c1_laplacian_1000_every_format_against_oracle(orc):
    rows, cols, rp, cj, va = G.laplacian2d(1000, device=DEV)
    assert rows == 1_000_000 and int(rp[-1]) == 4_996_000
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g)
    want = orc.mult_csr_base(rp.cpu().numpy(), cj.cpu().numpy(), va.cpu().numpy(), x.cpu().numpy())
    for fmt, p0, p1, exact in [(S.CSR_SCALAR, 0, 0, True), (S.CSR_VECTOR, 0, 0, False), (S.CSR_MERGE, 0, 0, False),
                               (S.CSR_BAL, 0, 0, False), (S.SELL, 64, 16384, True), (S.SELLP, 0, 0, True), (S.ELL, 0, 0, True)]:
        m = _dev_matrix(fmt, rows, cols, rp, cj, va, p0, p1)
        y = _mult(m, x, rows).cpu().numpy()
        if exact:
            assert y.tobytes() == want.tobytes(), S.FORMAT_NAMES[fmt]
        else:
            assert orc.rel_l2(y, want) <= 1e-13, S.FORMAT_NAMES[fmt]
        if fmt in (S.SELL, S.SELLP, S.ELL):
            v = m.analyze(2)
            y = _mult(m, x, rows).cpu().numpy()
            assert orc.rel_l2(y, want) <= 1e-13, (S.FORMAT_NAMES[fmt], v)
        m.free()


EXACT_SLICED = (0, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31)
TOL_REL, TOL_COMPONENT = 1e-13, 1e-12


def _host_csr(rp, cj, va):
    return (np.ascontiguousarray(rp.cpu().numpy()), np.ascontiguousarray(cj.cpu().numpy()),
            np.ascontiguousarray(va.cpu().numpy()))


def _check_reassociated(orc, y_dev, want, bound, what):
    y = y_dev.cpu().numpy()
    assert orc.rel_l2(y, want) <= TOL_REL, what
    worst = float(np.max(np.abs(y - want) - TOL_COMPONENT * bound))
    assert worst <= 0.0, (what, worst)


def test_c2_stencil_252_against_the_oracle(orc):
    n = 252
    rows, cols, rp, cj, va = G.stencil27(n, device=DEV)
    assert rows == 16_003_008 and int(rp[-1]) == 428_661_064
    lens = (rp[1:] - rp[:-1]).to(torch.float64)
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g) - 0.5
    z = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g) - 0.5
    rp_h, cj_h, va_h = _host_csr(rp, cj, va)
    x_h = x.cpu().numpy()
    want = orc.mult_csr_base_rows_parallel(rp_h, cj_h, va_h, x_h)
    bound = orc.abs_product(rp_h, cj_h, va_h, x_h)
    del cj_h, va_h

    # ---- SELL-64-1024: the headline kernel family
    sell = _dev_matrix(S.SELL, rows, cols, rp, cj, va, 64, 1024)
    info = sell.info
    assert info.nslices == 250_047 and info.stored >= info.nnz and info.stored < info.nnz * 1.01
    assert info.diag_columns * 64 * 2 >= info.stored            # the index-skipping variants are on offer here
    for v in EXACT_SLICED:
        sell.set_option("variant", v)
        assert _mult(sell, x, rows).cpu().numpy().tobytes() == want.tobytes(), ("sell", v)
    for bmap in (1, 5, 7, 1000):                                # workgroups dealt over the matrix in other ways: same bits
        sell.set_option("block_map", bmap)
        for v in (0, 22):
            sell.set_option("variant", v)
            assert _mult(sell, x, rows).cpu().numpy().tobytes() == want.tobytes(), ("sell", v, "block_map", bmap)
    sell.set_option("block_map", 0)
    for v in (4, 12):                                           # 16-byte loads: two partial sums per row
        sell.set_option("variant", v)
        _check_reassociated(orc, _mult(sell, x, rows), want, bound, ("sell", v))
    v = sell.analyze(3)
    y_tuned = _mult(sell, x, rows)
    if v in EXACT_SLICED:
        assert y_tuned.cpu().numpy().tobytes() == want.tobytes(), ("sell tuned", v)
    else:
        _check_reassociated(orc, y_tuned, want, bound, ("sell tuned", v))
    # size-independent properties on the tuned kernel
    ones = torch.ones(cols, dtype=torch.float64, device=DEV)
    y1 = _mult(sell, ones, rows)
    assert torch.equal(y1, 26.0 - (lens - 1.0))                       # analytic product, exact in fp64
    lhs, rhs = float(y_tuned.sum()), float(torch.dot(y1, x))           # A symmetric: sum(A x) = (A 1) . x
    scale = float(torch.dot(26.0 + (lens - 1.0), x.abs()))
    assert abs(lhs - rhs) <= 1e-12 * scale
    yz = _mult(sell, z, rows)
    assert _rel(_mult(sell, 2.0 * x - 3.0 * z, rows), 2.0 * y_tuned - 3.0 * yz) <= 1e-13
    sell.free()
    del y1, yz, y_tuned, ones
    torch.cuda.empty_cache()

    # ---- SELLP: no row order, boundary rows stay inside their slices
    sellp = _dev_matrix(S.SELLP, rows, cols, rp, cj, va, 64)
    for v in (0, 22):
        sellp.set_option("variant", v)
        assert _mult(sellp, x, rows).cpu().numpy().tobytes() == want.tobytes(), ("sellp", v)
    # a block of vectors (two rows per lane, operands of neighbouring diagonals shifted over the lanes): every
    # result column has the bits of its single product
    st = torch.cuda.current_stream().cuda_stream
    X = torch.stack([x, z, 2.0 * x - 3.0 * z, x * z, x, z, -x, 0.5 * z, z - x, x + 1.0])
    Y = torch.empty(X.shape[0], rows, dtype=torch.float64, device=DEV)
    sellp.set_option("variant", 0)
    for nv in (8, 10, 2):
        Y.fill_(-1.0)
        sellp.mult_mv_dev(nv, X.data_ptr(), cols, Y.data_ptr(), rows, st)
        torch.cuda.synchronize()
        assert Y[0].cpu().numpy().tobytes() == want.tobytes(), ("sellp mult_mv", nv)
        for k in range(1, nv):
            assert torch.equal(Y[k], _mult(sellp, X[k], rows)), ("sellp mult_mv", nv, k)
    sellp.free()
    sell = _dev_matrix(S.SELL, rows, cols, rp, cj, va, 64, 1024)
    Y.fill_(-1.0)
    sell.mult_mv_dev(4, X.data_ptr(), cols, Y.data_ptr(), rows, st)
    torch.cuda.synchronize()
    assert Y[0].cpu().numpy().tobytes() == want.tobytes(), "sell mult_mv"
    for k in range(1, 4):
        assert torch.equal(Y[k], _mult(sell, X[k], rows)), ("sell mult_mv", k)
    sell.free()
    del X, Y
    torch.cuda.empty_cache()

    # ---- lane-per-row CSR, plain and with patterned row blocks
    scalar = _dev_matrix(S.CSR_SCALAR, rows, cols, rp, cj, va)
    assert scalar.info.pattern_blocks * 2 * 64 >= rows
    for v in (0, 11, 16, 22):
        scalar.set_option("variant", v)
        assert _mult(scalar, x, rows).cpu().numpy().tobytes() == want.tobytes(), ("csr_scalar", v)
    scalar.set_option("block_map", 5)
    assert _mult(scalar, x, rows).cpu().numpy().tobytes() == want.tobytes(), ("csr_scalar", 22, "block_map", 5)
    # a block of vectors: patterned blocks walked by positions with operands shifted over the lanes, the others by rows
    X = torch.stack([x, z, 2.0 * x - 3.0 * z, x * z, -x, 0.5 * z, z - x, x + 1.0])
    Y = torch.full((8, rows), -1.0, dtype=torch.float64, device=DEV)
    scalar.mult_mv_dev(8, X.data_ptr(), cols, Y.data_ptr(), rows, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert Y[0].cpu().numpy().tobytes() == want.tobytes(), "csr_scalar mult_mv"
    for k in range(1, 8):
        assert torch.equal(Y[k], _mult(scalar, X[k], rows)), ("csr_scalar mult_mv", k)
    del X, Y
    scalar.free()

    # ---- the reassociating CSR kernels, every variant that analyze may pick here
    for fmt, variants in ((S.CSR_MERGE, (0, 1, 2)), (S.CSR_BAL, (0, 1, 2)), (S.CSR_VECTOR, (0, 5, 6, 7, 8))):
        m = _dev_matrix(fmt, rows, cols, rp, cj, va)
        for v in variants:
            m.set_option("variant", v)
            _check_reassociated(orc, _mult(m, x, rows), want, bound, (S.FORMAT_NAMES[fmt], v))
        y0 = _mult(m, x, rows).clone()
        m.set_option("block_map", 3)
        assert torch.equal(_mult(m, x, rows), y0), (S.FORMAT_NAMES[fmt], "block_map", 3)
        m.free()


def test_c3_rmat_scale26_against_the_oracle(orc):
    """The full C3 size (67 108 864 rows, ~1.06e9 nnz) against the oracle's product of the same CSR."""
    rows, cols, rp, cj, va = G.rmat(26, 16, device=DEV)
    nnz = int(rp[-1])
    assert rows == 1 << 26 and 1.0e9 < nnz < 2 ** 31
    g = torch.Generator(device=DEV).manual_seed(17)
    x = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g) - 0.5
    z = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g) - 0.5
    rp_h, cj_h, va_h = _host_csr(rp, cj, va)
    x_h = x.cpu().numpy()
    want = orc.mult_csr_base_rows_parallel(rp_h, cj_h, va_h, x_h)
    bound = orc.abs_product(rp_h, cj_h, va_h, x_h)
    del cj_h, va_h, rp_h

    scalar = _dev_matrix(S.CSR_SCALAR, rows, cols, rp, cj, va)
    for v in (0, 11):                                                  # two chunk sizes, plain and non-temporal loads
        scalar.set_option("variant", v)
        assert _mult(scalar, x, rows).cpu().numpy().tobytes() == want.tobytes(), ("csr_scalar", v)
    scalar.free()
    for fmt, p0 in [(S.CSR_MERGE, 0), (S.CSR_BAL, 0), (S.CSR_VECTOR, 64), (S.CSR_VECTOR, 0)]:
        m = _dev_matrix(fmt, rows, cols, rp, cj, va, p0)
        y = _mult(m, x, rows)
        _check_reassociated(orc, y, want, bound, (S.FORMAT_NAMES[fmt], p0))
        if fmt == S.CSR_MERGE:
            assert _rel(_mult(m, 2.0 * x - 3.0 * z, rows), 2.0 * y - 3.0 * _mult(m, z, rows)) <= 1e-13
            m.set_option("variant", 1)
            _check_reassociated(orc, _mult(m, x, rows), want, bound, ("csr_merge", "variant 1"))
        if fmt in (S.CSR_MERGE, S.CSR_BAL):                            # columns relabeled by reference count: same sums
            y_plain = _mult(m, x, rows)                                # under the variant now set
            m.set_option("hot_columns", 1)
            assert torch.equal(_mult(m, x, rows), y_plain)
            del y_plain
            _check_reassociated(orc, _mult(m, x, rows), want, bound, (S.FORMAT_NAMES[fmt], "hot_columns 1"))
        if fmt == S.CSR_BAL:                                           # ... and dealt over the XCDs (537 M virtual rows)
            m.set_option("hot_columns", 2)
            _check_reassociated(orc, _mult(m, x, rows), want, bound, ("csr_bal", "hot_columns 2"))
            for hv in (3, 4):                                          # each group's hottest labels of x in LDS
                m.set_option("hot_variant", hv)
                _check_reassociated(orc, _mult(m, x, rows), want, bound, ("csr_bal", "hot_columns 2", "tile kernel", hv))
        m.free()
        del y
        torch.cuda.empty_cache()


def test_c3_rmat_scale22_merge_and_balanced(orc):
    rows, cols, rp, cj, va = G.rmat(22, 16, device=DEV)
    lens = (rp[1:] - rp[:-1])
    assert rows == 1 << 22 and int(lens.max()) > 50 * float(lens.double().mean()) and int((lens == 0).sum()) > 0
    g = torch.Generator(device=DEV).manual_seed(13)
    x = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g) - 0.5
    rp_h, cj_h, va_h = _host_csr(rp, cj, va)
    want = orc.mult_csr_base(rp_h, cj_h, va_h, x.cpu().numpy())
    bound = orc.abs_product(rp_h, cj_h, va_h, x.cpu().numpy())
    scalar = _dev_matrix(S.CSR_SCALAR, rows, cols, rp, cj, va)
    assert _mult(scalar, x, rows).cpu().numpy().tobytes() == want.tobytes()
    scalar.free()
    for fmt, p0 in [(S.CSR_MERGE, 0), (S.CSR_MERGE, 2048), (S.CSR_BAL, 0), (S.CSR_BAL, 256), (S.CSR_VECTOR, 0)]:
        m = _dev_matrix(fmt, rows, cols, rp, cj, va, p0)
        y = _mult(m, x, rows)
        assert not torch.isnan(y).any()
        _check_reassociated(orc, y, want, bound, (S.FORMAT_NAMES[fmt], p0))
        m.free()


def test_rmat_scale24_auto_choice_is_right_whatever_it_relabels(orc):
    """create_auto on a graph matrix whose x (128 MB) is far beyond the L2: it may relabel the columns
    by reference count or deal them over the XCDs; y must stay within the reduction tolerance of the
    oracle's mult_csr_base."""
    rows, cols, rp, cj, va = G.rmat(24, 16, device=DEV)
    nnz = int(rp[-1])
    g = torch.Generator(device=DEV).manual_seed(23)
    x = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g) - 0.5
    rp_h, cj_h, va_h = _host_csr(rp, cj, va)
    want = orc.mult_csr_base_rows_parallel(rp_h, cj_h, va_h, x.cpu().numpy())
    bound = orc.abs_product(rp_h, cj_h, va_h, x.cpu().numpy())
    del cj_h, va_h
    st = torch.cuda.current_stream().cuda_stream
    m = Matrix.auto_from_device_csr(rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), S.F64, 2, st,
                                    keepalive=(rp, cj, va))
    info = m.info
    print("auto on rmat24:", S.FORMAT_NAMES[info.format], "variant", info.variant, "hot_columns", info.hot_columns)
    _check_reassociated(orc, _mult(m, x, rows), want, bound, "auto")
    _check_reassociated(orc, _mult(m, 2.0 * x, rows), 2.0 * want, 2.0 * bound, "auto, second product")   # buffers are reused
    assert info.hot_columns in (1, 2)                               # measured 1.6x: not a close call
    m.free()
