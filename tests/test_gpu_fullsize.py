"""Parity at the sizes of the benchmark configurations, where the CPU oracle is
too slow to be the only check: size-independent properties plus device-side
cross-checks between independent kernels.

  C1  2-D Laplacian 1000^2 (1 000 000 rows, 4 996 000 nnz): every format against
      the oracle's mult_csr_base directly (the oracle needs ~10 ms here).
  C2  27-point stencil 252^3 (16 003 008 rows, 428 661 064 nnz): SELL-64-1024
      against the analytic product A*1 (row r: 26 - (len_r - 1)), bit-for-bit
      against the independent csr_scalar kernel, linearity, and the
      checksum identity sum(A x) = (A^T 1) . x with A symmetric.
  C3' R-MAT scale 22 (4 194 304 rows, skewed, empty rows): merge-path and
      balanced CSR against csr_scalar on the device and the oracle on the host.
Tolerances as in test_gpu_parity.py.
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


def test_c2_stencil_252_sell_properties():
    n = 252
    rows, cols, rp, cj, va = G.stencil27(n, device=DEV)
    assert rows == 16_003_008 and int(rp[-1]) == 428_661_064
    lens = (rp[1:] - rp[:-1]).to(torch.float64)
    sell = _dev_matrix(S.SELL, rows, cols, rp, cj, va, 64, 1024)
    info = sell.info
    assert info.nslices == 250_047 and info.stored >= info.nnz and info.stored < info.nnz * 1.01
    scalar = _dev_matrix(S.CSR_SCALAR, rows, cols, rp, cj, va)

    ones = torch.ones(cols, dtype=torch.float64, device=DEV)
    y1 = _mult(sell, ones, rows)
    assert torch.equal(y1, 26.0 - (lens - 1.0))                       # analytic product, exact in fp64

    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g) - 0.5
    z = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g) - 0.5
    y_sell = _mult(sell, x, rows)
    y_csr = _mult(scalar, x, rows)
    assert torch.equal(y_sell, y_csr)                                  # two independent kernels, same summation order
    # checksum of checksums: A is symmetric, so sum(A x) = (A 1) . x
    lhs, rhs = float(y_sell.sum()), float(torch.dot(y1, x))
    scale = float(torch.dot(26.0 + (lens - 1.0), x.abs()))
    assert abs(lhs - rhs) <= 1e-12 * scale
    # linearity
    yz = _mult(sell, z, rows)
    y_lin = _mult(sell, 2.0 * x - 3.0 * z, rows)
    assert _rel(y_lin, 2.0 * y_sell - 3.0 * yz) <= 1e-13
    # the tuned (reassociating) variant agrees within tolerance
    v = sell.analyze(2)
    assert _rel(_mult(sell, x, rows), y_csr) <= 1e-13, v
    sell.free(); scalar.free()


def test_c3_rmat_scale26_kernels_agree():
    """The full C3 size (67 108 864 rows, ~1.06e9 nnz): independent kernels against the bit-exact
    lane-per-row kernel, linearity, and the 64-lanes-per-row launch whose grid must be folded."""
    rows, cols, rp, cj, va = G.rmat(26, 16, device=DEV)
    nnz = int(rp[-1])
    assert rows == 1 << 26 and 1.0e9 < nnz < 2 ** 31
    g = torch.Generator(device=DEV).manual_seed(17)
    x = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g) - 0.5
    z = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g) - 0.5
    scalar = _dev_matrix(S.CSR_SCALAR, rows, cols, rp, cj, va)
    y_ref = _mult(scalar, x, rows)
    scalar.set_option("variant", 11)                                   # another chunk size, non-temporal loads
    assert torch.equal(_mult(scalar, x, rows), y_ref)
    scalar.free()
    for fmt, p0 in [(S.CSR_MERGE, 0), (S.CSR_BAL, 0), (S.CSR_VECTOR, 64), (S.CSR_VECTOR, 0)]:
        m = _dev_matrix(fmt, rows, cols, rp, cj, va, p0)
        y = _mult(m, x, rows)
        assert _rel(y, y_ref) <= 1e-13, (S.FORMAT_NAMES[fmt], p0)
        if fmt == S.CSR_MERGE:
            assert _rel(_mult(m, 2.0 * x - 3.0 * z, rows), 2.0 * y - 3.0 * _mult(m, z, rows)) <= 1e-13
            m.set_option("variant", 1)
            assert _rel(_mult(m, x, rows), y_ref) <= 1e-13
        if fmt in (S.CSR_MERGE, S.CSR_BAL):                            # columns relabeled by reference count
            y_plain = _mult(m, x, rows)
            m.set_option("hot_columns", 1)
            assert torch.equal(_mult(m, x, rows), y_plain)
            del y_plain
        if fmt == S.CSR_BAL:                                           # ... and dealt over the XCDs (537 M virtual rows)
            m.set_option("hot_columns", 2)
            assert _rel(_mult(m, x, rows), y_ref) <= 1e-13
        m.free()
        del y
        torch.cuda.empty_cache()


def test_c3_rmat_scale22_merge_and_balanced(orc):
    rows, cols, rp, cj, va = G.rmat(22, 16, device=DEV)
    lens = (rp[1:] - rp[:-1])
    assert rows == 1 << 22 and int(lens.max()) > 50 * float(lens.double().mean()) and int((lens == 0).sum()) > 0
    g = torch.Generator(device=DEV).manual_seed(13)
    x = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g) - 0.5
    scalar = _dev_matrix(S.CSR_SCALAR, rows, cols, rp, cj, va)
    y_ref = _mult(scalar, x, rows)
    want = orc.mult_csr_base(rp.cpu().numpy(), cj.cpu().numpy(), va.cpu().numpy(), x.cpu().numpy())
    assert y_ref.cpu().numpy().tobytes() == want.tobytes()
    for fmt, p0 in [(S.CSR_MERGE, 0), (S.CSR_MERGE, 2048), (S.CSR_BAL, 0), (S.CSR_BAL, 256), (S.CSR_VECTOR, 0)]:
        m = _dev_matrix(fmt, rows, cols, rp, cj, va, p0)
        y = _mult(m, x, rows)
        assert not torch.isnan(y).any()
        assert _rel(y, y_ref) <= 1e-13, (S.FORMAT_NAMES[fmt], p0)
        m.free()
    scalar.free()


def test_rmat_scale24_auto_choice_is_right_whatever_it_relabels():
    """create_auto on a graph matrix whose x (128 MB) is far beyond the L2: it may relabel the columns
    by reference count or deal them over the XCDs; y must stay within the reduction tolerance of the
    bit-exact lane-per-row kernel."""
    rows, cols, rp, cj, va = G.rmat(24, 16, device=DEV)
    nnz = int(rp[-1])
    g = torch.Generator(device=DEV).manual_seed(23)
    x = torch.rand(cols, dtype=torch.float64, device=DEV, generator=g) - 0.5
    scalar = _dev_matrix(S.CSR_SCALAR, rows, cols, rp, cj, va)
    y_ref = _mult(scalar, x, rows)
    scalar.free()
    st = torch.cuda.current_stream().cuda_stream
    m = Matrix.auto_from_device_csr(rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), S.F64, 2, st,
                                    keepalive=(rp, cj, va))
    info = m.info
    print("auto on rmat24:", S.FORMAT_NAMES[info.format], "variant", info.variant, "hot_columns", info.hot_columns)
    assert _rel(_mult(m, x, rows), y_ref) <= 1e-13
    assert _rel(_mult(m, 2.0 * x, rows), 2.0 * y_ref) <= 1e-13      # a second product: buffers are reused
    assert info.hot_columns in (1, 2)                               # measured 1.6x: not a close call
    m.free()
