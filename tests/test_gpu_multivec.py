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


def _band_csr(n, half, short_every=0, drop=0):
    """Row i holds the columns i-half .. i+half that exist; every `short_every`-th row lacks its last `drop` entries
    (in a sliced layout those positions are padding inside otherwise regular slice columns)."""
    rp, cj = [0], []
    for i in range(n):
        cols_i = [c for c in range(i - half, i + half + 1) if 0 <= c < n]
        if short_every and i % short_every == short_every - 1:
            cols_i = cols_i[:max(len(cols_i) - drop, 1)]
        cj += cols_i
        rp.append(len(cj))
    rng = np.random.default_rng(n + half)
    return (n, n, np.asarray(rp, np.int32), np.asarray(cj, np.int32), rng.uniform(0.5, 1.5, len(cj)))


def _grid9(nx, ny):
    """9-point stencil on an nx x ny grid: the window sort of SELL-C-sigma takes the two end rows of every grid line out
    of the interior rows, which leaves slices whose rows are neighbours except at one place."""
    import itertools
    offs = [(dy, dx) for dy, dx in itertools.product((-1, 0, 1), repeat=2)]
    rows, cols, rp, cj, va = G._stencil((ny, nx), offs, 8.0, -1.0, 0, nx * ny, "cpu")
    rng = np.random.default_rng(nx)
    return (rows, cols, rp.numpy().astype(np.int32), cj.numpy().astype(np.int32), rng.uniform(0.5, 1.5, int(rp[-1])))


RUN_CASES = {
    "band_1": _band_csr(700, 1),                        # one run of three columns
    "band_4": _band_csr(5000, 4),                       # three runs
    "band_7": _band_csr(3000, 7),                       # five runs: two pairs and a single one
    "band_13": _band_csr(2100, 13),                     # 27 columns like the 27-point stencil
    "band_4_short_rows": _band_csr(4000, 4, 50, 3),     # padding inside runs
    "band_5": _band_csr(1500, 5),                       # 11 columns: runs and two columns left over
    "grid9_200x30": _grid9(200, 30),                    # SELL-64-1024: slices with one gap between their rows
    "grid9_131x40": _grid9(131, 40),
    "band_15": _band_csr(1300, 15),                     # 31 columns: a patterned 64-row block is longer than one chunk
}


def _same_bits_or_both_nan(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    nan = np.isnan(a)
    return np.array_equal(nan, np.isnan(b)) and bits_equal(np.where(nan, 0.0, a), np.where(nan, 0.0, b))


@pytest.mark.parametrize("case", sorted(RUN_CASES))
def test_mult_mv_runs_of_neighbouring_diagonals_bitwise(orc, case):
    """Slices of 64 neighbouring rows whose columns run along neighbouring diagonals take the two-rows-per-lane
    kernel's shifted operands: same bits as n single products, for any stride, with padding and non-finite x."""
    rows, cols, rp, cj, va = RUN_CASES[case]
    rng = np.random.default_rng(11)
    builders = _builders(rows, cols, rp, cj, va)
    for name in ("sellp", "sell_1024", "ell", "sell_dfc", "csr_scalar", "csr_merge", "csr_bal", "csr_vector_pad4"):
        m = builders[name](S.F64)
        if not case.startswith("band") or "short" in case:
            pass                                              # fewer regular blocks / slices: whatever path they take
        elif name == "csr_scalar":
            assert m.info.pattern_blocks * 2 * 64 >= rows, (case, name)         # blocks walked by positions, with runs
        elif name == "sell_dfc":
            assert m.info.diag_columns > 0, (case, name)
        elif not name.startswith("csr"):
            assert m.info.diag_columns * 2 * 64 >= m.info.stored, (case, name)  # the kernel with the runs is the one launched
        for n in (1, 2, 3, 4, 5, 6, 7, 8, 13):
            X = rng.standard_normal((n, cols))
            assert bits_equal(m.mult_mv(X), orc.mult_mv_csr(rp, cj, va, X)), (case, name, n)
        # vectors an odd number of elements apart (pairs of operands and results off their 16-byte alignment)
        n, ldx, ldy = 8, cols + 37, rows + 11
        xb = torch.from_numpy(rng.standard_normal((n, ldx))).cuda()
        yb = torch.full((n, ldy), -7.0, dtype=torch.float64, device="cuda")
        m.mult_mv_dev(n, xb.data_ptr(), ldx, yb.data_ptr(), ldy, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = yb.cpu().numpy()
        assert bits_equal(np.ascontiguousarray(got[:, :rows]),
                          orc.mult_mv_csr(rp, cj, va, np.ascontiguousarray(xb.cpu().numpy()[:, :cols]))), (case, name)
        assert np.all(got[:, rows:] == -7.0)
        # x[0] is what the padding of a sliced layout multiplies (sellcs_format.c:167,171; sellp.c:440-452): NaN / inf
        # there, and inside the band -- against the oracle's restatement of the reference's sliced kernels
        if name not in ("ell", "csr_vector_pad4"):
            X = rng.standard_normal((4, cols))
            X[0, 0] = np.nan
            X[1, 0] = np.inf
            X[2, cols // 2] = np.inf
            X[3, cols - 1] = np.nan
            with np.errstate(invalid="ignore"):
                if name.startswith("csr") or name == "sell_dfc":
                    want = orc.mult_mv_csr(rp, cj, va, X)                       # no padding in these layouts
                elif name == "sellp":
                    layout = orc.sliced_ell_from_csr(rp, cj, va, 64)
                    want = np.stack([orc.mult_sliced_ell(layout, X[k])[:rows] for k in range(4)])
                else:
                    layout = orc.sell_from_csr(rp, cj, va, 64, 1024)
                    want = np.stack([orc.mult_sell(layout, X[k])[:rows] for k in range(4)])
            assert _same_bits_or_both_nan(m.mult_mv(X), want), (case, name)
        m.free()
        m = builders[name](S.F32)
        X32 = rng.standard_normal((6, cols)).astype(np.float32)
        Y = m.mult_mv(X32.astype(np.float64)).astype(np.float32)
        for v in range(6):
            assert bits_equal(Y[v], orc.mult_csr_base_f32(rp, cj, va.astype(np.float32), X32[v])), (case, name, v)
        m.free()
