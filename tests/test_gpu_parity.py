"""GPU parity: every format of libspmv_hip (through its C ABI) against the
golden fixtures of the compiled reference and against the CPU oracle.

Tolerances (fp64): lane-per-row kernels (csr_scalar, SELL, SELLP, ELL) keep
mult_csr_base's summation order and must be BIT-EXACT with the reference's
y_csr_base; kernels that reassociate (csr_vector, csr_merge, csr_bal) must meet
rel-L2 <= 1e-13, far inside the reference's own bar of 1e-10
(reference src/driver_spmv.c:313).  fp32 (no reference counterpart): rel-L2
<= 2e-6 against the oracle's float restatement and <= 1e-5 against the fp64
oracle.  Layouts (row_order, slice pointers, columns, values, tile
coordinates) are compared bit for bit.
"""
import numpy as np
import pytest

import spmv_amd as S
from spmv_amd import generators as G
from spmv_amd.matrix import Matrix, coo_to_csr_host
from tests import cases as CS
from tests.util import bits_equal, coo_of, golden_names, load_golden

pytestmark = pytest.mark.gpu

NAMES = golden_names()
TOL_REASSOC = 1e-13
TOL_F32_VS_F32 = 2e-6
TOL_F32_VS_F64 = 1e-5


@pytest.fixture(scope="module", params=NAMES)
def g(request):
    d = load_golden(request.param)
    d["name"] = request.param
    return d


def rel(orc, y, ref):
    return orc.rel_l2(np.asarray(y, np.float64), np.asarray(ref, np.float64))


def test_library_loaded_is_the_in_tree_hip_build():
    assert S.LIB_PATH.endswith("spmv_amd/libspmv_hip.so")
    S.lib()


def test_create_sorts_coo_in_place_like_reference(g):
    coo = coo_of(g, "in")
    m = Matrix.csr(int(g["rows"]), int(g["cols"]), coo, 1, S.F64, S.CSR_SCALAR)
    assert bits_equal(coo["i"], g["coo_sorted_i"]) and bits_equal(coo["j"], g["coo_sorted_j"])
    assert bits_equal(coo["a"], g["coo_sorted_a"])
    m.free()


def test_csr_scalar_bitwise(g):
    m = Matrix.csr(int(g["rows"]), int(g["cols"]), coo_of(g, "in"), 1, S.F64, S.CSR_SCALAR)
    assert bits_equal(m.download(S.ARR_ROW_PTR), g["csr1_row_ptr"])
    assert bits_equal(m.download(S.ARR_COL), g["csr1_col"])
    assert bits_equal(m.download(S.ARR_VAL), g["csr1_val"])
    y = m.mult(g["x"])
    assert bits_equal(y, g["y_csr_base"])
    assert bits_equal(y, g["y_coo"])            # the driver's own check (driver_spmv.c:274-289)


# LDS-staged chunk sizes 0, 2, 3, 4 (+8: non-temporal loads), 1 = plain
SCALAR_VARIANTS = [0, 1, 2, 3, 4, 8, 10, 11, 12]
# the same chunked kernel reading one row's indices per block of 64 rows that repeat a pattern
SCALAR_PATTERN_VARIANTS = [16, 17, 18, 20, 21, 22]


@pytest.mark.parametrize("alignment", [1, 4])
def test_csr_scalar_variants_all_bitwise(orc, g, alignment):
    rows, cols = int(g["rows"]), int(g["cols"])
    csr = (g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"])
    x32 = g["x"].astype(np.float32)
    y32 = orc.mult_csr_base_f32(csr[0], csr[1], csr[2].astype(np.float32), x32)
    for dtype, xin, want in ((S.F64, g["x"], g["y_csr_base"]), (S.F32, x32.astype(np.float64), y32)):
        m = Matrix.csr(rows, cols, coo_of(g), alignment, dtype, S.CSR_SCALAR)
        for v in SCALAR_VARIANTS + SCALAR_PATTERN_VARIANTS:
            m.set_option("variant", v)
            y = m.mult(xin)
            assert bits_equal(y.astype(want.dtype), want), (v, dtype)
        assert m.analyze(2) in SCALAR_VARIANTS + SCALAR_PATTERN_VARIANTS
        assert bits_equal(m.mult(xin).astype(want.dtype), want)
        with pytest.raises(S.SpmvHipError, match="variant"):
            m.set_option("variant", 99)
            m.mult(xin)
        m.free()


@pytest.mark.parametrize("lanes", [0, 2, 4, 8, 16, 32, 64])
def test_csr_vector(orc, g, lanes):
    # alignment 4 = the reference's csr_vec mode (driver_spmv.c:189-191)
    m = Matrix.csr(int(g["rows"]), int(g["cols"]), coo_of(g), 4, S.F64, S.CSR_VECTOR, lanes)
    assert bits_equal(m.download(S.ARR_ROW_PTR), g["csr4_row_ptr"])
    assert bits_equal(m.download(S.ARR_COL), g["csr4_col"])
    assert bits_equal(m.download(S.ARR_VAL), g["csr4_val"])
    y = m.mult(g["x"])
    assert rel(orc, y, g["y_csr_base"]) <= TOL_REASSOC
    assert rel(orc, y, g["y_csr_vec"]) <= TOL_REASSOC
    # lanes left to the library: the LDS-staged kernel (variant 5) when rows average at most 32 entries
    mean = len(g["csr4_col"]) / max(int(g["rows"]), 1)
    assert m.info.variant == (0 if lanes or mean > 32 else 5)
    m.set_option("variant", 1)
    y1 = m.mult(g["x"])
    assert rel(orc, y1, g["y_csr_base"]) <= TOL_REASSOC
    for v in (2, 3, 4):                             # 1, 2, 4, 8 rows side by side per lane group: same sums
        m.set_option("variant", v)
        assert bits_equal(m.mult(g["x"]), y1), v
    for v in (5, 6):                                # LDS-staged kernel for short rows (min(lanes, 8) lanes per row)
        m.set_option("variant", v)
        assert rel(orc, m.mult(g["x"]), g["y_csr_base"]) <= TOL_REASSOC, v
    y5 = m.mult(g["x"])                             # variant 6
    for v in (7, 8):                                # ... reading one row's indices per patterned block: same sums
        m.set_option("variant", v)
        assert bits_equal(m.mult(g["x"]), y5), v


@pytest.mark.parametrize("items", [0, 256, 2048])
def test_csr_merge(orc, g, items):
    rows = int(g["rows"])
    m = Matrix.csr_merge(rows, int(g["cols"]), coo_of(g), S.F64, items)
    assert m.info.items_per_tile == (items or 512)
    info = m.info
    rp = g["csr1_row_ptr"]
    nnz = len(g["csr1_col"])
    coords = m.download(S.ARR_TILE_COORD).reshape(-1, 2)
    assert info.ntiles == -(-(rows + nnz) // info.items_per_tile)
    for t in range(info.ntiles + 1):               # csr_merge.c:25-30 on tile diagonals
        d = min(t * info.items_per_tile, rows + nnz)
        assert tuple(coords[t]) == orc.merge_path_search(d, rp, nnz)
    y = m.mult(g["x"])
    assert rel(orc, y, g["y_csr_base"]) <= TOL_REASSOC
    assert rel(orc, y, g["y_csr_merge_t8"]) <= TOL_REASSOC
    m.set_option("variant", 1)                      # line-aligned non-temporal loads
    y1 = m.mult(g["x"])
    assert rel(orc, y1, g["y_csr_base"]) <= TOL_REASSOC
    if len(g["csr1_col"]):
        m.set_option("variant", 2)                  # patterned tiles read one row's indices: same sums
        assert bits_equal(m.mult(g["x"]), y1)
    assert m.analyze(2) in (0, 1, 2)
    with pytest.raises(S.SpmvHipError, match="variant"):
        m.set_option("variant", 7)
        m.mult(g["x"])


@pytest.mark.parametrize("chunk", [0, 256, 1024])
def test_csr_bal_handles_empty_rows_too(orc, g, chunk):
    m = Matrix.csr_bal(int(g["rows"]), int(g["cols"]), coo_of(g), S.F64, chunk)
    y = m.mult(g["x"])
    assert not np.isnan(y).any()
    assert rel(orc, y, g["y_csr_base"]) <= TOL_REASSOC
    if "y_csr_bal_t8" in g:
        assert rel(orc, y, g["y_csr_bal_t8"]) <= TOL_REASSOC
    m.set_option("variant", 1)
    y1 = m.mult(g["x"])
    assert rel(orc, y1, g["y_csr_base"]) <= TOL_REASSOC
    if len(g["csr1_col"]):
        m.set_option("variant", 2)
        assert bits_equal(m.mult(g["x"]), y1)


def test_csr_vector_analyze_picks_a_group_size(orc, g):
    m = Matrix.csr(int(g["rows"]), int(g["cols"]), coo_of(g), 1, S.F64, S.CSR_VECTOR)
    m.analyze(2)
    assert m.info.lanes_per_row in (2, 4, 8, 16, 32, 64)
    assert rel(orc, m.mult(g["x"]), g["y_csr_base"]) <= TOL_REASSOC


@pytest.mark.parametrize("sigma", CS.SELL_SIGMAS)
def test_sell_layout_and_product(g, sigma):
    rows = int(g["rows"])
    m = Matrix.sell(rows, int(g["cols"]), g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"], S.F64, 64, sigma)
    assert m.info.slice_height == 64 and m.info.sigma == sigma
    assert bits_equal(m.download(S.ARR_ROW_ORDER), g[f"sell_s{sigma}_row_order"])
    assert bits_equal(m.download(S.ARR_SLICE_PTR), g[f"sell_s{sigma}_slice_ptr"])
    assert bits_equal(m.download(S.ARR_COL), g[f"sell_s{sigma}_col"])
    assert bits_equal(m.download(S.ARR_VAL), g[f"sell_s{sigma}_val"])
    y = m.mult(g["x"])
    assert bits_equal(y, g["y_csr_base"])
    if f"y_sell_s{sigma}" in g:
        assert bits_equal(y, g[f"y_sell_s{sigma}"])


@pytest.mark.parametrize("sigma", [64, 1024])
def test_sell_dfc_layout_and_product(orc, g, sigma):
    """SELL-C-sigma without padding: the layout of the reference's csr_to_sellcs_dfc (oracle restatement,
    itself checked against the compiled reference) bit for bit; product in mult_csr_base's order."""
    rows, cols = int(g["rows"]), int(g["cols"])
    csr = (g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"])
    want = orc.sell_dfc_from_csr(*csr, 64, sigma)
    m = Matrix.sell_dfc(rows, cols, *csr, S.F64, 64, sigma)
    info = m.info
    assert info.format == S.SELL_DFC and info.stored == info.nnz == len(csr[1]) and info.slice_height == 64
    assert bits_equal(m.download(S.ARR_ROW_ORDER), want["row_order"])
    assert bits_equal(m.download(S.ARR_VOP_PTR), want["vop_ptr"])
    assert bits_equal(m.download(S.ARR_VOP_LEN), want["vop_len"])
    assert bits_equal(m.download(S.ARR_SLICE_PTR), want["slice_ptr"])
    assert bits_equal(m.download(S.ARR_COL), want["col"])
    assert bits_equal(m.download(S.ARR_VAL), want["val"])
    for v in range(13):                             # masked-load and LDS-staged handling of the partial columns;
        m.set_option("variant", v)                  # 9..12: indices of columns along a diagonal not read
        assert bits_equal(m.mult(g["x"]), g["y_csr_base"]), v
    assert 0 <= info.diag_columns <= int(want["vop_ptr"][-1])
    assert m.analyze(2) in range(13)
    m.free()
    x32 = g["x"].astype(np.float32)
    m = Matrix.sell_dfc(rows, cols, *csr, S.F32, 64, sigma)
    y32 = orc.mult_csr_base_f32(csr[0], csr[1], csr[2].astype(np.float32), x32)
    for v in (0, 9, 12):
        m.set_option("variant", v)
        assert bits_equal(m.mult(x32.astype(np.float64)).astype(np.float32), y32), v
    m.free()


def test_sell_window_with_a_very_long_row(orc):
    """A row of 2^17 entries or more takes the counting path of the window sort: same order as the reference's."""
    n = (1 << 17) + 640
    dense = 777
    i = np.concatenate([np.arange(n), np.full(n, dense)]).astype(np.int32)
    j = np.concatenate([np.arange(n), np.arange(n)]).astype(np.int32)
    keep = np.ones(2 * n, bool)
    keep[n + dense] = False                                        # (dense, dense) only once
    rng = np.random.default_rng(3)
    coo = S.make_coo(i[keep], j[keep], rng.standard_normal(int(keep.sum())))
    S.sort_coo(coo)
    rp, cj, va = coo_to_csr_host(n, coo)
    x = rng.standard_normal(n)
    want = orc.mult_csr_base(rp, cj, va, x)
    for sigma in (1024, 16384):
        m = Matrix.sell(n, n, rp, cj, va, S.F64, 64, sigma)
        ref = orc.sell_from_csr(rp, cj, va, 64, sigma)            # sellcs_format.c:138-201 restated
        assert bits_equal(m.download(S.ARR_ROW_ORDER), ref["row_order"])
        assert bits_equal(m.download(S.ARR_SLICE_PTR), ref["slice_ptr"])
        assert bits_equal(m.mult(x), want)
        m.free()


def test_sell_windows_longer_than_the_lds_sort(orc):
    """Sorting windows above 32768 rows (one window over the whole matrix included) go through a stable radix
    sort on (window, longer first): same row_order and slices as the reference's window sort
    (radix_sort.c:103-122, restated by the oracle), same bits in y."""
    rng = np.random.default_rng(11)
    n = 100_000
    lens = rng.integers(0, 9, n)
    lens[rng.integers(0, n, 50)] = rng.integers(40, 200, 50)       # a few long rows, ties everywhere else
    rp = np.zeros(n + 1, np.int32)
    rp[1:] = np.cumsum(lens)
    cj = np.concatenate([np.sort(rng.choice(n, l, replace=False)) for l in lens]).astype(np.int32)
    va = rng.standard_normal(len(cj))
    x = rng.standard_normal(n)
    want = orc.mult_csr_base(rp, cj, va, x)
    for sigma in (65536, 1 << 20):
        ref = orc.sell_from_csr(rp, cj, va, 64, sigma)
        m = Matrix.sell(n, n, rp, cj, va, S.F64, 64, sigma)
        assert bits_equal(m.download(S.ARR_ROW_ORDER), ref["row_order"])
        assert bits_equal(m.download(S.ARR_SLICE_PTR), ref["slice_ptr"])
        assert bits_equal(m.download(S.ARR_COL), ref["col"])
        assert bits_equal(m.mult(x), want)
        m.free()
        d = Matrix.sell_dfc(n, n, rp, cj, va, S.F64, 64, sigma)
        assert bits_equal(d.mult(x), want)
        d.free()


SLICED_VARIANTS = [1, 2, 3, 4, 9, 10, 11, 12]
# lane per row, one fma per column in order, column indices of diagonal slice columns not read
DIAG_VARIANTS = [16, 17, 18, 19, 20, 21, 22, 23]


@pytest.mark.parametrize("fmt", ["sell", "sellp"])
def test_sliced_wide_variants_and_analyze(orc, g, fmt):
    """16-byte-load kernels reassociate the row sums: tolerance parity; analyze() picks one of them or 0."""
    rows, cols = int(g["rows"]), int(g["cols"])
    csr = (g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"])
    for dtype, tol, ref in ((S.F64, TOL_REASSOC, g["y_csr_base"]),
                            (S.F32, TOL_F32_VS_F32, orc.mult_csr_base_f32(csr[0], csr[1], csr[2].astype(np.float32),
                                                                          g["x"].astype(np.float32)))):
        m = Matrix.sell(rows, cols, *csr, dtype, 64, 1024) if fmt == "sell" else Matrix.sellp(rows, cols, *csr, dtype, 64)
        xin = g["x"].astype(np.float32).astype(np.float64) if dtype == S.F32 else g["x"]
        for v in SLICED_VARIANTS:
            m.set_option("variant", v)
            assert m.info.variant == v
            y = m.mult(xin)
            assert not np.isnan(y).any()
            assert rel(orc, y, ref) <= tol, (v, dtype)
        chosen = m.analyze(2)
        assert chosen in [0] + SLICED_VARIANTS + DIAG_VARIANTS + [24, 25, 26, 27, 28, 29, 30, 31]
        assert rel(orc, m.mult(xin), ref) <= tol
        with pytest.raises(S.SpmvHipError, match="variant"):
            m.set_option("variant", 99)
            m.mult(xin)
        m.free()


@pytest.mark.parametrize("fmt", ["sell", "sell64", "sellp", "ell"])
def test_sliced_diagonal_variants_bitwise(orc, g, fmt):
    """The kernels that replace the indices of a diagonal slice column by one distance keep the
    lane-per-row order: bit-identical to the reference's y_csr_base, whatever share of the columns
    is diagonal (none on the random fixtures, nearly all on the grids)."""
    rows, cols = int(g["rows"]), int(g["cols"])
    csr = (g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"])
    want32 = orc.mult_csr_base_f32(csr[0], csr[1], csr[2].astype(np.float32), g["x"].astype(np.float32))
    for dtype in (S.F64, S.F32):
        if fmt == "sell":
            m = Matrix.sell(rows, cols, *csr, dtype, 64, 1024)
        elif fmt == "sell64":
            m = Matrix.sell(rows, cols, *csr, dtype, 64, 64)
        elif fmt == "sellp":
            m = Matrix.sellp(rows, cols, *csr, dtype, 64)
        else:
            m = Matrix.ell(rows, cols, coo_of(g, "in"), dtype, 64)
        info = m.info
        assert 0 <= info.diag_columns <= info.stored // 64
        if g["name"] in ("band_500_5", "diag_64") and fmt != "sell":      # small grids break off at every grid line
            assert info.diag_columns * 64 >= 0.5 * info.stored, (g["name"], info.diag_columns, info.stored)
        xin = g["x"].astype(np.float32).astype(np.float64) if dtype == S.F32 else g["x"]
        for v in DIAG_VARIANTS:
            m.set_option("variant", v)
            y = m.mult(xin)
            if dtype == S.F64:
                assert bits_equal(y, g["y_csr_base"]), (fmt, v)
            else:
                assert bits_equal(y.astype(np.float32), want32), (fmt, v)
        m.free()


@pytest.mark.parametrize("height", [4, 64, 16])
def test_sellp(g, orc, height):
    rows = int(g["rows"])
    m = Matrix.sellp(rows, int(g["cols"]), g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"], S.F64, height)
    if height == 4:                                 # the reference's _BLOCK (AVX2)
        assert bits_equal(m.download(S.ARR_SLICE_PTR), g["sellp4_block_ptr"])
        assert bits_equal(m.download(S.ARR_COL), g["sellp4_j"])
        assert bits_equal(m.download(S.ARR_VAL), g["sellp4_A"])
    else:
        E = orc.sliced_ell_from_csr(g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"], height)
        assert bits_equal(m.download(S.ARR_SLICE_PTR), E["block_ptr"])
        assert bits_equal(m.download(S.ARR_COL), E["j"])
        assert bits_equal(m.download(S.ARR_VAL), E["A"])
    y = m.mult(g["x"])
    assert bits_equal(y, g["y_csr_base"])
    assert bits_equal(y, g["y_sellp_autovec"])


@pytest.mark.parametrize("height", [4, 0])
def test_ell(g, height):
    m = Matrix.ell(int(g["rows"]), int(g["cols"]), coo_of(g, "in"), S.F64, height)
    if height == 4:                                 # SIMD_WIDTH of ell.c:4; reference j is int64
        assert bits_equal(m.download(S.ARR_SLICE_PTR), g["ell4_block_ptr"])
        assert bits_equal(m.download(S.ARR_COL).astype(np.int64), g["ell4_j"])
        assert bits_equal(m.download(S.ARR_VAL), g["ell4_A"])
    else:
        assert m.info.slice_height == 64
    y = m.mult(g["x"])
    assert bits_equal(y, g["y_ell"])


FP32_BUILDERS = {
    "csr_scalar": lambda r, c, coo, csr: Matrix.csr(r, c, coo, 1, S.F32, S.CSR_SCALAR),
    "csr_vector": lambda r, c, coo, csr: Matrix.csr(r, c, coo, 1, S.F32, S.CSR_VECTOR),
    "csr_merge": lambda r, c, coo, csr: Matrix.csr_merge(r, c, coo, S.F32),
    "csr_bal": lambda r, c, coo, csr: Matrix.csr_bal(r, c, coo, S.F32),
    "sell": lambda r, c, coo, csr: Matrix.sell(r, c, *csr, S.F32, 64, 1024),
    "sellp": lambda r, c, coo, csr: Matrix.sellp(r, c, *csr, S.F32),
    "ell": lambda r, c, coo, csr: Matrix.ell(r, c, coo, S.F32),
}


@pytest.mark.parametrize("fmt", sorted(FP32_BUILDERS))
def test_fp32(orc, g, fmt):
    rows, cols = int(g["rows"]), int(g["cols"])
    csr = (g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"])
    m = FP32_BUILDERS[fmt](rows, cols, coo_of(g), csr)
    assert m.info.dtype == S.F32
    x32 = g["x"].astype(np.float32)
    y = m.mult(x32.astype(np.float64))
    y32 = orc.mult_csr_base_f32(csr[0], csr[1], csr[2].astype(np.float32), x32)
    if fmt in ("csr_scalar", "sell", "sellp", "ell"):
        assert bits_equal(y.astype(np.float32), y32)
    else:
        assert rel(orc, y, y32) <= TOL_F32_VS_F32
    assert rel(orc, y, g["y_csr_base"]) <= TOL_F32_VS_F64


def test_modes_of_the_reference_driver(orc, g):
    """MODE strings of driver_spmv.c:186-261 each give y == y_ref within its bar (:313)."""
    rows, cols = int(g["rows"]), int(g["cols"])
    for mode in S.MODES:
        m = S.create(mode, rows, cols, coo_of(g, "in"))
        y = m.mult(g["x"])
        assert rel(orc, y, g["y_coo"]) <= TOL_REASSOC, mode
        if mode == "coo":                           # mult_coo (coo.c:3-14) sums like mult_csr_base: same bits
            assert bits_equal(y, g["y_coo"])
        m.free()
    with pytest.raises(ValueError):
        S.create("no_such_mode", rows, cols, coo_of(g))


AUTO_FORMATS = (S.CSR_MERGE, S.CSR_SCALAR, S.CSR_VECTOR, S.SELLP, S.SELL)


@pytest.mark.parametrize("dtype", [S.F64, S.F32])
def test_auto_format_choice(orc, g, dtype):
    """create_auto returns one of its candidates, tuned; the product meets the bar of the format chosen."""
    import torch
    rows, cols = int(g["rows"]), int(g["cols"])
    csr = (g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"])
    nnz = len(csr[1])
    if dtype == S.F64:
        xin, want, tol = g["x"], g["y_csr_base"], TOL_REASSOC
    else:
        x32 = g["x"].astype(np.float32)
        xin, tol = x32.astype(np.float64), TOL_F32_VS_F32
        want = orc.mult_csr_base_f32(csr[0], csr[1], csr[2].astype(np.float32), x32)
    coo = coo_of(g, "in")
    m = S.create("auto", rows, cols, coo, dtype)
    assert bits_equal(coo["i"], g["coo_sorted_i"])            # sorted in place like every creator
    info = m.info
    assert info.format in AUTO_FORMATS and info.nnz == nnz and info.dtype == dtype
    if info.format in (S.SELL, S.SELLP):
        assert info.stored <= 2 * max(nnz, 1)
    if info.format == S.SELL:
        assert info.sigma in (1024, 16384) and info.slice_height == 64      # the two windows the choice builds
    assert rel(orc, m.mult(xin), want) <= tol
    m.free()
    # the same choice over CSR arrays that stay in HBM
    dev = torch.device("cuda:0")
    rp = torch.from_numpy(csr[0]).to(dev)
    cj = torch.from_numpy(csr[1]).to(dev) if nnz else torch.zeros(1, dtype=torch.int32, device=dev)
    td = torch.float64 if dtype == S.F64 else torch.float32
    va = torch.from_numpy(csr[2]).to(dev).to(td) if nnz else torch.zeros(1, dtype=td, device=dev)
    m = Matrix.auto_from_device_csr(rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), dtype, 2, 0,
                                    keepalive=(rp, cj, va))
    assert m.info.format in AUTO_FORMATS
    assert rel(orc, m.mult(xin), want) <= tol
    m.free()


def test_device_resident_path(orc, g):
    """CSR handed over in HBM, x/y resident, launches on torch's stream."""
    import torch
    rows, cols = int(g["rows"]), int(g["cols"])
    dev = torch.device("cuda:0")
    rp = torch.from_numpy(g["csr1_row_ptr"]).to(dev)
    cj = torch.from_numpy(g["csr1_col"]).to(dev)
    va = torch.from_numpy(g["csr1_val"]).to(dev)
    if cj.numel() == 0:
        cj = torch.zeros(1, dtype=torch.int32, device=dev); va = torch.zeros(1, dtype=torch.float64, device=dev)
    x = torch.from_numpy(g["x"]).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    nnz = len(g["csr1_col"])
    for fmt, p0, p1, exact in [(S.CSR_SCALAR, 0, 0, True), (S.CSR_VECTOR, 0, 0, False), (S.CSR_MERGE, 0, 0, False),
                               (S.CSR_BAL, 0, 0, False), (S.SELL, 64, 1024, True), (S.SELLP, 0, 0, True),
                               (S.ELL, 0, 0, True)]:
        m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), S.F64, p0, p1, st,
                                   keepalive=(rp, cj, va))
        y = torch.full((max(rows, 1),), float("nan"), dtype=torch.float64, device=dev)
        m.mult_dev(x.data_ptr(), y.data_ptr(), st)
        torch.cuda.synchronize()
        yh = y[:rows].cpu().numpy()
        if exact:
            assert bits_equal(yh, g["y_csr_base"]), S.FORMAT_NAMES[fmt]
        else:
            assert rel(orc, yh, g["y_csr_base"]) <= TOL_REASSOC, S.FORMAT_NAMES[fmt]
        m.free()


def test_norm_and_scale_match_driver(orc, g):
    import ctypes as C
    import torch
    n = int(g["cols"])
    dev = torch.device("cuda:0")
    v = torch.from_numpy(g["x"] * 3.0 + 0.25).to(dev)
    out = torch.zeros(1, dtype=torch.float64, device=dev)
    xs = torch.empty_like(v)
    L = S.lib()
    assert L.spmv_hip_norm2_dev(n, v.data_ptr(), S.F64, out.data_ptr(), 0) == 0
    assert L.spmv_hip_scale_dev(n, v.data_ptr(), xs.data_ptr(), S.F64, out.data_ptr(), 0) == 0
    torch.cuda.synchronize()
    want = orc.lib().orc_vec_norm(n, np.ascontiguousarray(v.cpu().numpy()))     # driver_spmv.c:21-26
    assert abs(out.item() - want) <= 1e-14 * max(want, 1.0)
    np.testing.assert_allclose(xs.cpu().numpy(), v.cpu().numpy() / out.item(), rtol=1e-15, atol=0)


def test_bench_loop_is_the_drivers_power_iteration(orc, g):
    rows, cols = int(g["rows"]), int(g["cols"])
    if rows != cols or len(g["csr1_col"]) == 0:
        pytest.skip("power loop needs a square non-empty matrix (driver_spmv.c:176)")
    m = Matrix.sell(rows, cols, g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"], S.F64, 64, 1024)
    iters, seconds, gflops, x_last = m.bench(g["x"], min_iters=10, min_seconds=0.0)
    assert iters == 10 and seconds > 0 and gflops > 0
    # replay driver_spmv.c:295-302 on the oracle
    x = g["x"].copy()
    for _ in range(iters):
        y = orc.mult_csr_base(g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"], x)
        nrm = orc.lib().orc_vec_norm(rows, y)
        if nrm == 0:
            pytest.skip("iterate vanished")
        x = y / nrm
    assert rel(orc, x_last, x) <= 1e-12


def test_coo_to_csr_in_hbm_equals_the_host_conversion(g, monkeypatch):
    """create_csr (csr.c:160-203) done in HBM for large inputs: same arrays, same in-place sort of the
    caller's COO, same products as the host conversion."""
    rows, cols = int(g["rows"]), int(g["cols"])
    monkeypatch.setenv("SPMV_HIP_DEVICE_CSR_MIN", "1")
    coo = coo_of(g, "in")
    m = Matrix.csr(rows, cols, coo, 1, S.F64, S.CSR_SCALAR)
    assert bits_equal(coo["i"], g["coo_sorted_i"]) and bits_equal(coo["j"], g["coo_sorted_j"])
    assert bits_equal(coo["a"], g["coo_sorted_a"])
    assert bits_equal(m.download(S.ARR_ROW_PTR), g["csr1_row_ptr"])
    assert bits_equal(m.download(S.ARR_COL), g["csr1_col"])
    assert bits_equal(m.download(S.ARR_VAL), g["csr1_val"])
    assert bits_equal(m.mult(g["x"]), g["y_csr_base"])
    m.free()
    m = Matrix.csr(rows, cols, coo_of(g, "in"), 1, S.F32, S.CSR_SCALAR)
    assert bits_equal(m.download(S.ARR_VAL), g["csr1_val"].astype(np.float32))
    m.free()
    for mode in ("csr_merge", "csr_bal", "ell", "auto"):
        m = S.create(mode, rows, cols, coo_of(g, "in"))
        assert orc_rel(m.mult(g["x"]), g["y_csr_base"]) <= TOL_REASSOC, mode
        m.free()
    if len(g["csr1_col"]):
        bad = coo_of(g, "in")
        bad["j"][0] = cols
        with pytest.raises(S.SpmvHipError, match="column index"):
            Matrix.csr(rows, cols, bad, 1, S.F64, S.CSR_SCALAR)


def orc_rel(y, ref):
    import oracle
    return oracle.rel_l2(np.asarray(y, np.float64), np.asarray(ref, np.float64))


def test_large_coo_is_sorted_in_hbm_like_the_host_sort(orc, monkeypatch):
    """sort_mtx order (mtx.c:9-34) from the device radix sort: same array as the host sort, stable for duplicates."""
    from spmv_amd.matrix import sort_coo
    rng = np.random.default_rng(8)
    n = 300000
    i = rng.integers(0, 5000, n).astype(np.int32)
    j = rng.integers(0, 5000, n).astype(np.int32)
    coo = S.make_coo(i, j, np.arange(n, dtype=np.float64))        # values tag the original position
    coo[:1000]["j"] = coo[1000:2000]["j"]; coo[:1000]["i"] = coo[1000:2000]["i"]   # duplicates
    order = np.lexsort((np.arange(n), coo["j"], coo["i"]))        # (i, j), ties by position = stable
    want = coo[order]
    host = coo.copy()
    monkeypatch.setenv("SPMV_HIP_DEVICE_SORT_MIN", str(10 ** 9))
    sort_coo(host)
    dev = coo.copy()
    monkeypatch.setenv("SPMV_HIP_DEVICE_SORT_MIN", "1000")
    sort_coo(dev)
    assert host.tobytes() == want.tobytes() and dev.tobytes() == want.tobytes()
    uniq = coo[np.unique(coo["i"].astype(np.int64) * 5000 + coo["j"], return_index=True)[1]]
    uniq = uniq[rng.permutation(len(uniq))]
    mine = uniq.copy()
    sort_coo(mine)
    assert mine.tobytes() == orc.sort_coo(uniq).tobytes()          # the reference's qsort order on distinct keys


def test_create_free_cycles_return_all_device_memory():
    import torch
    rows, cols, rp, cj, va = G.stencil27(24)
    coo = G.csr_to_coo_numpy(rows, rp, cj, va)
    rp, cj, va = rp.numpy(), cj.numpy(), va.numpy()
    x = np.ones(cols)

    def cycle():
        for mode in list(S.MODES) + ["auto", "sell_dfc"]:
            m = S.create(mode, rows, cols, coo.copy())
            m.analyze(1)
            m.mult(x)
            m.free()
        with pytest.raises(S.SpmvHipError):                        # failed builds release what they took
            Matrix.sell(rows, cols, rp, cj, va, S.F64, 64, 100)
    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(5):
        cycle()
    torch.cuda.synchronize()
    assert abs(torch.cuda.mem_get_info()[0] - free0) <= (8 << 20)


def test_errors_are_reported_not_fatal():
    coo = S.make_coo([0, 1], [0, 5], [1.0, 2.0])
    with pytest.raises(S.SpmvHipError, match="column index"):
        Matrix.csr(2, 2, coo)
    rp = np.array([0, 1, 2], np.int32); cj = np.array([0, 1], np.int32); va = np.ones(2)
    with pytest.raises(S.SpmvHipError, match="multiple of C"):
        Matrix.sell(2 * 100, 200, np.arange(201, dtype=np.int32), np.arange(200, dtype=np.int32), np.ones(200),
                    S.F64, 64, 100)
    with pytest.raises(S.SpmvHipError, match="power of two"):
        Matrix.sellp(2, 2, rp, cj, va, S.F64, 3)
    with pytest.raises(S.SpmvHipError, match="tile size"):
        Matrix.csr_merge(2, 2, S.make_coo([0, 1], [0, 1], [1.0, 2.0]), S.F64, 1000)


def test_csr_scalar_pattern_blocks_on_grids_and_bands(orc):
    """Blocks of 64 rows that repeat one pattern (grid interiors, bands) take the index-free path,
    blocks that do not (boundaries, the ragged last block) the ordinary one, in one launch; hot_columns
    relabels the columns and must fall back to reading every index."""
    for gen in (lambda: G.laplacian2d(200), lambda: G.stencil27(20, 21, 22), lambda: G.band(5000, 7, seed=2),
                lambda: G.band(64 * 40 + 13, 33, seed=3)):
        rows, cols, rp, cj, va = gen()
        rp, cj, va = rp.numpy().astype(np.int32), cj.numpy().astype(np.int32), va.numpy().astype(np.float64)
        x = np.random.default_rng(1).standard_normal(cols)
        want = orc.mult_csr_base(rp, cj, va, x)
        coo = S.make_coo(np.repeat(np.arange(rows, dtype=np.int32), np.diff(rp)), cj, va)
        for make in (Matrix.csr_merge, Matrix.csr_bal):                         # the tile kernels' pattern path
            for items in (0, 2048):
                mt = make(rows, cols, coo.copy(), S.F64, items)
                mt.set_option("variant", 1)
                y1 = mt.mult(x)
                assert orc.rel_l2(y1, want) <= TOL_REASSOC
                mt.set_option("variant", 2)
                assert bits_equal(mt.mult(x), y1)
                mt.set_option("hot_columns", 1)
                assert bits_equal(mt.mult(x), y1)
                mt.free()
        mv = Matrix.csr(rows, cols, coo.copy(), 1, S.F64, S.CSR_VECTOR)        # the vector kernel's pattern path
        mv.set_option("variant", 6)
        y6 = mv.mult(x)
        assert orc.rel_l2(y6, want) <= TOL_REASSOC
        for v in (7, 8):
            mv.set_option("variant", v)
            assert bits_equal(mv.mult(x), y6), v
        mv.set_option("hot_columns", 1)
        assert bits_equal(mv.mult(x), y6)
        mv.free()
        m = Matrix.csr(rows, cols, coo, 1, S.F64, S.CSR_SCALAR)
        for v in SCALAR_PATTERN_VARIANTS:
            m.set_option("variant", v)
            assert bits_equal(m.mult(x), want), v
        m.set_option("hot_columns", 1)
        assert bits_equal(m.mult(x), want)
        m.set_option("hot_columns", 0)
        assert bits_equal(m.mult(x), want)
        m.free()
        # fp32: each index-skipping variant against its index-reading twin, and the scalar one against the oracle
        x32 = x.astype(np.float32)
        want32 = orc.mult_csr_base_f32(rp, cj, va.astype(np.float32), x32)
        xin = x32.astype(np.float64)
        m = Matrix.csr(rows, cols, coo.copy(), 1, S.F32, S.CSR_SCALAR)
        for v in (16, 22):
            m.set_option("variant", v)
            assert bits_equal(m.mult(xin).astype(np.float32), want32), v
        m.free()
        for build, twin, skipping in ((lambda: Matrix.csr(rows, cols, coo.copy(), 1, S.F32, S.CSR_VECTOR), 6, 8),
                                      (lambda: Matrix.csr_merge(rows, cols, coo.copy(), S.F32), 1, 2),
                                      (lambda: Matrix.csr_bal(rows, cols, coo.copy(), S.F32), 1, 2)):
            m = build()
            m.set_option("variant", twin)
            y_twin = m.mult(xin)
            assert orc.rel_l2(y_twin, want32.astype(np.float64)) <= TOL_F32_VS_F32
            m.set_option("variant", skipping)
            assert bits_equal(m.mult(xin), y_twin), skipping
            m.free()


def test_reassociating_kernels_meet_the_componentwise_bound(orc, g):
    """|y - y_csr_base|_i <= 1e-12 (|A||x|)_i for every kernel that sums a row in another order than
    mult_csr_base (reference csr.c:4-13): vector CSR (all group sizes and the LDS-staged variants),
    merge-path and balanced tiles, the 16-byte-load sliced variants.  The rel-L2 bar alone would let a
    single badly summed small entry hide behind large ones."""
    rows, cols = int(g["rows"]), int(g["cols"])
    csr = (g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"])
    bound = orc.abs_product(*csr, g["x"])
    want = g["y_csr_base"]

    def check(m, what):
        y = m.mult(g["x"])
        assert (np.abs(y - want) <= 1e-12 * bound).all(), (g["name"], what)

    for lanes in (2, 8, 64):
        m = Matrix.csr(rows, cols, coo_of(g), 1, S.F64, S.CSR_VECTOR, lanes)
        for v in (0, 3, 5, 6):
            m.set_option("variant", v)
            check(m, ("csr_vector", lanes, v))
        m.free()
    for make, name in ((Matrix.csr_merge, "csr_merge"), (Matrix.csr_bal, "csr_bal")):
        for items in (256, 2048):
            m = make(rows, cols, coo_of(g), S.F64, items)
            for v in (0, 1):
                m.set_option("variant", v)
                check(m, (name, items, v))
            m.free()
    for m, name in ((Matrix.sell(rows, cols, *csr, S.F64, 64, 1024), "sell"), (Matrix.sellp(rows, cols, *csr, S.F64, 64), "sellp")):
        for v in (1, 2, 3, 4, 9, 12):
            m.set_option("variant", v)
            check(m, (name, v))
        m.free()


def test_block_map_leaves_the_bits_alone(g):
    """Option block_map only changes which workgroup takes which piece of the matrix: for every format, y under
    block_map 1 (consecutive) and under runs of 2, 5, 64 workgroups per XCD equals y under block_map 0 bit for
    bit (kernel variants of every family; grids smaller and larger than a round of 8 runs)."""
    rows, cols = int(g["rows"]), int(g["cols"])
    csr = (g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"])
    for dtype in (S.F64, S.F32):
        xin = g["x"].astype(np.float32).astype(np.float64) if dtype == S.F32 else g["x"]
        handles = [(Matrix.sell(rows, cols, *csr, dtype, 64, 1024), [0, 2, 12, 16, 22]),
                   (Matrix.sellp(rows, cols, *csr, dtype, 64), [0, 4, 9, 18, 21]),
                   (Matrix.sellp(rows, cols, *csr, dtype, 4), [0]),
                   (Matrix.sell_dfc(rows, cols, *csr, dtype, 64, 1024), [0, 3, 5, 8]),
                   (Matrix.csr(rows, cols, coo_of(g, "in"), 1, dtype, S.CSR_SCALAR), [0, 1, 3, 11]),
                   (Matrix.csr(rows, cols, coo_of(g, "in"), 1, dtype, S.CSR_VECTOR), [None]),
                   (Matrix.csr_merge(rows, cols, coo_of(g, "in"), dtype), [0, 1]),
                   (Matrix.csr_bal(rows, cols, coo_of(g, "in"), dtype), [0, 1])]
        for m, variants in handles:
            for v in variants:
                if v is not None:
                    m.set_option("variant", v)
                m.set_option("block_map", 0)
                y0 = m.mult(xin)
                for bmap in (1, 2, 5, 64):
                    m.set_option("block_map", bmap)
                    assert m.info.block_map == bmap
                    assert bits_equal(y0, m.mult(xin)), (m.info.format, v, dtype, bmap)
            with pytest.raises(S.SpmvHipError, match="block_map"):
                m.set_option("block_map", -1)
            m.free()


WIDE_EXACT_VARIANTS = [24, 25, 26, 27, 28, 29, 30, 31]


@pytest.mark.parametrize("fmt", ["sell", "sell64", "sellp", "ell"])
def test_sliced_wide_exact_variants_bitwise(orc, g, fmt):
    """16-byte loads with the rows handed back to single lanes by half-wave swaps (fp64 variants 24..27, and 28..31, which also leave the indices of diagonal slice columns unread): every row is
    summed by one lane in ascending column order, so y is bit-identical to the reference's y_csr_base on every fixture
    (odd and even slice widths, ragged last slices, empty rows); an fp32 handle refuses them."""
    rows, cols = int(g["rows"]), int(g["cols"])
    csr = (g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"])

    def make(dtype):
        if fmt == "sell":
            return Matrix.sell(rows, cols, *csr, dtype, 64, 1024)
        if fmt == "sell64":
            return Matrix.sell(rows, cols, *csr, dtype, 64, 64)
        if fmt == "sellp":
            return Matrix.sellp(rows, cols, *csr, dtype, 64)
        return Matrix.ell(rows, cols, coo_of(g, "in"), dtype, 64)

    m = make(S.F64)
    for v in WIDE_EXACT_VARIANTS:
        m.set_option("variant", v)
        for bmap in (0, 2):
            m.set_option("block_map", bmap)
            assert bits_equal(m.mult(g["x"]), g["y_csr_base"]), (fmt, v, bmap)
    m.set_option("block_map", 0)
    m.set_option("exact_order", 1)
    chosen = m.analyze(2)
    assert chosen == 0 or chosen >= 16
    assert bits_equal(m.mult(g["x"]), g["y_csr_base"]), chosen
    m.free()
    m = make(S.F32)
    if m.info.stored > 0:
        with pytest.raises(S.SpmvHipError, match="fp64"):
            m.set_option("variant", 24)
            m.mult(g["x"])
    m.free()
