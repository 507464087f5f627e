"""Pins the CPU oracle (oracle/spmv_oracle.c) against the fixtures that
tests/golden/make_golden.py wrote from the compiled reference.

Bar: bit-exact for every index array and layout; bit-exact for the products
whose summation order the oracle reproduces (csr_base, COO, merge path, SELL,
SELLP/ELL scalar loops); <= 1e-13 relative L2 for the reference kernels that
reassociate with SIMD lanes (csr_vec AVX2, csr_autovec / csr_bal `omp simd`,
sellp AVX2 is lane-per-row and stays exact).
"""
import numpy as np
import pytest

from tests.util import GOLDEN_DIR, bits_equal, coo_of, golden_names, load_golden
from tests import cases as CS

NAMES = golden_names()


@pytest.fixture(scope="module", params=NAMES)
def g(request):
    d = load_golden(request.param)
    d["name"] = request.param
    return d


def test_manifest_lists_every_fixture():
    import json, os, hashlib
    man = json.load(open(os.path.join(GOLDEN_DIR, "MANIFEST.json")))
    assert sorted(man) == NAMES
    for name, meta in man.items():
        blob = open(os.path.join(GOLDEN_DIR, name + ".npz"), "rb").read()
        assert hashlib.sha256(blob).hexdigest() == meta["sha256"]


def test_sort_coo(orc, g):
    got = orc.sort_coo(coo_of(g, "in"))
    assert bits_equal(got["i"], g["coo_sorted_i"]) and bits_equal(got["j"], g["coo_sorted_j"])
    assert bits_equal(got["a"], g["coo_sorted_a"])


def test_mult_coo(orc, g):
    y = orc.mult_coo(int(g["rows"]), coo_of(g), g["x"])
    assert bits_equal(y, g["y_coo"])


@pytest.mark.parametrize("align", [1, 4])
def test_csr_layout(orc, g, align):
    rp, cj, va = orc.csr_from_coo(int(g["rows"]), coo_of(g), align)
    assert bits_equal(rp, g[f"csr{align}_row_ptr"])
    assert bits_equal(cj, g[f"csr{align}_col"])
    assert bits_equal(va, g[f"csr{align}_val"])


def test_mult_csr_base_bitwise(orc, g):
    y = orc.mult_csr_base(g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"], g["x"])
    assert bits_equal(y, g["y_csr_base"])


def test_reference_simd_kernels_close_to_csr_base(orc, g):
    for key in ("y_csr_autovec", "y_csr_vec"):
        assert orc.rel_l2(g[key], g["y_csr_base"]) <= 1e-13, key


@pytest.mark.parametrize("nt", CS.MERGE_THREADS)
def test_merge_path(orc, g, nt):
    rp, cj, va = g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"]
    rows, nnz = int(g["rows"]), len(cj)
    items = rows + nnz
    per = (items + nt - 1) // nt
    for t in range(nt + 1):
        d = min(per * t, items)
        assert orc.merge_path_search(d, rp, nnz) == tuple(g[f"merge_coords_t{nt}"][t]), (t, d)
    y, rc, vc = orc.mult_csr_merge(rp, cj, va, g["x"], nt)
    assert bits_equal(rc, g[f"merge_row_carry_t{nt}"])
    assert bits_equal(vc, g[f"merge_val_carry_t{nt}"])
    assert bits_equal(y, g[f"y_csr_merge_t{nt}"])


@pytest.mark.parametrize("nt", CS.BAL_THREADS)
def test_csr_bal(orc, g, nt):
    key = f"y_csr_bal_t{nt}"
    if key not in g:
        pytest.skip("reference csr_bal is undefined for this matrix (empty rows / nnz < threads)")
    coo = coo_of(g)
    s, l, fr, lr, nr = orc.csr_bal_partition(coo, nt)
    part = g[f"bal_part_t{nt}"]
    assert bits_equal(nr, part[:, 0].copy()) and bits_equal(fr, part[:, 1].copy()) and bits_equal(lr, part[:, 2].copy())
    y = orc.mult_csr_bal(int(g["rows"]), coo, g["x"], nt)
    assert orc.rel_l2(y, g[key]) <= 1e-13          # reference sums with `omp simd`
    assert orc.rel_l2(y, g["y_csr_base"]) <= 1e-13


@pytest.mark.parametrize("sigma", CS.SELL_SIGMAS)
def test_sell_layout_and_product(orc, g, sigma):
    S = orc.sell_from_csr(g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"], 64, sigma)
    assert bits_equal(S["row_order"], g[f"sell_s{sigma}_row_order"])
    assert bits_equal(S["slice_widths"], g[f"sell_s{sigma}_slice_widths"])
    assert bits_equal(S["slice_ptr"], g[f"sell_s{sigma}_slice_ptr"])
    assert bits_equal(S["col"], g[f"sell_s{sigma}_col"])
    assert bits_equal(S["val"], g[f"sell_s{sigma}_val"])
    y = orc.mult_sell(S, g["x"])
    if f"y_sell_s{sigma}" in g:
        assert bits_equal(y, g[f"y_sell_s{sigma}"])
    # lane-per-row keeps the in-row order of csr_base
    assert bits_equal(y, g["y_csr_base"])


@pytest.mark.parametrize("sigma", CS.SELL_SIGMAS)
def test_sell_dfc_is_the_sell_layout_without_its_padding(orc, g, sigma):
    """csr_to_sellcs_dfc (sellcs_format.c:53-136) against the padded layout of the fixtures: same row order
    and widths; dropping the padding of every column of every slice gives exactly the DFC arrays."""
    D = orc.sell_dfc_from_csr(g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"], 64, sigma)
    assert bits_equal(D["row_order"], g[f"sell_s{sigma}_row_order"])
    assert bits_equal(D["slice_widths"], g[f"sell_s{sigma}_slice_widths"])
    sp, sc, sv = g[f"sell_s{sigma}_slice_ptr"], g[f"sell_s{sigma}_col"], g[f"sell_s{sigma}_val"]
    lens = D["rows_size_sorted"]
    cols, vals, active = [], [], []
    for s, w in enumerate(D["slice_widths"]):
        ls = lens[s * 64:(s + 1) * 64]
        for v in range(int(w)):
            a = int((ls > v).sum())
            active.append(a - 1)
            cols.append(sc[sp[s] + 64 * v: sp[s] + 64 * v + a]); vals.append(sv[sp[s] + 64 * v: sp[s] + 64 * v + a])
    assert bits_equal(D["vop_len"], np.array(active, np.uint8))
    assert bits_equal(D["col"], np.concatenate(cols) if cols else np.zeros(0, np.int32))
    assert bits_equal(D["val"], np.concatenate(vals) if vals else np.zeros(0))
    assert bits_equal(orc.mult_sell_dfc(D, g["x"]), g["y_csr_base"])


def test_sellp_and_ell_layout_and_product(orc, g):
    E = orc.sliced_ell_from_csr(g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"], 4)
    assert bits_equal(E["block_ptr"], g["sellp4_block_ptr"])
    assert bits_equal(E["j"], g["sellp4_j"]) and bits_equal(E["A"], g["sellp4_A"])
    assert bits_equal(E["block_ptr"], g["ell4_block_ptr"])
    assert bits_equal(E["j"].astype(np.int64), g["ell4_j"]) and bits_equal(E["A"], g["ell4_A"])
    y = orc.mult_sliced_ell(E, g["x"])
    for key in ("y_sellp_autovec", "y_sellp_vec", "y_ell"):
        assert bits_equal(y, g[key]), key
    assert bits_equal(y, g["y_csr_base"])


def test_rows_parallel_product_has_the_bits_of_csr_base(orc):
    """The OpenMP row-parallel restatement used by the full-size GPU tests is mult_csr_base bit for bit."""
    for name in golden_names():
        g = load_golden(name)
        y = orc.mult_csr_base_rows_parallel(g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"], g["x"])
        assert bits_equal(y, g["y_csr_base"]), name
        bound = orc.abs_product(g["csr1_row_ptr"], g["csr1_col"], g["csr1_val"], g["x"])
        assert (np.abs(g["y_csr_base"]) <= bound * (1 + 1e-15) + 1e-300).all(), name
