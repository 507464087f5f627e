"""Stand-ins of the reference's sweep list (matrix_lists/matrix_list_160.txt) and the sweep driver."""
import os

import numpy as np
import pytest

from spmv_amd import standins as SI

REF_LIST = "/root/reference/matrix_lists/matrix_list_160.txt"


def test_table_covers_the_list():
    names = SI.names()
    assert len(names) == 159 and len(set(names)) == 159           # matrix_list_160.txt has 159 lines
    assert all(fam in SI.FAMILIES and rows > 0 and nnz >= rows for _, fam, rows, nnz in SI.TABLE)
    if os.path.exists(REF_LIST):                                   # only where the reference tree is present
        with open(REF_LIST) as f:
            want = [os.path.basename(ln.strip())[:-len(".mtx")] for ln in f if ln.strip()]
        assert names == want


@pytest.mark.parametrize("family", SI.FAMILIES)
def test_generated_csr_is_well_formed_and_near_target(family):
    rows_t, nnz_t = 20000, 20000 * {"grid5": 5, "grid7": 7, "fem": 45, "qchem": 40}.get(family, 9)
    rows, cols, rp, cj, va = SI.generate(family, rows_t, nnz_t, seed=7)
    rp, cj, va = rp.numpy().astype(np.int64), cj.numpy().astype(np.int64), va.numpy()
    assert rows == cols and len(rp) == rows + 1 and rp[0] == 0 and rp[-1] == len(cj) == len(va)
    assert abs(rows - rows_t) <= 0.1 * rows_t
    assert 0.6 * nnz_t <= len(cj) <= 1.4 * nnz_t
    assert cj.min() >= 0 and cj.max() < cols and np.isfinite(va).all()
    key = np.repeat(np.arange(rows, dtype=np.int64), np.diff(rp)) * (1 << 32) + cj
    assert (np.diff(key) > 0).all()                                # (row, col) strictly ascending: sorted, no duplicates
    diag = np.zeros(rows, bool)
    diag[cj[cj == np.repeat(np.arange(rows), np.diff(rp))]] = True
    assert diag.all()                                              # every row holds its diagonal
    again = SI.generate(family, rows_t, nnz_t, seed=7)
    assert np.array_equal(again[3].numpy(), cj) and np.array_equal(again[4].numpy(), va)


def test_circuit_family_is_skewed():
    rows, _, rp, _, _ = SI.generate("circuit", 200000, 1200000, seed=3)
    lens = np.diff(rp.numpy().astype(np.int64))
    assert lens.max() > 50 * np.median(lens)


def test_shard_deals_every_matrix_once():
    from spmv_amd import sweep
    names = SI.names()
    for world in (1, 2, 8):
        parts = [sweep.shard(names, r, world) for r in range(world)]
        assert sorted(n for p in parts for n in p) == sorted(names)
        sizes = [sum(SI.BY_NAME[n][3] for n in p) for p in parts]
        assert max(sizes) <= 2.2 * (sum(sizes) / world)           # largest-first dealing keeps ranks comparable
    assert sweep.algorithmic_bytes(10, 10, 50, 8) == 50 * 12 + 10 * 12 + 10 * 8


@pytest.mark.gpu
def test_mini_sweep_on_gpu(tmp_path):
    from spmv_amd import sweep
    out = tmp_path / "sweep.csv"
    rc = sweep.main(["--out", str(out), "--scale", "0.02", "--only", "pwtk,circuit5M,ecology1,nlpkkt80,CO",
                     "--formats", "csr_base,csr_vec,csr_merge,csr_bal,ell,sellp_vec,sell", "--min-seconds", "0.0"])
    assert rc == 0
    lines = out.read_text().strip().split("\n")
    assert lines[0] == sweep.CSV_HEADER
    assert len(lines) == 1 + 5 * 7 * 2
    for ln in lines[1:]:
        f = ln.split(";")
        assert f[-1] == "OK", ln
        assert float(f[5]) > 0 and int(f[11]) >= 10
