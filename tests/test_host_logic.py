"""CPU-only checks of the host side: the C-ABI library loads without a GPU and
exports every symbol include/spmv_hip.h declares, the MatrixMarket loader and
COO sort follow the reference's rules (checked against the oracle's
restatement of reference src/mtx.c), the MODE table mirrors the reference
driver, and the synthetic generators have the documented shapes.
No compute call is made on the library here.
"""
import ctypes
import os
import re

import numpy as np
import pytest

import spmv_amd as S
from spmv_amd import generators as G
from spmv_amd import _lib
from spmv_amd.matrix import COO_DTYPE, load_mtx, make_coo, sort_coo
from tests.util import ROOT, bits_equal


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "spmv_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(spmv_hip_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 20
    lib = ctypes.CDLL(S.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in spmv_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes signature table and header disagree"


def test_info_struct_matches_header_field_order():
    header = open(os.path.join(ROOT, "include", "spmv_hip.h")).read()
    body = header[header.index("typedef struct spmv_hip_info {"):header.index("} spmv_hip_info;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split("{", 1)[1].split(";"):
        decl = decl.strip()
        if decl:
            names = decl.split(None, 1)[1]
            fields += [n.strip() for n in names.split(",")]
    assert fields == [f[0] for f in _lib.Info._fields_]


def test_no_gpu_means_loud_failure_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    coo = make_coo([0], [0], [1.0])
    with pytest.raises(S.SpmvHipError):
        S.Matrix.csr(1, 1, coo)


MTX_CASES = {
    "general_real": ("real", "general", [(1, 1, 2.5), (3, 2, -1.0), (2, 3, 0.0), (1, 3, 1e-3), (3, 3, 7.0)]),
    "symmetric": ("real", "symmetric", [(1, 1, 4.0), (2, 1, -1.0), (3, 2, -1.5), (3, 3, 2.0)]),
    "skew": ("real", "skew-symmetric", [(2, 1, 3.0), (3, 1, -2.0)]),
    "integer": ("integer", "general", [(1, 2, 3), (2, 2, -4), (3, 1, 5)]),
    "pattern": ("pattern", "general", [(1, 1), (2, 3), (3, 2)]),
    "pattern_sym": ("pattern", "symmetric", [(1, 1), (3, 1), (3, 2)]),
}


def _write(path, field, symmetry, entries, n=3, comments=True):
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate {field} {symmetry}\n")
        if comments:
            f.write("% a comment\n%another\n")
        f.write(f"{n} {n} {len(entries)}\n")
        for e in entries:
            f.write(" ".join(str(v) for v in e) + "\n")


@pytest.mark.parametrize("name", sorted(MTX_CASES))
def test_load_mtx_follows_reference_rules(orc, tmp_path, name):
    field, symmetry, entries = MTX_CASES[name]
    path = str(tmp_path / f"{name}.mtx")
    _write(path, field, symmetry, entries)
    rows, cols, coo = load_mtx(path)
    orows, ocols, ocoo = orc.load_mtx(path)
    assert (rows, cols) == (orows, ocols) == (3, 3)
    assert bits_equal(coo["i"], ocoo["i"]) and bits_equal(coo["j"], ocoo["j"]) and bits_equal(coo["a"], ocoo["a"])
    if name == "general_real":
        assert len(coo) == 4                       # the explicit zero is dropped (mtx.c:118)
    if name == "skew":
        assert coo["a"].tolist() == [3.0, 3.0, -2.0, -2.0]     # mirrored WITHOUT sign change (mtx.c:127-132)
    if field == "pattern":
        assert set(coo["a"].tolist()) == {1.0}


def test_load_mtx_roundtrip_of_generated_matrix(orc, tmp_path):
    rows, cols, rp, cj, va = G.stencil27(5, 4, 3)
    coo = G.csr_to_coo_numpy(rows, rp, cj, va)
    path = str(tmp_path / "stencil.mtx")
    G.write_mtx(path, rows, cols, coo)
    r2, c2, coo2 = load_mtx(path)
    assert (r2, c2) == (rows, cols)
    assert bits_equal(coo2["i"], coo["i"]) and bits_equal(coo2["j"], coo["j"]) and bits_equal(coo2["a"], coo["a"])


@pytest.mark.parametrize("content,msg", [
    ("%%NotMatrixMarket matrix coordinate real general\n1 1 1\n1 1 1.0\n", "Missing header"),
    ("%%MatrixMarket matrix array real general\n1 1\n1.0\n", "not supported"),
    ("%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1.0 0.0\n", "complex"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 1\n3 1 1.0\n", "out of range"),
    ("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n", "Error reading matrix element"),
])
def test_load_mtx_errors_where_reference_exits(orc, tmp_path, content, msg):
    path = str(tmp_path / "bad.mtx")
    open(path, "w").write(content)
    with pytest.raises(S.SpmvHipError, match=msg):
        load_mtx(path)
    with pytest.raises(ValueError):                  # the oracle restatement rejects it too
        orc.load_mtx(path)
    with pytest.raises(S.SpmvHipError, match="Error opening file"):
        load_mtx(str(tmp_path / "missing.mtx"))


def _big_mtx(path, n=3000, per_row=40, symmetric=False, seed=11):
    """~3 MB of entry lines, enough for the loader to split the file over threads (one part per MiB)."""
    rng = np.random.default_rng(seed)
    i = np.repeat(np.arange(1, n + 1), per_row)
    j = rng.integers(1, n + 1, size=len(i))
    if symmetric:
        i, j = np.maximum(i, j), np.minimum(i, j)
    a = rng.standard_normal(len(i))
    a[::97] = 0.0                                                 # explicit zeros are dropped (mtx.c:118)
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate real {'symmetric' if symmetric else 'general'}\n% big\n")
        f.write(f"{n} {n} {len(i)}\n")
        f.write("".join(f"{r} {c} {v!r}  \n" for r, c, v in zip(i.tolist(), j.tolist(), a.tolist())))
    return len(i)


@pytest.mark.parametrize("symmetric", [False, True])
def test_load_mtx_threads_agree_with_sequential_and_oracle(orc, tmp_path, monkeypatch, symmetric):
    path = str(tmp_path / "big.mtx")
    _big_mtx(path, symmetric=symmetric)
    assert os.path.getsize(path) > (2 << 20)
    got = {}
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("SPMV_HIP_IO_THREADS", threads)
        got[threads] = load_mtx(path)
    orows, ocols, ocoo = orc.load_mtx(path)
    for rows, cols, coo in got.values():
        assert (rows, cols) == (orows, ocols)
        assert coo.tobytes() == ocoo.tobytes()                    # same entries in file order, mirrors after their source


def test_load_mtx_irregular_layouts_fall_back_to_token_parsing(orc, tmp_path, monkeypatch):
    monkeypatch.setenv("SPMV_HIP_IO_THREADS", "4")
    # entries split over lines, several per line, and trailing lines beyond nnz (fscanf-style readers accept all three)
    text = "%%MatrixMarket matrix coordinate real general\n3 3 3\n1 1\n 2.0 2 2 3.0\n3\n3 4.0\n1 2 9.0\n"
    path = str(tmp_path / "odd.mtx")
    open(path, "w").write(text)
    rows, cols, coo = load_mtx(path)
    orows, ocols, ocoo = orc.load_mtx(path)
    assert (rows, cols) == (orows, ocols) == (3, 3)
    assert coo.tobytes() == ocoo.tobytes()
    assert coo["a"].tolist() == [2.0, 3.0, 4.0]                   # the line after the third entry is ignored


def test_bin_roundtrip_and_errors(tmp_path):
    rows, cols, rp, cj, va = G.stencil27(4, 3, 5)
    coo = G.csr_to_coo_numpy(rows, rp, cj, va)
    path = str(tmp_path / "m.bin")
    S.save_bin(path, rows, cols, coo)
    assert os.path.getsize(path) == 4 + 4 + 8 + 16 * len(coo)    # mtx.c:207-247
    r2, c2, coo2 = S.load_bin(path)
    assert (r2, c2) == (rows, cols) and coo2.tobytes() == coo.tobytes()
    with open(path, "r+b") as f:                                  # truncated payload
        f.truncate(16 + 16 * (len(coo) - 1) + 3)
    with pytest.raises(S.SpmvHipError, match="Error reading file"):
        S.load_bin(path)
    bad = coo.copy()
    bad["j"][0] = cols
    S.save_bin(path, rows, cols, bad)
    with pytest.raises(S.SpmvHipError, match="out of range"):
        S.load_bin(path)
    with pytest.raises(S.SpmvHipError, match="Error opening file"):
        S.load_bin(str(tmp_path / "missing.bin"))
    S.save_bin(path, 0, 0, coo[:0])
    assert S.load_bin(path)[0:2] == (0, 0)


def test_bin_files_interchange_with_compiled_reference(tmp_path):
    """save_bin / load_bin of the compiled reference (mtx.c:139-247) and of the library read each other's files."""
    import ctypes as C
    import oracle
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built")
    R = oracle.ref()
    rows, cols, rp, cj, va = G.rmat(8, 6, seed=5)
    coo = G.csr_to_coo_numpy(rows, rp, cj, va)
    ours, theirs = str(tmp_path / "ours.bin"), str(tmp_path / "theirs.bin")
    S.save_bin(ours, rows, cols, coo)
    R.ref_save_bin(theirs.encode(), rows, cols, len(coo), coo.copy())
    assert open(ours, "rb").read() == open(theirs, "rb").read()
    r2, c2, coo2 = S.load_bin(theirs)
    assert (r2, c2) == (rows, cols) and coo2.tobytes() == coo.tobytes()
    r, c, n, p = C.c_int(), C.c_int(), C.c_int(), C.c_void_p()
    R.ref_load_bin(ours.encode(), C.byref(r), C.byref(c), C.byref(n), C.byref(p))
    back = np.frombuffer((C.c_char * (16 * n.value)).from_address(p.value), dtype=COO_DTYPE, count=n.value).copy()
    R.ref_free(p)
    assert (r.value, c.value, n.value) == (rows, cols, len(coo)) and back.tobytes() == coo.tobytes()


def test_sort_coo_matches_reference_order(orc):
    rng = np.random.default_rng(4)
    n = 5000
    keys = rng.choice(300 * 300, size=n, replace=False)          # unique (i, j): order is fully determined
    coo = make_coo((keys // 300).astype(np.int32), (keys % 300).astype(np.int32), rng.standard_normal(n))
    want = orc.sort_coo(coo)
    sort_coo(coo)
    assert coo.tobytes() == want.tobytes()
    again = coo.copy()
    sort_coo(again)                                               # already sorted: untouched
    assert again.tobytes() == coo.tobytes()


def test_mode_table_mirrors_reference_driver():
    # MODE strings the reference driver accepts for the formats on this path (driver_spmv.c:186-261)
    for mode in ("coo", "csr_base", "csr_autovec", "csr_vec", "csr_highway", "csr_merge", "csr_bal", "sellp_vec",
                 "sellp_autovec", "sellp_highway", "ell", "sell"):
        assert mode in S.MODES
    with pytest.raises(ValueError, match="Unknown execution mode"):
        S.create("coo2", 1, 1, np.zeros(0, COO_DTYPE))


def test_generators_have_documented_shapes():
    rows, cols, rp, cj, va = G.laplacian2d(30)
    assert rows == cols == 900 and int(rp[-1]) == 5 * 900 - 4 * 30
    rows, cols, rp, cj, va = G.stencil27(6, 7, 8)
    assert rows == 6 * 7 * 8 and int(rp[-1]) == (3 * 6 - 2) * (3 * 7 - 2) * (3 * 8 - 2)
    assert float(va.sum()) == 26.0 * rows - (int(rp[-1]) - rows)
    # a row block of the global stencil equals the matching slice of the full one
    r0, r1 = 6 * 7 * 2, 6 * 7 * 5
    _, _, rp_b, cj_b, va_b = G.stencil27(6, 7, 8, row_begin=r0, row_end=r1)
    a, b = int(rp[r0]), int(rp[r1])
    assert np.array_equal((rp[r0:r1 + 1] - rp[r0]).numpy(), rp_b.numpy())
    assert np.array_equal(cj[a:b].numpy(), cj_b.numpy()) and np.array_equal(va[a:b].numpy(), va_b.numpy())
    rows, cols, rp, cj, va = G.rmat(8, 8)
    lens = np.diff(rp.numpy())
    assert rows == 256 and lens.max() > 8 * lens.mean() / 2 and (lens == 0).any()      # skewed, with empty rows
    keys = np.repeat(np.arange(rows), lens).astype(np.int64) * rows + cj.numpy()
    assert np.all(np.diff(keys) > 0)                                                   # sorted, no duplicates
    r2 = G.rmat(8, 8)
    assert np.array_equal(r2[3].numpy(), cj.numpy()) and np.array_equal(r2[4].numpy(), va.numpy())  # reproducible
