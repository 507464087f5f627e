"""The C driver build/spmv_hip: command line, CSV schema and OK/ERR check of the
reference driver (reference src/driver_spmv.c:112-347) on top of libspmv_hip."""
import os
import struct
import subprocess

import numpy as np
import pytest

from spmv_amd import generators as G
from tests.util import ROOT

DRIVER = os.path.join(ROOT, "spmv_amd", "csrc", "build", "spmv_hip")
REF_HEADER = "Format;File;Rows;Columns;NNZ;GFlops;Time(s)"          # driver_spmv.c:139


def test_usage_and_unknown_mode_exit_like_reference(tmp_path):
    assert os.path.exists(DRIVER), "build the driver with make -C spmv_amd/csrc"
    p = subprocess.run([DRIVER], capture_output=True, text=True)
    assert p.returncode == 255 and "Usage" in p.stdout                # exit(-1), driver_spmv.c:126-129
    p = subprocess.run([DRIVER, "sell", str(tmp_path / "nolist.txt"), str(tmp_path / "o.csv")], capture_output=True, text=True)
    assert p.returncode == 255                                        # list missing: exit(-1), :134-137


def _write_bin(path, rows, cols, coo):
    """Layout of the reference's save_bin (mtx.c:207-247)."""
    with open(path, "wb") as f:
        f.write(struct.pack("iiq", rows, cols, len(coo)))
        f.write(coo.tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["csr_base", "csr_vec", "csr_merge", "csr_bal", "sellp_vec", "ell", "sell", "sell:f32",
                                  "sell:1024", "sell:64:f32"])
def test_driver_runs_every_mode(tmp_path, mode):
    mats = []
    for name, gen in (("lap2d_40", G.laplacian2d(40)), ("stencil_9", G.stencil27(9))):
        rows, cols, rp, cj, va = gen
        coo = G.csr_to_coo_numpy(rows, rp, cj, va)
        path = str(tmp_path / f"{name}.mtx")
        G.write_mtx(path, rows, cols, coo)
        mats.append((path, rows, len(coo)))
    rows, cols, rp, cj, va = G.band(300, 4)
    coo = G.csr_to_coo_numpy(rows, rp, cj, va)
    bpath = str(tmp_path / "band_300.bin")
    _write_bin(bpath, rows, cols, coo)
    mats.append((bpath, rows, len(coo)))
    lst = str(tmp_path / "list.txt")
    open(lst, "w").write("\n".join(m[0] for m in mats) + "\n")
    out = str(tmp_path / "out.csv")
    env = dict(os.environ, SPMV_MIN_SECONDS="0.05")      # the reference's 2 s rule, shortened for the test
    p = subprocess.run([DRIVER, mode, lst, out], capture_output=True, text=True, timeout=110, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = open(out).read().strip().split("\n")
    assert lines[0].startswith(REF_HEADER)
    assert len(lines) == 1 + len(mats)
    for line, (path, rows, nnz) in zip(lines[1:], mats):
        f = line.split(";")
        assert f[0] == mode and f[1] == os.path.basename(path)
        assert int(f[2]) == rows and int(f[3]) == rows and int(f[4]) == nnz
        assert float(f[5]) > 0 and float(f[6]) > 0
    verdicts = [l.split()[-1] for l in p.stdout.splitlines() if l.strip() and l.split()[-1] in ("OK", "ERR", "---")]
    assert verdicts == ["OK"] * len(mats), p.stdout          # driver_spmv.c:313-314


@pytest.mark.gpu
def test_driver_rejects_non_square_and_unknown_mode(tmp_path):
    from spmv_amd.matrix import make_coo
    coo = make_coo([0, 1], [0, 2], [1.0, 2.0])
    path = str(tmp_path / "rect.mtx")
    G.write_mtx(path, 2, 3, coo)
    lst = str(tmp_path / "list.txt")
    open(lst, "w").write(path + "\n")
    p = subprocess.run([DRIVER, "csr_base", lst, str(tmp_path / "o.csv")], capture_output=True, text=True, timeout=60)
    assert p.returncode == 1 and "Matrix is not square" in p.stdout        # driver_spmv.c:176-179
    rows, cols, rp, cj, va = G.laplacian2d(5)
    sq = str(tmp_path / "sq.mtx")
    G.write_mtx(sq, rows, cols, G.csr_to_coo_numpy(rows, rp, cj, va))
    open(lst, "w").write(sq + "\n")
    p = subprocess.run([DRIVER, "nonsense", lst, str(tmp_path / "o.csv")], capture_output=True, text=True, timeout=60)
    assert p.returncode == 255 and "Unknown execution mode" in p.stdout    # driver_spmv.c:258-260


def _small_list(tmp_path):
    mats = []
    for name, gen in (("stencil_12", G.stencil27(12, 13, 30)), ("lap2d_70", G.laplacian2d(70)), ("powerlaw", G.power_law_rows(3000))):
        rows, cols, rp, cj, va = gen
        coo = G.csr_to_coo_numpy(rows, rp, cj, va)
        path = str(tmp_path / f"{name}.bin")
        _write_bin(path, rows, cols, coo)
        mats.append((path, rows, len(coo)))
    lst = str(tmp_path / "list.txt")
    open(lst, "w").write("\n".join(m[0] for m in mats) + "\n")
    return lst, mats


@pytest.mark.gpu
def test_driver_sigma_reaches_the_library(tmp_path):
    """MODE suffix and SPMV_SIGMA select the sorting window (reference globals, sell.c:4-5); a sigma that
    is no multiple of C is the library's error, reported like the reference's helpers do (exit 1)."""
    lst, mats = _small_list(tmp_path)
    env = dict(os.environ, SPMV_MIN_SECONDS="0.02")
    for mode, extra in (("sell:1024", {}), ("sell", {"SPMV_SIGMA": "128"}), ("sell", {})):
        p = subprocess.run([DRIVER, mode, lst, str(tmp_path / "o.csv")], capture_output=True, text=True, timeout=110,
                           env=dict(env, **extra))
        assert p.returncode == 0 and p.stdout.count(" OK") == len(mats), p.stdout + p.stderr
    p = subprocess.run([DRIVER, "sell:100", lst, str(tmp_path / "o.csv")], capture_output=True, text=True, timeout=60, env=env)
    assert p.returncode == 1 and "multiple of C" in p.stderr
    p = subprocess.run([DRIVER, "csr_base:1024", lst, str(tmp_path / "o.csv")], capture_output=True, text=True, timeout=60, env=env)
    assert p.returncode == 255 and "Unknown execution mode" in p.stdout


@pytest.mark.gpu
def test_driver_skips_what_does_not_fit_in_device_memory(tmp_path):
    """A matrix whose estimate (2*16*nnz + 8*(rows+cols) bytes, driver_spmv.c:332-337) exceeds the free
    device memory gets the reference's zero row and the sweep goes on with the next file."""
    from spmv_amd import _lib
    import ctypes
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert _lib.lib().spmv_hip_mem_info(ctypes.byref(free), ctypes.byref(total)) == 0
    assert 0 < free.value <= total.value
    nnz = int(total.value) // 16                    # estimate = 2 * total: does not fit whatever the host has
    big = str(tmp_path / "too_big.mtx")
    open(big, "w").write(f"%%MatrixMarket matrix coordinate real general\n2000000000 2000000000 {nnz}\n")
    rows, cols, rp, cj, va = G.laplacian2d(30)
    small = str(tmp_path / "lap2d_30.mtx")
    G.write_mtx(small, rows, cols, G.csr_to_coo_numpy(rows, rp, cj, va))
    lst = str(tmp_path / "list.txt")
    open(lst, "w").write(big + "\n" + small + "\n")
    out = str(tmp_path / "o.csv")
    p = subprocess.run([DRIVER, "csr_base", lst, out], capture_output=True, text=True, timeout=110,
                       env=dict(os.environ, SPMV_MIN_SECONDS="0.02"))
    assert p.returncode == 0, p.stdout + p.stderr
    verdicts = [l.split()[-1] for l in p.stdout.splitlines() if l.strip() and l.split()[-1] in ("OK", "ERR", "---")]
    assert verdicts == ["---", "OK"], p.stdout
    lines = open(out).read().strip().split("\n")
    assert len(lines) == 3 and float(lines[1].split(";")[5]) == 0.0 and float(lines[2].split(";")[5]) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["sell:1024", "csr_base", "csr_merge", "sellp_vec", "csr_bal:f32"])
@pytest.mark.parametrize("ngpus", [2, 3])
def test_driver_row_blocks_over_several_gpus(tmp_path, mode, ngpus):
    """SPMV_NGPUS cuts every matrix into row blocks (here all on device 0): same OK verdict against the
    host COO product, NGPUs column in the CSV."""
    lst, mats = _small_list(tmp_path)
    out = str(tmp_path / "out.csv")
    env = dict(os.environ, SPMV_MIN_SECONDS="0.02", SPMV_NGPUS=str(ngpus), SPMV_DEVICES=",".join(["0"] * ngpus))
    p = subprocess.run([DRIVER, mode, lst, out], capture_output=True, text=True, timeout=110, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = open(out).read().strip().split("\n")
    assert lines[0].startswith(REF_HEADER) and lines[0].endswith(";NGPUs")
    assert len(lines) == 1 + len(mats) and all(l.split(";")[-1] == str(ngpus) for l in lines[1:])
    assert p.stdout.count(" OK") == len(mats), p.stdout


def test_default_matrix_list_is_generated_and_loads(tmp_path):
    """matrix_lists/synthetic_5.txt is committed; its matrices come from tools/make_matrix_list.py and
    load through the library's .bin reader (reference layout, mtx.c:139-247)."""
    import spmv_amd as S
    lst = os.path.join(ROOT, "matrix_lists", "synthetic_5.txt")
    assert os.path.exists(lst)
    p = subprocess.run(["python3", os.path.join(ROOT, "tools", "make_matrix_list.py"), "synthetic_5"], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    paths = open(lst).read().split()
    assert len(paths) == 5
    for rel in paths:
        rows, cols, coo = S.load_bin(os.path.join(ROOT, rel))
        assert rows == cols and len(coo) > 0 and int(coo["i"].max()) < rows and int(coo["j"].max()) < cols
    assert len(open(os.path.join(ROOT, "matrix_lists", "standins_160.txt")).read().split()) == 159


@pytest.mark.gpu
def test_run_script_works_out_of_the_box(tmp_path):
    """./spmv_hip.run with its defaults (reference spmv.run:5-37): list, MODE sell, output/<count>.csv."""
    out = str(tmp_path / "out")
    env = dict(os.environ, OUT=out, SPMV_MIN_SECONDS="0.02")
    for k in ("LIST", "MODE"):
        env.pop(k, None)
    for count in (1, 2):                     # `ls -l | wc -l` counts the "total" line too: the first file is 1.csv (spmv.run:35)
        p = subprocess.run([os.path.join(ROOT, "spmv_hip.run")], capture_output=True, text=True, timeout=200, env=env)
        assert p.returncode == 0, p.stdout + p.stderr
        lines = open(os.path.join(out, f"{count}.csv")).read().strip().split("\n")
        assert lines[0].startswith(REF_HEADER) and len(lines) == 6
        assert p.stdout.count(" OK") == 5, p.stdout
    env["LIST"] = "matrix_lists/absent.txt"
    p = subprocess.run([os.path.join(ROOT, "spmv_hip.run")], capture_output=True, text=True, timeout=60, env=env)
    assert p.returncode != 0 and "not found" in p.stderr
