"""Stand-ins of the reference's sweep list (reference matrix_lists/matrix_list_160.txt:1-159) against the oracle.

Every family of spmv_amd.standins -- structured grids, block and scalar FEM, circuit, KKT, quantum chemistry,
random -- with the entries of the list the families are best known by, scaled down to a few million non-zeros.
Each is built through every MODE of the reference driver's table (driver_spmv.c:186-261, plus sell:1024, sell_dfc
and auto) in fp64 and fp32 from its host COO, tuned by the kernel trial, and multiplied; y is compared with the
oracle's mult_csr_base (csr.c:4-13) on the same CSR:
  lane-per-row formats (csr_base, sell, sellp, ell, sell_dfc): bit for bit, also after the trial (exact_order);
  reassociating formats (csr_vec, csr_merge, csr_bal, auto): rel-L2 <= 1e-13 and |y - y_ref|_i <= 1e-12 (|A||x|)_i;
  fp32: bit for bit against the oracle's float restatement (mult_csr_base_f32) for the lane-per-row formats,
        rel-L2 <= 2e-6 against it otherwise -- unpinned by the reference, which has no float path.
The same matrices then go through the C driver as .bin files (the reference's save_bin layout, mtx.c:207-247),
whose own check is the host COO loop with the 1e-10 bar (driver_spmv.c:274-289,313).
"""
import os
import subprocess

import numpy as np
import pytest
import torch

import spmv_amd as S
from spmv_amd import generators as G
from spmv_amd import standins as SI
from spmv_amd.matrix import create, save_bin
from tests.util import ROOT

pytestmark = pytest.mark.gpu

NAMES = ["G3_circuit", "atmosmodd", "audikw_1", "pwtk", "dielFilterV2real", "thermal2", "circuit5M", "Freescale1",
         "nlpkkt80", "stokes", "Ga41As41H72", "CO", "m133-b3", "mac_econ_fwd500"]
EXACT = ("coo", "csr_base", "sellp_vec", "ell", "sell", "sell:1024", "sell_dfc")
REASSOCIATING = ("csr_vec", "csr_merge", "csr_bal", "auto")
DRIVER = os.path.join(ROOT, "spmv_amd", "csrc", "build", "spmv_hip")


def test_the_selection_covers_every_family():
    assert len(NAMES) >= 12 and {SI.BY_NAME[n][1] for n in NAMES} == set(SI.FAMILIES)


def _standin(name, cap):
    nnz = SI.BY_NAME[name][3]
    rows, cols, rp, cj, va = SI.standin(name, min(1.0, cap / nnz), device="cuda:0")
    return rows, cols, rp.cpu(), cj.cpu(), va.cpu()


@pytest.mark.parametrize("name", NAMES)
def test_every_mode_of_a_standin_matches_the_oracle(orc, name):
    rows, cols, rp, cj, va = _standin(name, 6e6)
    rp_n, cj_n, va_n = rp.numpy(), cj.numpy(), va.numpy()
    assert len(cj_n) <= 2e7
    coo = G.csr_to_coo_numpy(rows, rp, cj, va)
    x = np.random.default_rng(5).random(cols)
    x /= np.linalg.norm(x)
    want = orc.mult_csr_base_rows_parallel(rp_n, cj_n, va_n, x)
    scale = orc.abs_product(rp_n, cj_n, va_n, x)
    x32 = x.astype(np.float32)
    want32 = orc.mult_csr_base_f32(rp_n, cj_n, va_n.astype(np.float32), x32)
    for mode in EXACT + REASSOCIATING:
        for dtype in (S.F64, S.F32):
            sigma = 1024 if mode == "sell:1024" else 0
            m = create(mode.split(":")[0], rows, cols, coo, dtype, sigma)
            if mode in EXACT:
                m.set_option("exact_order", 1)
            m.analyze(2)
            tag = f"{name} {mode} {'f64' if dtype == S.F64 else 'f32'} variant {m.info.variant}"
            if dtype == S.F64:
                y = m.mult(x)
                if mode in EXACT:
                    assert y.tobytes() == want.tobytes(), tag
                else:
                    assert orc.rel_l2(y, want) <= 1e-13, tag
                    assert (np.abs(y - want) <= 1e-12 * scale).all(), tag
            else:
                y = m.mult(x32.astype(np.float64)).astype(np.float32)
                if mode in EXACT:
                    assert y.tobytes() == want32.tobytes(), tag
                else:
                    assert orc.rel_l2(y.astype(np.float64), want32.astype(np.float64)) <= 2e-6, tag
            m.free()


@pytest.fixture(scope="module")
def bin_list(tmp_path_factory):
    d = tmp_path_factory.mktemp("standins_bin")
    paths = []
    for name in NAMES:
        rows, cols, rp, cj, va = _standin(name, 1e6)
        path = str(d / f"{name}.bin")
        save_bin(path, rows, cols, G.csr_to_coo_numpy(rows, rp, cj, va))
        paths.append(path)
    lst = str(d / "list.txt")
    with open(lst, "w") as f:
        f.write("\n".join(paths) + "\n")
    return lst, str(d)


@pytest.mark.parametrize("mode", ["csr_base", "csr_vec", "csr_merge", "csr_bal", "sellp_vec", "ell", "sell", "sell:1024",
                                  "sell_dfc", "auto"])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_standins_as_bin_files_through_the_c_driver(bin_list, mode, dtype):
    lst, d = bin_list
    out = os.path.join(d, f"{mode.replace(':', '_')}_{dtype}.csv")
    env = dict(os.environ, SPMV_MIN_SECONDS="0.0")
    p = subprocess.run([DRIVER, f"{mode}:{dtype}", lst, out], capture_output=True, text=True, timeout=170, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    verdicts = [ln.split()[-1] for ln in p.stdout.splitlines() if ln.strip() and ln.split()[-1] in ("OK", "ERR", "---")]
    assert verdicts == ["OK"] * len(NAMES), p.stdout
    assert len(open(out).read().strip().split("\n")) == 1 + len(NAMES)
