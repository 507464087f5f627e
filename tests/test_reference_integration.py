"""The drop-in built for real: the reference's own driver (reference src/driver_spmv.c:112-347) with the
"#ifdef HIP" binding of integration/reference_driver_hip.patch applied (the hunks INTEGRATION.md shows), compiled
with the reference's flags (reference Makefile:6,21,25) plus -DHIP -DSPMV_HIP_NO_MTX_STRUCT and linked with the
reference's objects and libspmv_hip.so by oracle/Makefile (target ref_driver_hip -> oracle/_ref/spmv_ref_hip).

On CPU (where /root/reference exists): the patch only adds lines, applies, compiles and links, and every
spmv_hip_* symbol the patched driver needs is exported by the library.  On the GPU box the binary built here runs
MODEs hip_sell / hip_csr_base / hip_csr_merge on a generated .mtx: the reference's loader, its COO check with the
1e-10 bar (driver_spmv.c:274-289,313) and its table, with the HIP engine behind the function pointer.
"""
import ctypes
import os
import re
import subprocess

import pytest

from spmv_amd import generators as G
from tests.util import ROOT

REF = "/root/reference"
PATCH = os.path.join(ROOT, "integration", "reference_driver_hip.patch")
BINARY = os.path.join(ROOT, "oracle", "_ref", "spmv_ref_hip")


def test_patch_only_adds_lines():
    body = [ln for ln in open(PATCH).read().split("\n") if ln and not ln.startswith(("---", "+++", "@@"))]
    assert body and all(ln.startswith("+") for ln in body)           # nothing of the reference's text is in the patch
    hunks = re.findall(r"^@@ -(\d+),0 \+\d+,\d+ @@", open(PATCH).read(), re.M)
    assert len(hunks) == 5                                           # include, MODE branches, trial, timed loop, free


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "driver_spmv.c")), reason="reference tree not present")
def test_patched_reference_driver_compiles_and_links():
    lib = os.path.join(ROOT, "spmv_amd", "libspmv_hip.so")
    assert os.path.exists(lib), "make -C spmv_amd/csrc first"
    p = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "ref_driver_hip"], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert os.path.exists(BINARY)
    undefined = subprocess.run(["nm", "-D", "--undefined-only", BINARY], capture_output=True, text=True, check=True).stdout
    needed = sorted(set(re.findall(r"\bspmv_hip_\w+", undefined)))
    assert {"spmv_hip_create_sell", "spmv_hip_create_csr_merge", "spmv_hip_mult", "spmv_hip_analyze", "spmv_hip_bench",
            "spmv_hip_free"} <= set(needed)
    handle = ctypes.CDLL(lib)
    for name in needed:
        assert hasattr(handle, name), f"{name} is not exported by libspmv_hip.so"
    ldd = subprocess.run(["ldd", BINARY], capture_output=True, text=True).stdout
    assert "libspmv_hip.so" in ldd and "not found" not in ldd
    # the reference's own symbols are still what the unpatched branches call
    defined = subprocess.run(["nm", "--defined-only", BINARY], capture_output=True, text=True, check=True).stdout
    assert "mult_csr_base" in defined and "create_sell" in defined


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BINARY), reason="oracle/_ref/spmv_ref_hip was not built (needs the reference tree)")
@pytest.mark.parametrize("mode", ["hip_sell", "hip_csr_base", "hip_csr_merge"])
def test_patched_reference_driver_runs_the_hip_modes(tmp_path, mode):
    rows, cols, rp, cj, va = G.stencil27(11, 12, 13)
    path = str(tmp_path / "stencil_11_12_13.mtx")
    G.write_mtx(path, rows, cols, G.csr_to_coo_numpy(rows, rp, cj, va))
    lst = str(tmp_path / "list.txt")
    open(lst, "w").write(path + "\n")
    out = str(tmp_path / "out.csv")
    p = subprocess.run([BINARY, mode, lst, out], capture_output=True, text=True, timeout=150)
    assert p.returncode == 0, p.stdout + p.stderr
    table = [ln for ln in p.stdout.splitlines() if "stencil_11_12_13" in ln]
    assert len(table) == 1 and "OK" in table[0] and "ERR" not in table[0], p.stdout     # driver_spmv.c:313-314
    fields = open(out).read().strip().split("\n")[1].split(";")                          # Format;File;Rows;Columns;NNZ;GFlops;Time(s)
    assert fields[0] == mode and int(fields[2]) == rows and int(fields[4]) == len(cj)
    assert float(fields[5]) > 0 and float(fields[6]) > 0
