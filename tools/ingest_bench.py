#!/usr/bin/env python3
"""Times the ingest steps in front of the product on one matrix: MatrixMarket parse (by thread
count), binary load, (row, col) sort on the host and in HBM, and format builds from a device CSR.

    python tools/ingest_bench.py [--n 100] [--entries 20000000]
"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import numpy as np  # noqa: E402

import spmv_amd as S  # noqa: E402
from spmv_amd import generators as G  # noqa: E402
from spmv_amd.matrix import sort_coo  # noqa: E402


def timed(label, fn):
    t = time.perf_counter()
    out = fn()
    print(f"{label:46s} {time.perf_counter() - t:8.3f} s", flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=80, help="edge of the 27-point stencil written as .mtx / .bin")
    ap.add_argument("--entries", type=int, default=20_000_000, help="entries of the random COO that is sorted")
    args = ap.parse_args()
    rows, cols, rp, cj, va = G.stencil27(args.n)
    coo = G.csr_to_coo_numpy(rows, rp, cj, va)
    with tempfile.TemporaryDirectory() as d:
        mtx, binf = os.path.join(d, "m.mtx"), os.path.join(d, "m.bin")
        with open(mtx, "w") as f:
            f.write(f"%%MatrixMarket matrix coordinate real general\n{rows} {cols} {len(coo)}\n")
            np.savetxt(f, np.column_stack([coo["i"] + 1, coo["j"] + 1, coo["a"]]), fmt="%d %d %.17g")
        print(f"# stencil27 {args.n}^3: {rows} rows, {len(coo)} entries, .mtx {os.path.getsize(mtx) / 1e6:.0f} MB")
        for th in (1, 4, 16):
            os.environ["SPMV_HIP_IO_THREADS"] = str(th)
            timed(f"load_mtx, {th} threads", lambda: S.load_mtx(mtx))
        S.save_bin(binf, rows, cols, coo)
        timed("load_bin", lambda: S.load_bin(binf))
    rng = np.random.default_rng(1)
    n = args.entries
    big = S.make_coo(rng.integers(0, 1 << 22, n).astype(np.int32), rng.integers(0, 1 << 22, n).astype(np.int32),
                     rng.random(n))
    a, b = big.copy(), big.copy()
    os.environ["SPMV_HIP_DEVICE_SORT_MIN"] = str(10 ** 12)
    timed(f"sort_coo on the host, {n} entries", lambda: sort_coo(a))
    os.environ["SPMV_HIP_DEVICE_SORT_MIN"] = "1"
    timed("sort_coo in HBM (incl. both PCIe copies), first", lambda: sort_coo(b))
    b[:] = big
    timed("sort_coo in HBM (incl. both PCIe copies), again", lambda: sort_coo(b))
    assert a.tobytes() == b.tobytes()
    # the whole creator from an unsorted host COO: sort + COO -> CSR + upload
    dim = 1 << 22
    for label, csr_min, sort_min in (("on the host", 10 ** 12, 10 ** 12), ("in HBM", 1, 1)):
        os.environ["SPMV_HIP_DEVICE_CSR_MIN"] = str(csr_min)
        os.environ["SPMV_HIP_DEVICE_SORT_MIN"] = str(sort_min)
        c = big.copy()
        m = timed(f"create_csr from the unsorted COO, {label}", lambda: S.Matrix.csr(dim, dim, c, 1, S.F64, S.CSR_SCALAR))
        m.free()


if __name__ == "__main__":
    main()
