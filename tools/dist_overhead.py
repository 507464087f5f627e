#!/usr/bin/env python3
"""What the row-block machinery costs beside the plain product, on one GPU: the same matrix as one handle
(spmv_hip_bench) and cut into N row blocks that all sit on device 0 (spmv_hip_dist_bench: pack, peer copies
on the same device, interior and boundary products on a stream pair per block).  With one GPU per block the
blocks run side by side; here they share the GPU, so the ideal is the plain product's time.

    python tools/dist_overhead.py --n 160 --format sell --blocks 1,2,4,8

Every block count runs twice: launches issued by the handle's helper threads, one per block (the default), and
by the calling thread alone (SPMV_DIST_THREADS=0).
"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import numpy as np  # noqa: E402

import spmv_amd as S  # noqa: E402
from spmv_amd import _lib as L  # noqa: E402
from spmv_amd import generators as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=160)
    ap.add_argument("--format", default="sell")
    ap.add_argument("--blocks", default="1,2,4,8")
    ap.add_argument("--seconds", type=float, default=0.5)
    args = ap.parse_args()
    fmt = {"sell": S.SELL, "csr_scalar": S.CSR_SCALAR, "csr_merge": S.CSR_MERGE, "sellp": S.SELLP}[args.format]
    p0, p1 = (64, 1024) if fmt == S.SELL else (0, 0)
    rows, cols, rp, cj, va = G.stencil27(args.n)
    coo = G.csr_to_coo_numpy(rows, rp, cj, va)
    nnz = len(coo)
    lib = S.lib()
    lib.spmv_hip_init(0)
    it, sec = C.c_int(0), C.c_double(0.0)
    print(f"stencil27 {args.n}^3: {rows} rows, {nnz} nnz, {args.format}")
    for nb in [int(v) for v in args.blocks.split(",")]:
        for threads in ("1", "0"):
            os.environ["SPMV_DIST_THREADS"] = threads          # read when the handle is created
            devices = (C.c_int * nb)(*([0] * nb))
            d = L.check_handle(lib.spmv_hip_dist_create(fmt, rows, cols, nnz, coo.ctypes.data_as(C.c_void_p), S.F64, p0, p1, nb, devices))
            L.check(lib.spmv_hip_dist_analyze(d, 3))
            x = np.full(rows, 1.0 / np.sqrt(rows))
            L.check(lib.spmv_hip_dist_bench(d, 10, args.seconds, x.ctypes.data_as(C.c_void_p), C.byref(it), C.byref(sec)))
            ms = sec.value / it.value * 1e3
            who = "a helper thread per block" if threads == "1" else "the calling thread alone "
            issue, total = C.c_double(0.0), C.c_double(0.0)
            L.check(lib.spmv_hip_dist_issue_time(d, 200, C.byref(issue), C.byref(total)))
            print(f"  {nb} block(s) on device 0, {who}: {ms:.4f} ms per product (host clock, {it.value} products), "
                  f"{2.0 * nnz / ms / 1e6:.0f} GFLOP/s; issuing one product takes the host {issue.value / 200 * 1e6:.1f} us",
                  flush=True)
            lib.spmv_hip_dist_free(d)


if __name__ == "__main__":
    main()
