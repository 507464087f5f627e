#!/usr/bin/env python3
"""Time of building each device layout from a CSR resident in HBM (27-point stencil 252^3 by default).

    python tools/build_time.py [--n 252]
    python tools/build_time.py --matrix rmat --n 26     # balanced CSR and its hot_columns options on R-MAT
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402

import spmv_amd as S  # noqa: E402
from spmv_amd import generators as G  # noqa: E402
from spmv_amd.matrix import Matrix  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=252)
    ap.add_argument("--matrix", default="stencil27", choices=["stencil27", "rmat"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    if args.matrix == "rmat":
        rows, cols, rp, cj, va = G.rmat(args.n, 16, device=dev)
        nnz = int(rp[-1])
        lens = (rp[1:] - rp[:-1]).to(torch.int64)
        print(f"# rmat scale {args.n}: {rows} rows, {nnz} nnz, longest row {int(lens.max())}, empty rows {int((lens == 0).sum())}")
        # rows whose entries all fall into one of 8 equally referenced column groups: all rows of one entry, 1/8 of
        # those with two, 1/64 with three, ...
        hist = torch.bincount(lens.clamp(max=8), minlength=9).tolist()
        single = sum(hist[k] * 8.0 ** (1 - k) for k in range(1, 8))
        print(f"# rows by length 1..7: {hist[1:8]}; expected rows with a single column group: {single / max(rows - hist[0], 1):.3f} "
              f"of the non-empty rows")
        for _ in range(2):
            m = Matrix.from_device_csr(S.CSR_BAL, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), S.F64, 0, 0, st)
            for hot in (1, 2):
                torch.cuda.synchronize()
                t = time.perf_counter()
                m.set_option("hot_columns", hot)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t
                print(f"hot_columns {hot} build {dt * 1e3:8.2f} ms", flush=True)
                m.set_option("hot_columns", 0)
            m.free()
        return
    rows, cols, rp, cj, va = G.stencil27(args.n, device=dev)
    nnz = int(rp[-1])
    print(f"# stencil27 {args.n}^3: {rows} rows, {nnz} nnz")
    for name, fmt, p0, p1 in [("sell-64-1024", S.SELL, 64, 1024), ("sell-64-16384", S.SELL, 64, 16384),
                              ("sellp", S.SELLP, 0, 0), ("merge", S.CSR_MERGE, 0, 0), ("bal", S.CSR_BAL, 0, 0)]:
        for _ in range(2):                     # the second build is the one reported (first: code load)
            torch.cuda.synchronize()
            t = time.perf_counter()
            m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), S.F64, p0, p1, st)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            m.free()
        print(f"{name:16s} build {dt * 1e3:8.2f} ms")


if __name__ == "__main__":
    main()
