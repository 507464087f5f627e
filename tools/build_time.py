#!/usr/bin/env python3
"""Time of building each device layout from a CSR resident in HBM (27-point stencil 252^3 by default).

    python tools/build_time.py [--n 252]
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
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rows, cols, rp, cj, va = G.stencil27(args.n, device=dev)
    nnz = int(rp[-1])
    st = torch.cuda.current_stream().cuda_stream
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
