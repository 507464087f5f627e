#!/usr/bin/env python3
"""How much of a CSR kernel's time goes into the lines its gathers touch per load instruction.

Three matrices with the rows and entries of the 27-point stencil on n^3 points (27 entries in every row,
same streamed bytes), differing only in where the columns point:
  stencil   the real 27-point stencil (a wave's 64 consecutive entries name ~27 different lines of x)
  packed    row r names the 27 consecutive columns starting at 64 * (r // 64): 64 consecutive entries
            name one or two lines
  spread    entry e of the matrix names column (e * 16) % columns: 64 consecutive entries, 64 lines
              python tools/gather_order_probe.py --n 252 --format csr_scalar --variants 3,11
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402

import spmv_amd as S  # noqa: E402
from spmv_amd.matrix import Matrix  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=252)
    ap.add_argument("--format", default="csr_scalar")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = cols = args.n ** 3
    L = 27
    nnz = rows * L
    rp = (torch.arange(rows + 1, device=dev, dtype=torch.int64) * L).to(torch.int32)
    va = torch.rand(nnz, device=dev, dtype=torch.float64)
    e = torch.arange(nnz, device=dev, dtype=torch.int64)
    fmt = {"csr_scalar": S.CSR_SCALAR, "csr_vector": S.CSR_VECTOR, "csr_merge": S.CSR_MERGE, "csr_bal": S.CSR_BAL,
           "sell": S.SELL, "sellp": S.SELLP}[args.format]
    st = torch.cuda.current_stream().cuda_stream
    x = torch.rand(cols, dtype=torch.float64, device=dev)
    y = torch.empty(rows, dtype=torch.float64, device=dev)
    b_alg = nnz * 12 + rows * 12 + cols * 8
    for name in ("packed", "spread"):
        if name == "packed":
            r = e // L
            cj = torch.clamp((r // 64) * 64 + e % L, max=cols - 1).to(torch.int32)
        else:
            cj = ((e * 16) % cols).to(torch.int32)
        p0, p1 = (64, 1024) if fmt == S.SELL else (0, 0)
        m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), S.F64, p0, p1, st)
        for var in [int(v) for v in args.variants.split(",")]:
            m.set_option("variant", var)
            for _ in range(3):
                m.mult_dev(x.data_ptr(), y.data_ptr(), st)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            for a, b in ev:
                a.record(); m.mult_dev(x.data_ptr(), y.data_ptr(), st); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
            print(f"{args.format} {name:8s} variant {var:3d}  {ms:.4f} ms  {b_alg / ms / 1e6:.0f} GB/s alg "
                  f"({b_alg / ms / 8e7:.1f} % of 8 TB/s)", flush=True)
        m.free()
        del cj
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
