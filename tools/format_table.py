#!/usr/bin/env python3
"""Every format on one synthetic matrix, tuned by analyze(): a markdown table of kernel time,
algorithmic GB/s and the fraction of the 8 TB/s HBM peak.

    python tools/format_table.py --matrix stencil27 --n 252 --dtypes f64,f32
    python tools/format_table.py --matrix rmat --n 26 --formats csr_merge,csr_bal
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402

import spmv_amd as S  # noqa: E402
from spmv_amd import generators as G  # noqa: E402
from spmv_amd.matrix import Matrix  # noqa: E402

FORMATS = {"csr_scalar": (S.CSR_SCALAR, 0, 0), "csr_vector": (S.CSR_VECTOR, 0, 0), "csr_merge": (S.CSR_MERGE, 0, 0),
           "csr_bal": (S.CSR_BAL, 0, 0), "sell": (S.SELL, 64, 1024), "sell_dfc": (S.SELL_DFC, 64, 1024), "sellp": (S.SELLP, 0, 0),
           "ell": (S.ELL, 0, 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="stencil27", choices=["stencil27", "laplacian2d", "rmat"])
    ap.add_argument("--n", type=int, default=252)
    ap.add_argument("--formats", default=",".join(FORMATS))
    ap.add_argument("--dtypes", default="f64")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.matrix == "stencil27":
        rows, cols, rp, cj, va = G.stencil27(args.n, device=dev)
    elif args.matrix == "laplacian2d":
        rows, cols, rp, cj, va = G.laplacian2d(args.n, device=dev)
    else:
        rows, cols, rp, cj, va = G.rmat(args.n, 16, device=dev)
    nnz = int(rp[-1])
    st = torch.cuda.current_stream().cuda_stream
    print(f"{args.matrix} n={args.n}: {rows} rows, {nnz} nnz\n")
    print("| format | dtype | variant | ms / SpMV | GFLOP/s | GB/s (alg) | % of 8 TB/s | stored / nnz |")
    print("|---|---|---|---|---|---|---|---|")
    for dname in args.dtypes.split(","):
        dtype, td, vb = (S.F64, torch.float64, 8) if dname == "f64" else (S.F32, torch.float32, 4)
        vals = va.to(td)
        b_alg = nnz * (vb + 4) + rows * (4 + vb) + cols * vb
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.rand(cols, dtype=td, device=dev, generator=g)
        y = torch.empty(rows, dtype=td, device=dev)
        for name in args.formats.split(","):
            fmt, p0, p1 = FORMATS[name.split("+")[0]]       # "csr_merge+hot": with the hot_columns option
            try:
                m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), vals.data_ptr(), dtype,
                                           p0, p1, st)
            except S.SpmvHipError as e:
                print(f"| {name} | {dname} | - | failed: {str(e)[:60]} | | | | |", flush=True)
                continue
            if name.endswith("+hot"):
                m.set_option("hot_columns", 1)
            if name.endswith("+xcd"):                       # columns dealt over the XCDs
                m.set_option("hot_columns", 2)
            m.analyze(3)
            info = m.info
            variant = f"{info.lanes_per_row} lanes, {info.variant}" if fmt == S.CSR_VECTOR else str(info.variant)
            if info.hot_columns == 2:
                variant += f", tiles {info.hot_variant}"
            for _ in range(3):
                m.mult_dev(x.data_ptr(), y.data_ptr(), st)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            for a, b in ev:
                a.record(); m.mult_dev(x.data_ptr(), y.data_ptr(), st); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]
            print(f"| {name} | {dname} | {variant} | {ms:.3f} | {2 * nnz / ms / 1e6:.0f} | {b_alg / ms / 1e6:.0f} | "
                  f"{b_alg / ms / 1e6 / 80:.1f} | {info.stored / max(nnz, 1):.3f} |", flush=True)
            m.free()
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
