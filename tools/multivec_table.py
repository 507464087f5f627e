#!/usr/bin/env python3
"""Time y_v = A x_v for blocks of n vectors (spmv_hip_mult_mv_dev) on one GPU.

Algorithmic bytes of one block product: nnz*(V+4) + rows*(4 + n*V) + cols*n*V -- the matrix once,
every vector element once.  Prints one markdown row per (format, n): time, GFLOP/s = 2*nnz*n/t,
the fraction of 8 TB/s, and the time of n single products for comparison.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import spmv_amd as S  # noqa: E402
from spmv_amd import generators as G  # noqa: E402
from spmv_amd.matrix import Matrix  # noqa: E402

FORMATS = {"csr_scalar": (S.CSR_SCALAR, 0, 0), "csr_merge": (S.CSR_MERGE, 0, 0), "csr_bal": (S.CSR_BAL, 0, 0),
           "csr_vector": (S.CSR_VECTOR, 0, 0), "sell": (S.SELL, 64, 1024), "sell_dfc": (S.SELL_DFC, 64, 1024),
           "sellp": (S.SELLP, 0, 0)}


def median_ms(fn, steps):
    for _ in range(2):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="stencil27", choices=["stencil27", "laplacian2d", "rmat"])
    ap.add_argument("--n", type=int, default=252)
    ap.add_argument("--formats", default="sell,sellp,csr_scalar,sell_dfc")
    ap.add_argument("--vectors", default="1,2,4,8,16")
    ap.add_argument("--dtypes", default="f64")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--block-maps", default="", help="comma list: time the block products under these block_map values "
                    "instead of the one analyze picked for single products")
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
    print("| format | dtype | vectors | ms / block | ms of n single products | GFLOP/s | GB/s (alg) | % of 8 TB/s |")
    print("|---|---|---|---|---|---|---|---|")
    nmax = max(int(v) for v in args.vectors.split(","))
    for dname in args.dtypes.split(","):
        dtype, td, vb = (S.F64, torch.float64, 8) if dname == "f64" else (S.F32, torch.float32, 4)
        vals = va.to(td)
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.rand(nmax, cols, dtype=td, device=dev, generator=g)
        y = torch.empty(nmax, rows, dtype=td, device=dev)
        for name in args.formats.split(","):
            fmt, p0, p1 = FORMATS[name]
            m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), vals.data_ptr(), dtype, p0, p1, st)
            m.analyze(3)
            single = median_ms(lambda: m.mult_dev(x.data_ptr(), y.data_ptr(), st), args.steps)
            maps = [int(v) for v in args.block_maps.split(",")] if args.block_maps else [None]
            for n in (int(v) for v in args.vectors.split(",")):
                for bm in maps:
                    if bm is not None:
                        m.set_option("block_map", bm)
                    ms = median_ms(lambda: m.mult_mv_dev(n, x.data_ptr(), cols, y.data_ptr(), rows, st), args.steps)
                    b_alg = nnz * (vb + 4) + rows * (4 + n * vb) + cols * n * vb
                    label = name if bm is None else f"{name} block_map {bm}"
                    print(f"| {label} | {dname} | {n} | {ms:.3f} | {single * n:.3f} | {2 * nnz * n / ms / 1e6:.0f} | "
                          f"{b_alg / ms / 1e6:.0f} | {b_alg / ms / 1e6 / 80:.1f} |", flush=True)
            m.free()
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
