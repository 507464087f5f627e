#!/usr/bin/env python3
"""Time the kernel variants of one format on a synthetic workload (one GPU).

    python tools/tune.py --format sell --matrix stencil27 --n 252 --variants 0,1,2

Prints per variant: median/min kernel ms (HIP events on the launch stream),
algorithmic GB/s, and rel-L2 difference of y against variant 0.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402

import spmv_amd as S  # noqa: E402
from spmv_amd import generators as G  # noqa: E402
from spmv_amd.matrix import Matrix  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--format", default="sell")
    ap.add_argument("--matrix", default="stencil27")
    ap.add_argument("--n", type=int, default=252)
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--sigma", type=int, default=1024)
    ap.add_argument("--p0", default="0", help="comma list of p0 values (lanes / tile size / slice height)")
    ap.add_argument("--variants", default="0")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--in-a-row", action="store_true", help="also time the same number of products queued between one event pair")
    ap.add_argument("--block-maps", default="", help="comma list of option block_map values to time every variant under (0, 1)")
    ap.add_argument("--hot", type=int, default=0, help="option hot_columns of CSR-based handles (0, 1, 2)")
    ap.add_argument("--hot-variants", default="", help="with --hot 2: comma list of tile kernels of the stacked matrix (0, 1, 3, 4)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = S.F64 if args.dtype == "f64" else S.F32
    td = torch.float64 if dtype == S.F64 else torch.float32
    vb = 8 if dtype == S.F64 else 4
    if args.matrix == "stencil27":
        rows, cols, rp, cj, va = G.stencil27(args.n, device=dev)
    elif args.matrix == "laplacian2d":
        rows, cols, rp, cj, va = G.laplacian2d(args.n, device=dev)
    elif args.matrix.startswith("standin:"):        # a stand-in of the reference's sweep list, e.g. standin:audikw_1
        from spmv_amd import standins
        rows, cols, rp, cj, va = standins.standin(args.matrix.split(":", 1)[1], 1.0, device=dev)
    else:
        rows, cols, rp, cj, va = G.rmat(args.n, 16, device=dev)
    va = va.to(td)
    nnz = int(rp[-1])
    b_alg = nnz * (vb + 4) + rows * (4 + vb) + cols * vb
    fmt = {"sell": S.SELL, "sellp": S.SELLP, "ell": S.ELL, "csr_scalar": S.CSR_SCALAR, "csr_vector": S.CSR_VECTOR,
           "csr_merge": S.CSR_MERGE, "csr_bal": S.CSR_BAL, "sell_dfc": S.SELL_DFC}[args.format]
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.rand(cols, dtype=td, device=dev, generator=g)
    y = torch.empty(rows, dtype=td, device=dev)
    y0 = None
    print(f"# {args.format} {args.matrix} n={args.n} {args.dtype}: rows {rows} nnz {nnz} B_alg {b_alg / 1e9:.3f} GB")
    for p0 in [int(v) for v in args.p0.split(",")]:
        pp0 = p0 or (64 if fmt in (S.SELL, S.SELL_DFC) else 0)
        m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), dtype, pp0,
                                   args.sigma if fmt in (S.SELL, S.SELL_DFC) else 0, st)
        if args.hot:
            m.set_option("hot_columns", args.hot)
        variants = [v if v == "auto" else int(v) for v in args.variants.split(",")]
        if args.hot_variants:
            variants = [("hot", int(v)) for v in args.hot_variants.split(",")]
        maps = [int(v) for v in args.block_maps.split(",")] if args.block_maps else [None]
        for var, bmap in [(v, b) for v in variants for b in maps]:
            if bmap is not None:
                m.set_option("block_map", bmap)
            if isinstance(var, tuple):
                m.set_option("hot_variant", var[1])
                var = 100 + var[1]
            elif var == "auto":
                m.analyze(3)
                var = m.info.variant
            else:
                m.set_option("variant", var)
            for _ in range(3):
                m.mult_dev(x.data_ptr(), y.data_ptr(), st)
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
            for a, b in ev:
                a.record(); m.mult_dev(x.data_ptr(), y.data_ptr(), st); b.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(b) for a, b in ev)
            in_a_row = None
            if args.in_a_row:           # the same launches queued between one event pair (products of a few microseconds:
                a, b = ev[0]            # an event pair around a launch on an empty queue reads ~2 us more than the kernel)
                a.record()
                for _ in range(args.steps):
                    m.mult_dev(x.data_ptr(), y.data_ptr(), st)
                b.record()
                torch.cuda.synchronize()
                in_a_row = a.elapsed_time(b) / args.steps
            if y0 is None:
                y0 = y.clone()
            err = float(torch.linalg.vector_norm(y - y0) / torch.linalg.vector_norm(y0))
            med = ms[len(ms) // 2]
            print(f"p0={p0:6d} variant={var:3d}" + (f" map={bmap}" if bmap is not None else "") + f"  median {med:8.4f} ms  min {ms[0]:8.4f} ms  "
                  f"{b_alg / med / 1e6:8.1f} GB/s alg ({b_alg / med / 1e6 / 80:.1f}% of 8 TB/s)  "
                  f"{2 * nnz / med / 1e6:8.1f} GFLOP/s  stored {m.info.stored}  relerr-vs-first {err:.2e}", flush=True)
            info = m.info
            print("RESULT " + json.dumps({"format": args.format, "matrix": args.matrix, "n": args.n, "dtype": args.dtype,
                                          "hot": args.hot, "hot_variant": info.hot_variant, "variant": info.variant, "block_map": info.block_map, "lanes_per_row": info.lanes_per_row,
                                          "median_ms": med, "min_ms": ms[0], "in_a_row_ms": in_a_row, "rows": rows, "cols": cols, "nnz": nnz,
                                          "stored": info.stored, "b_alg": b_alg}), flush=True)
        m.free()


if __name__ == "__main__":
    main()
