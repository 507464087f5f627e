#!/usr/bin/env python3
"""Run the products of many (kernel variant, block_map) settings of ONE handle back to back, for a rocprofv3
counter pass around this process (tools/profile_bench.py starts it):

    rocprofv3 --pmc FETCH_SIZE ... -- python3 tools/traffic_sweep.py --matrix stencil27 --size 252 --format sell \
        --variants 18,22,4,12 --maps 0,1,2,3,4,5,8 --manifest /tmp/m.json

Every setting gets exactly --launches products and nothing else launches a product kernel (no kernel trial), so
the dispatches of the product kernels, in dispatch order, split evenly over the settings in the order of the
manifest this process writes.
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
    ap.add_argument("--matrix", default="stencil27", choices=["stencil27", "laplacian2d", "rmat"])
    ap.add_argument("--size", type=int, default=252)
    ap.add_argument("--format", default="sell")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--sigma", type=int, default=1024)
    ap.add_argument("--variants", default="0")
    ap.add_argument("--maps", default="0")
    ap.add_argument("--launches", type=int, default=8)
    ap.add_argument("--manifest", required=True)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dtype = S.F64 if args.dtype == "f64" else S.F32
    td = torch.float64 if dtype == S.F64 else torch.float32
    if args.matrix == "stencil27":
        rows, cols, rp, cj, va = G.stencil27(args.size, device=dev)
    elif args.matrix == "laplacian2d":
        rows, cols, rp, cj, va = G.laplacian2d(args.size, device=dev)
    else:
        rows, cols, rp, cj, va = G.rmat(args.size, 16, device=dev)
    va = va.to(td)
    nnz = int(rp[-1])
    fmt = {"sell": S.SELL, "sellp": S.SELLP, "ell": S.ELL, "csr_scalar": S.CSR_SCALAR, "csr_vector": S.CSR_VECTOR,
           "csr_merge": S.CSR_MERGE, "csr_bal": S.CSR_BAL, "sell_dfc": S.SELL_DFC}[args.format]
    st = torch.cuda.current_stream().cuda_stream
    sliced = fmt in (S.SELL, S.SELL_DFC)
    m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), dtype, 64 if sliced else 0,
                               args.sigma if sliced else 0, st, keepalive=(rp, cj, va))
    g = torch.Generator(device=dev).manual_seed(1234)
    x = torch.rand(cols, dtype=td, device=dev, generator=g)
    x /= torch.linalg.vector_norm(x)
    y = torch.empty(rows, dtype=td, device=dev)
    order = []
    for v in [int(t) for t in args.variants.split(",")]:
        for k in [int(t) for t in args.maps.split(",")]:
            m.set_option("variant", v)
            m.set_option("block_map", k)
            for _ in range(args.launches):
                m.mult_dev(x.data_ptr(), y.data_ptr(), st)
            torch.cuda.synchronize()
            order.append([v, k])
    m.free()
    with open(args.manifest, "w") as f:
        json.dump({"format": args.format, "matrix": args.matrix, "size": args.size, "dtype": args.dtype, "order": order,
                   "launches": args.launches}, f)
    print(f"traffic_sweep: {len(order)} settings x {args.launches} products")


if __name__ == "__main__":
    main()
