#!/usr/bin/env python3
"""What one rank of the weak-scaling bench does locally per product, on one GPU: the interior product and the
boundary product of a middle z-slab (two neighbours) of the 252 x 252 x (252 N) stencil, without the exchange,
beside the product of the same slab as a matrix of its own (the N = 1 workload).

    python tools/rowblock_local_time.py [--size 252] [--ranks 3]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402

import spmv_amd as S  # noqa: E402
from spmv_amd import generators as G  # noqa: E402
from spmv_amd.distributed import HipEngine  # noqa: E402
from spmv_amd.matrix import Matrix  # noqa: E402


def timed(fn, steps=30):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[len(ms) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=252)
    ap.add_argument("--ranks", type=int, default=3)
    ap.add_argument("--format", default="sell")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    n, P = args.size, args.ranks
    plane = n * n
    rank = P // 2
    r0, r1 = rank * n * plane, (rank + 1) * n * plane
    fmt = {"sell": S.SELL, "sellp": S.SELLP, "csr_scalar": S.CSR_SCALAR, "csr_merge": S.CSR_MERGE}[args.format]
    p0, p1 = (64, 1024) if fmt == S.SELL else (0, 0)
    rows, cols, rp, cj, va = G.stencil27(n, n, n * P, device=dev, row_begin=r0, row_end=r1)
    eng = HipEngine(S.F64)
    blk = eng.rowblock(fmt, rows, r0, cols, rp, cj, va, p0, p1)
    info = blk.info
    x_ext = torch.rand(rows + info.halo, dtype=torch.float64, device=dev)
    y = torch.empty(rows, dtype=torch.float64, device=dev)
    t_int = timed(lambda: blk.mult_interior(x_ext[:rows], y))
    t_bnd = timed(lambda: blk.mult_boundary(x_ext, y))
    t_both = timed(lambda: (blk.mult_interior(x_ext[:rows], y), blk.mult_boundary(x_ext, y)))
    print(f"slab of {n} planes, rank {rank} of {P}: interior {info.interior_nnz} nnz (variant {info.interior_variant}), "
          f"boundary rows {info.boundary_rows} ({info.boundary_nnz} nnz), halo {info.halo}")
    print(f"interior {t_int:.4f} ms, boundary {t_bnd:.4f} ms, both back to back {t_both:.4f} ms")
    blk.free()
    del x_ext
    rows1, cols1, rp1, cj1, va1 = G.stencil27(n, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    m = Matrix.from_device_csr(fmt, rows1, cols1, int(rp1[-1]), rp1.data_ptr(), cj1.data_ptr(), va1.data_ptr(), S.F64, p0, p1, st,
                               keepalive=(rp1, cj1, va1))
    m.analyze(3)
    x = torch.rand(cols1, dtype=torch.float64, device=dev)
    y1 = torch.empty(rows1, dtype=torch.float64, device=dev)
    t_one = timed(lambda: m.mult_dev(x.data_ptr(), y1.data_ptr(), st))
    print(f"the slab as a matrix of its own (N = 1): {t_one:.4f} ms (variant {m.info.variant}, block_map {m.info.block_map}); "
          f"local work of a rank / N = 1 product = {t_both / t_one:.3f}")


if __name__ == "__main__":
    main()
