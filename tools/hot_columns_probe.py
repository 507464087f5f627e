#!/usr/bin/env python3
"""Does packing the most-referenced columns of a power-law matrix into one contiguous range of x pay?

Relabels the columns of an R-MAT matrix in descending order of their reference count (entry order
inside the rows untouched, so every row's sum is taken in the same order), and times the merge-path
and balanced kernels on both labelings.  x is permuted outside the timed region here.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import spmv_amd as S  # noqa: E402
from spmv_amd import generators as G  # noqa: E402
from spmv_amd.matrix import Matrix  # noqa: E402


def median_ms(fn, steps=10):
    for _ in range(2):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[len(ev) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rows, cols, rp, cj, va = G.rmat(args.scale, 16, device=dev)
    nnz = int(rp[-1])
    st = torch.cuda.current_stream().cuda_stream
    count = torch.zeros(cols, dtype=torch.int32, device=dev)
    count.index_add_(0, cj.long(), torch.ones(1, dtype=torch.int32, device=dev).expand(nnz))
    old_of_new = torch.argsort(count, descending=True, stable=True)
    new_of_old = torch.empty(cols, dtype=torch.int32, device=dev)
    new_of_old[old_of_new] = torch.arange(cols, dtype=torch.int32, device=dev)
    cj_hot = new_of_old[cj.long()].contiguous()
    csum = torch.cumsum(count[old_of_new].double(), 0) / nnz
    for k in (1 << 16, 1 << 18, 1 << 19, 1 << 20, 1 << 22):
        if k <= cols:
            print(f"top {k} columns take {float(csum[k - 1]):.3f} of the references", flush=True)
    x = torch.rand(cols, dtype=torch.float64, device=dev)
    x_hot = x[old_of_new].contiguous()
    y = torch.empty(rows, dtype=torch.float64, device=dev)
    y2 = torch.empty(rows, dtype=torch.float64, device=dev)
    for fmt, name in ((S.CSR_MERGE, "csr_merge"), (S.CSR_BAL, "csr_bal")):
        for variant in (0, 1):
            m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), S.F64, 0, 0, st)
            m.set_option("variant", variant)
            a = median_ms(lambda: m.mult_dev(x.data_ptr(), y.data_ptr(), st))
            m.free()
            m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj_hot.data_ptr(), va.data_ptr(), S.F64, 0, 0, st)
            m.set_option("variant", variant)
            b = median_ms(lambda: m.mult_dev(x_hot.data_ptr(), y2.data_ptr(), st))
            m.free()
            perm_ms = median_ms(lambda: torch.index_select(x, 0, old_of_new, out=x_hot))
            print(f"{name} variant {variant}: {a:.3f} ms as given, {b:.3f} ms relabeled (+ {perm_ms:.3f} ms to permute x), "
                  f"max |dy| {float((y - y2).abs().max()):.1e}", flush=True)


if __name__ == "__main__":
    main()
