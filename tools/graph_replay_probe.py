#!/usr/bin/env python3
"""What a product of a small matrix costs when its launches are queued ahead: `--steps` products captured into one HIP graph
(torch.cuda.CUDAGraph on the stream the library launches on), the replay timed with one event pair, against the same
products launched one by one from Python with an event pair each (what bench.py does)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402

import spmv_amd as S  # noqa: E402
from spmv_amd import generators as G  # noqa: E402
from spmv_amd.matrix import Matrix  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rows, cols, rp, cj, va = G.laplacian2d(args.n, device=dev)
    nnz = int(rp[-1])
    x = torch.rand(cols, dtype=torch.float64, device=dev)
    y = torch.empty(rows, dtype=torch.float64, device=dev)
    b_alg = 12 * nnz + 12 * rows + 8 * cols
    for name, fmt, p0 in (("csr_scalar", S.CSR_SCALAR, 0), ("sellp", S.SELLP, 64)):
        st = torch.cuda.current_stream().cuda_stream
        m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), S.F64, p0, 0, st,
                                   keepalive=(rp, cj, va))
        m.analyze(3)
        for _ in range(5):
            m.mult_dev(x.data_ptr(), y.data_ptr(), st)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in ev:
            a.record(); m.mult_dev(x.data_ptr(), y.data_ptr(), st); b.record()
        torch.cuda.synchronize()
        one_by_one = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(args.steps):
            m.mult_dev(x.data_ptr(), y.data_ptr(), st)
        b.record()
        torch.cuda.synchronize()
        in_a_row = a.elapsed_time(b) / args.steps
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for _ in range(args.steps):
                    m.mult_dev(x.data_ptr(), y.data_ptr(), side.cuda_stream)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        a.record(); graph.replay(); b.record()
        torch.cuda.synchronize()
        replay = a.elapsed_time(b) / args.steps
        print(f"{name}: variant {m.info.variant}  event pair per launch {one_by_one * 1e3:.2f} us  {args.steps} launches in a row "
              f"{in_a_row * 1e3:.2f} us  graph replay {replay * 1e3:.2f} us per product "
              f"({b_alg / replay / 1e6 / 8000 * 100:.1f} % of 8 TB/s algorithmic)", flush=True)
        m.free()


if __name__ == "__main__":
    main()
