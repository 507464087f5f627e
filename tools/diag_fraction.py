#!/usr/bin/env python3
"""Share of the C2 stencil that the index-skipping kernels cover: diagonal slice columns of the sliced
formats, patterned 64-row blocks of lane-per-row CSR."""
import sys, torch
sys.path.insert(0, "/root/repo")
import spmv_amd as S
from spmv_amd import generators as G
from spmv_amd.matrix import Matrix
dev = torch.device("cuda:0")
rows, cols, rp, cj, va = G.stencil27(252, device=dev)
nnz = int(rp[-1]); st = torch.cuda.current_stream().cuda_stream
for fmt, p0, p1, name in ((S.SELL, 64, 1024, "sell-64-1024"), (S.SELL, 64, 64, "sell-64-64"), (S.SELLP, 0, 0, "sellp")):
    m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), S.F64, p0, p1, st)
    i = m.info
    print(name, "diag columns", i.diag_columns, "of", i.stored // 64, f"= {100.0 * i.diag_columns / (i.stored // 64):.2f} %")
    m.free()
m = Matrix.from_device_csr(S.CSR_SCALAR, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), S.F64, 0, 0, st)
nb = (rows + 63) // 64
print("csr_scalar pattern blocks", m.info.pattern_blocks, "of", nb, f"= {100.0 * m.info.pattern_blocks / nb:.2f} %")
m.free()
