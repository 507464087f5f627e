import sys, time, torch
sys.path.insert(0, '/root/repo')
import spmv_amd as S
from spmv_amd import generators as G
from spmv_amd.matrix import Matrix
dev = torch.device("cuda:0")
rows, cols, rp, cj, va = G.stencil27(252, device=dev)
nnz = int(rp[-1]); st = torch.cuda.current_stream().cuda_stream
for name, fmt, p0, p1 in [("sell-64-1024", S.SELL, 64, 1024), ("sell-64-16384", S.SELL, 64, 16384), ("sellp", S.SELLP, 0, 0), ("merge", S.CSR_MERGE, 0, 0), ("bal", S.CSR_BAL, 0, 0)]:
    for rep in range(2):
        torch.cuda.synchronize(); t = time.perf_counter()
        m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), S.F64, p0, p1, st)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        m.free()
    print(f"{name:16s} build {dt*1e3:8.2f} ms")
