"""Row-block SpMV with the HIP engine: two ranks rehearsed on ONE GPU.

The GPU box has a single MI355X, so both ranks use cuda:0 and talk over gloo
(halo staged through the host); the local products, the interior/boundary split,
the row-mapped boundary kernel and the stream ordering are the real ones.  The
8-GPU RCCL run itself is done by the round-end driver.  Tolerance: rel-L2 <=
1e-13 against the oracle's single-process product (analyze() may pick a
reassociating SELL kernel).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spmv_amd import generators as G

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, exchange, fmt_name, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import spmv_amd as S
        from spmv_amd.distributed import RowBlockSpmv
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        S.lib().spmv_hip_init(0)
        nx, ny, nz = 20, 18, 8 * world
        total = nx * ny * nz
        per = (nz // world) * nx * ny
        r0, r1 = rank * per, (rank + 1) * per
        rows, cols, rp, cj, va = G.stencil27(nx, ny, nz, device=dev, row_begin=r0, row_end=r1)
        fmt = {"sell": S.SELL, "csr_merge": S.CSR_MERGE}[fmt_name]
        op = RowBlockSpmv(fmt, rows, cols, rp, cj, va, S.F64, 64 if fmt == S.SELL else 0, 1024 if fmt == S.SELL else 0,
                          row_begin=r0, exchange=exchange, device=dev)
        g = torch.Generator().manual_seed(5)
        x_full = torch.rand(total, dtype=torch.float64, generator=g)
        y = torch.full((rows,), float("nan"), dtype=torch.float64, device=dev)
        for scale in (1.0, -3.0):                       # twice: buffers and streams are reused
            op.x_local.copy_((x_full[r0:r1] * scale).to(dev))
            op.mult(op.x_local, y)
            torch.cuda.synchronize()
        out_q.put((rank, r0, y.cpu().numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,fmt_name", [("halo", "sell"), ("allgather", "sell"), ("halo", "csr_merge")])
def test_two_ranks_on_one_gpu(orc, exchange, fmt_name):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, exchange, fmt_name, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=110) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rows, cols, rp, cj, va = G.stencil27(20, 18, 8 * world)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(rows, dtype=torch.float64, generator=g).numpy() * -3.0
    want = orc.mult_csr_base(rp.numpy(), cj.numpy(), va.numpy(), x)
    got = np.full(rows, np.nan)
    for rank, r0, y in results:
        got[r0:r0 + len(y)] = y
    assert not np.isnan(got).any()
    assert orc.rel_l2(got, want) <= 1e-13
