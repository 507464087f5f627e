"""Row-block multi-process SpMV on CPU: gloo, world_size 2 and 3.

The partitioning, halo bookkeeping and exchange of spmv_amd.distributed are the
code under test; the local products are done by the CPU oracle through an
injected engine (on the GPU box the same class uses the HIP engine over RCCL).
Bar: bit-exact against the oracle's single-process mult_csr_base, because every
row is summed by one rank in ascending column order.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spmv_amd import generators as G


class OracleEngine:
    """Test-only engine: local products by oracle.mult_csr_base."""

    def matrix(self, fmt, rows, cols, rp, cj, va, p0=0, p1=0, row_map=None):
        import oracle
        rp_n, cj_n, va_n = rp.numpy().astype(np.int32), cj.numpy().astype(np.int32), va.numpy().astype(np.float64)
        rm = None if row_map is None else row_map.numpy().astype(np.int64)

        class Local:
            stored = len(cj_n)
            variant = 0

            def mult(self, x, y):
                res = oracle.mult_csr_base(rp_n, cj_n, va_n, np.ascontiguousarray(x.numpy(), np.float64))
                if rm is None:
                    y.numpy()[:rows] = res
                else:
                    y.numpy()[rm] = res
        return Local()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, exchange, kind, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from spmv_amd.distributed import RowBlockSpmv
        import spmv_amd as S
        if kind == "stencil":
            nx, ny, nz = 7, 6, 4 * world
            total = nx * ny * nz
            per = (nz // world) * nx * ny
            r0, r1 = rank * per, (rank + 1) * per
            rows, cols, rp, cj, va = G.stencil27(nx, ny, nz, row_begin=r0, row_end=r1)
        else:                                   # skewed rows, empty rows, long-range columns
            total = 90 * world
            _, _, rp_f, cj_f, va_f = G.power_law_rows(total, seed=9)
            per = total // world
            r0, r1 = rank * per, (rank + 1) * per
            a, b = int(rp_f[r0]), int(rp_f[r1])
            rows, cols = per, total
            rp = (rp_f[r0:r1 + 1] - rp_f[r0]).to(torch.int32)
            cj, va = cj_f[a:b].clone(), va_f[a:b].clone()
        op = RowBlockSpmv(S.SELL, rows, cols, rp, cj, va, S.F64, 64, 1024, row_begin=r0, exchange=exchange,
                          device=torch.device("cpu"), engine=OracleEngine())
        g = torch.Generator().manual_seed(77)
        x_full = torch.rand(total, dtype=torch.float64, generator=g)
        y = torch.full((rows,), float("nan"), dtype=torch.float64)
        if exchange == "halo":
            op.x_local.copy_(x_full[r0:r1])
            op.mult(op.x_local, y)
        else:
            op.mult(x_full[r0:r1].clone(), y)
        # second product with other values: the exchange buffers are reused
        y2 = torch.empty_like(y)
        op.mult((x_full[r0:r1] * 2.0).clone(), y2)
        nrm = op.norm2(y)
        out_q.put((rank, r0, y.numpy().copy(), y2.numpy().copy(), nrm, op.nhalo))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("exchange", ["halo", "allgather"])
@pytest.mark.parametrize("kind", ["stencil", "powerlaw"])
def test_row_block_spmv_matches_single_process(orc, world, exchange, kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, exchange, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if kind == "stencil":
        rows, cols, rp, cj, va = G.stencil27(7, 6, 4 * world)
    else:
        rows, cols, rp, cj, va = G.power_law_rows(90 * world, seed=9)
    g = torch.Generator().manual_seed(77)
    x = torch.rand(rows, dtype=torch.float64, generator=g).numpy()
    want = orc.mult_csr_base(rp.numpy(), cj.numpy(), va.numpy(), x)
    want2 = orc.mult_csr_base(rp.numpy(), cj.numpy(), va.numpy(), x * 2.0)
    got = np.full(rows, np.nan); got2 = np.full(rows, np.nan)
    for rank, r0, y, y2, nrm, nhalo in results:
        got[r0:r0 + len(y)] = y
        got2[r0:r0 + len(y2)] = y2
        assert abs(nrm - np.linalg.norm(want)) <= 1e-12 * np.linalg.norm(want)
        if exchange == "halo" and kind == "stencil":
            planes = 2 if 0 < rank < world - 1 else 1
            assert nhalo == planes * 7 * 6                 # only the neighbouring planes travel
    assert got.tobytes() == want.tobytes()
    assert got2.tobytes() == want2.tobytes()
