"""Row-block multi-process SpMV on CPU: gloo, world_size 2, 3 and 8.

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
    """Test-only engine: local products by oracle.mult_csr_base; the row-block split restated in numpy
    with the semantics of spmv_hip_rowblock_* (include/spmv_hip.h)."""

    def matrix(self, fmt, rows, cols, rp, cj, va, p0=0, p1=0):
        import oracle
        rp_n, cj_n, va_n = rp.numpy().astype(np.int32), cj.numpy().astype(np.int32), va.numpy().astype(np.float64)

        class Local:
            stored = len(cj_n)
            variant = 0

            def mult(self, x, y):
                y.numpy()[:rows] = oracle.mult_csr_base(rp_n, cj_n, va_n, np.ascontiguousarray(x.numpy(), np.float64))
        return Local()

    def rowblock(self, fmt, rows, row_begin, cols_global, rp, cj, va, p0=0, p1=0):
        import oracle
        rp_n, cj_n, va_n = rp.numpy().astype(np.int64), cj.numpy().astype(np.int64), va.numpy().astype(np.float64)
        end = row_begin + rows
        row_of = np.repeat(np.arange(rows), np.diff(rp_n))
        remote = (cj_n < row_begin) | (cj_n >= end)
        bnd = np.zeros(rows, bool)
        bnd[row_of[remote]] = True
        halo = np.unique(cj_n[remote])
        ext = np.where(remote, rows + np.searchsorted(halo, cj_n), cj_n - row_begin).astype(np.int32)
        keep_i = ~bnd[row_of]
        rp_i = np.zeros(rows + 1, np.int32)
        rp_i[1:] = np.cumsum(np.where(bnd, 0, np.diff(rp_n)))
        bnd_rows = np.nonzero(bnd)[0]
        rp_b = np.zeros(len(bnd_rows) + 1, np.int32)
        rp_b[1:] = np.cumsum(np.diff(rp_n)[bnd])
        ci, vi, cb, vb = ext[keep_i], va_n[keep_i], ext[~keep_i], va_n[~keep_i]

        class Block:
            stored = len(cj_n)
            variant = 0
            halo_columns = torch.from_numpy(halo.astype(np.int64))
            send = np.zeros(0, np.int64)

            def set_send_list(self, index):
                self.send = index.numpy().astype(np.int64)
                assert ((self.send >= 0) & (self.send < rows)).all()

            def pack(self, x, send):
                send.numpy()[:] = x.numpy()[self.send]

            def mult_interior(self, x, y):
                y.numpy()[:rows] = oracle.mult_csr_base(rp_i, ci, vi, np.ascontiguousarray(x.numpy()[:rows], np.float64))

            def mult_boundary(self, x_ext, y):
                if len(bnd_rows):
                    y.numpy()[bnd_rows] = oracle.mult_csr_base(rp_b, cb, vb, np.ascontiguousarray(x_ext.numpy(), np.float64))
        return Block()


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
        elif kind == "stencil_uneven":          # 53 planes dealt like 530 over 8 GPUs: slabs of 6 or 7 (there: 66 or 67)
            nx, ny, nz = 5, 4, 53
            total = nx * ny * nz
            r0, r1 = (rank * nz // world) * nx * ny, ((rank + 1) * nz // world) * nx * ny
            rows, cols, rp, cj, va = G.stencil27(nx, ny, nz, row_begin=r0, row_end=r1)
        elif kind == "blockdiag":               # no rank needs anything from another one
            per = 50
            total = per * world
            r0, r1 = rank * per, (rank + 1) * per
            rows, _, rp, cj, va = G.band(per, 3, seed=rank + 1)
            cj = cj + r0
            cols = total
        else:                                   # skewed rows, empty rows, long-range columns
            from spmv_amd.distributed import row_blocks_by_nnz, slice_row_block
            total = 90 * world
            _, _, rp_f, cj_f, va_f = G.power_law_rows(total, seed=9)
            if kind == "powerlaw_nnz":          # unequal blocks with equal shares of the non-zeros
                bounds = row_blocks_by_nnz(rp_f, world, snap=2)
            else:
                bounds = [r * (total // world) for r in range(world)] + [total]
            r0, r1 = bounds[rank], bounds[rank + 1]
            rows, cols = r1 - r0, total
            rp, cj, va = slice_row_block(rp_f, cj_f, va_f, r0, r1)
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


def test_row_blocks_by_nnz_balances_and_snaps():
    from spmv_amd.distributed import row_blocks_by_nnz, slice_row_block
    rows, cols, rp, cj, va = G.rmat(14, 8, seed=5)
    nnz = int(rp[-1])
    for world in (2, 4, 8):
        b = row_blocks_by_nnz(rp, world, snap=64)
        assert b[0] == 0 and b[-1] == rows and len(b) == world + 1 and all(x <= y for x, y in zip(b, b[1:]))
        assert all(x % 64 == 0 for x in b[1:-1])
        shares = [int(rp[b[p + 1]]) - int(rp[b[p]]) for p in range(world)]
        longest = int((rp[1:] - rp[:-1]).max())
        assert max(abs(s - nnz / world) for s in shares) <= 64 * longest        # off by at most one snap of rows
    rp_b, cj_b, va_b = slice_row_block(rp, cj, va, 128, 640)
    assert int(rp_b[0]) == 0 and int(rp_b[-1]) == len(cj_b) == int(rp[640]) - int(rp[128])
    assert torch.equal(cj_b, cj[int(rp[128]):int(rp[640])])
    # equal shares of the stencil fall on plane boundaries when they can
    rows, cols, rp, cj, va = G.stencil27(8, 8, 32)
    assert row_blocks_by_nnz(rp, 2, snap=64) == [0, 1024, 2048]


CASES = [(w, e, k) for w in (2, 3) for e, k in (("halo", "stencil"), ("allgather", "stencil"), ("halo", "powerlaw"),
                                                   ("allgather", "powerlaw"), ("halo", "powerlaw_nnz"), ("halo", "blockdiag"))]
# the node's own shape: 8 ranks -- uneven slabs (ranks with one and with two neighbours), equal slabs through
# the all-gather, equal nnz shares of skewed rows
CASES += [(8, "halo", "stencil_uneven"), (8, "allgather", "stencil"), (8, "halo", "powerlaw_nnz")]


@pytest.mark.parametrize("world,exchange,kind", CASES)
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
    elif kind == "stencil_uneven":
        rows, cols, rp, cj, va = G.stencil27(5, 4, 53)
    elif kind == "blockdiag":
        blocks = [G.band(50, 3, seed=r + 1) for r in range(world)]
        rows = cols = 50 * world
        cj = torch.cat([b[3] + 50 * r for r, b in enumerate(blocks)])
        va = torch.cat([b[4] for b in blocks])
        counts = torch.cat([(b[2][1:] - b[2][:-1]) for b in blocks]).to(torch.int64)
        rp = torch.zeros(rows + 1, dtype=torch.int64)
        rp[1:] = torch.cumsum(counts, 0)
        rp = rp.to(torch.int32)
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
        if exchange == "halo" and kind in ("stencil", "stencil_uneven"):
            planes = 2 if 0 < rank < world - 1 else 1
            assert nhalo == planes * (7 * 6 if kind == "stencil" else 5 * 4)     # only the neighbouring planes travel
    assert got.tobytes() == want.tobytes()
    assert got2.tobytes() == want2.tobytes()


def test_partition_of_the_530_cube_into_8_blocks():
    """BASELINE config 4's matrix (27-point stencil 530^3: 148 877 000 rows, 1588^3 = 4 004 529 472 non-zeros, more
    than an int counts) never exists in one piece; its row pointers follow from the grid: a row has
    3 - [on a face] entries per dimension.  spmv_hip_partition_rows64 and row_blocks_by_nnz must cut them alike:
    equal shares of the non-zeros (reference rule: csr_bal.c:83-85), boundaries on multiples of 1024 rows."""
    import ctypes as C
    import spmv_amd as S
    from spmv_amd.distributed import row_blocks_by_nnz
    n = 530
    per_dim = np.full(n, 3, np.int64)
    per_dim[0] = per_dim[-1] = 2
    lens = (per_dim[:, None, None] * per_dim[None, :, None] * per_dim[None, None, :]).reshape(-1)     # z slowest
    rp = np.zeros(n ** 3 + 1, np.int64)
    np.cumsum(lens, out=rp[1:])
    del lens
    assert rp[-1] == (3 * n - 2) ** 3 == 4_004_529_472
    bounds = np.zeros(9, np.int32)
    assert S.lib().spmv_hip_partition_rows64(n ** 3, rp.ctypes.data_as(C.c_void_p), 8, 1024, bounds.ctypes.data_as(C.c_void_p)) == 0
    assert bounds.tolist() == row_blocks_by_nnz(torch.from_numpy(rp), 8, 1024)
    assert bounds[0] == 0 and bounds[-1] == n ** 3 and all(b % 1024 == 0 for b in bounds[1:-1])
    shares = np.diff(rp[bounds.astype(np.int64)])
    assert shares.max() < 2 ** 31 - 4096                                    # every block fits the 32-bit positions of one GPU
    assert np.abs(shares - rp[-1] / 8).max() <= 1024 * 27                   # off by at most one snap of full rows
    # each cut lies within half a plane of the plane boundary an even split by planes (66 or 67 each) would use,
    # a little outward of it for the first and last blocks, whose outer faces hold shorter rows
    plane = n * n
    assert all(abs(int(b) - plane * (530 * k // 8)) < plane for k, b in enumerate(bounds))
