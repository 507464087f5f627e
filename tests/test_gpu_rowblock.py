"""The row-block partitioned product behind the C ABI (spmv_hip_partition_rows, spmv_hip_rowblock_*,
spmv_hip_dist_*): the split built in HBM against a numpy restatement, and whole products of 1, 2, 3,
5 and 8 blocks -- all on this box's one GPU, which runs the same pack / peer copy / interior / boundary
sequence as distinct devices do -- against the oracle's single-process mult_csr_base.

The reference has no partitioned product (its precedent: the equal-nnz chunks of create_csr_bal,
reference src/csr_bal.c:83-85), so parity is anchored on mult_csr_base (csr.c:4-13): every row is
still summed whole, by one kernel.  Bar: rel-L2 <= 1e-13 (the interior kernel trial may pick a
reassociating variant), bit-exact for lane-per-row CSR.
"""
import ctypes as C

import numpy as np
import pytest
import torch

import spmv_amd as S
from spmv_amd import _lib as L
from spmv_amd import generators as G
from spmv_amd.distributed import HipEngine, row_blocks_by_nnz, slice_row_block

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _matrices():
    return {"stencil": G.stencil27(10, 11, 40), "powerlaw": G.power_law_rows(5000, seed=4), "lap2d": G.laplacian2d(90),
            "rmat": G.rmat(12, 8)}


def test_partition_rows_is_the_equal_share_rule():
    rows, cols, rp, cj, va = G.rmat(14, 8, seed=5)
    rp_n = np.ascontiguousarray(rp.numpy(), np.int32)
    for parts, snap in ((2, 64), (4, 64), (8, 1024), (3, 1)):
        b = np.zeros(parts + 1, np.int32)
        assert S.lib().spmv_hip_partition_rows(rows, rp_n.ctypes.data_as(C.c_void_p), parts, snap,
                                               b.ctypes.data_as(C.c_void_p)) == 0
        assert b.tolist() == row_blocks_by_nnz(rp, parts, snap)
    rows, cols, rp, cj, va = G.stencil27(8, 8, 32)
    b = np.zeros(3, np.int32)
    rp_n = np.ascontiguousarray(rp.numpy(), np.int32)
    S.lib().spmv_hip_partition_rows(rows, rp_n.ctypes.data_as(C.c_void_p), 2, 64, b.ctypes.data_as(C.c_void_p))
    assert b.tolist() == [0, 1024, 2048]


@pytest.mark.parametrize("name", ["stencil", "powerlaw", "lap2d", "rmat"])
@pytest.mark.parametrize("fmt", [S.SELL, S.CSR_SCALAR, S.CSR_MERGE])
def test_rowblock_split_and_products(orc, name, fmt):
    rows, cols, rp, cj, va = _matrices()[name]
    bounds = row_blocks_by_nnz(rp, 3, snap=64)
    x = np.random.default_rng(1).random(cols)
    want = orc.mult_csr_base(rp.numpy(), cj.numpy(), va.numpy(), x)
    eng = HipEngine(S.F64)
    for p in range(3):
        r0, r1 = bounds[p], bounds[p + 1]
        n = r1 - r0
        rp_b, cj_b, va_b = slice_row_block(rp, cj, va, r0, r1)
        blk = eng.rowblock(fmt, n, r0, cols, rp_b.to(DEV), cj_b.to(DEV), va_b.to(DEV), 64 if fmt == S.SELL else 0,
                           1024 if fmt == S.SELL else 0)
        cjn, rpn = cj_b.numpy().astype(np.int64), rp_b.numpy().astype(np.int64)
        remote = (cjn < r0) | (cjn >= r1)
        row_of = np.repeat(np.arange(n), np.diff(rpn))
        bnd = np.zeros(n, bool)
        bnd[row_of[remote]] = True
        halo = np.unique(cjn[remote])
        info = blk.info
        assert (info.rows, info.row_begin, info.columns_global, info.nnz) == (n, r0, cols, len(cjn))
        assert info.halo == len(halo) and np.array_equal(blk.halo_columns.numpy(), halo)
        assert info.boundary_rows == int(bnd.sum())
        assert info.boundary_nnz == int(np.diff(rpn)[bnd].sum()) and info.interior_nnz == len(cjn) - info.boundary_nnz
        # products: x_ext = [x_local | halo values]; a send list of this block's own rows exercises pack
        if fmt == S.SELL:
            blk.set_option("exact_order", 1)
        x_ext = torch.from_numpy(np.concatenate([x[r0:r1], x[halo]])).to(DEV)
        y = torch.full((n,), float("nan"), dtype=torch.float64, device=DEV)
        blk.mult_interior(x_ext, y)
        blk.mult_boundary(x_ext, y)
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        if fmt in (S.CSR_SCALAR, S.SELL):            # lane-per-row everywhere: the bits of mult_csr_base
            assert got.tobytes() == want[r0:r1].tobytes()
        else:
            assert orc.rel_l2(got, want[r0:r1]) <= 1e-13 if n else True
        idx = torch.arange(0, n, 3)
        blk.set_send_list(idx)
        send = torch.zeros(len(idx), dtype=torch.float64, device=DEV)
        blk.pack(x_ext, send)
        torch.cuda.synchronize()
        assert np.array_equal(send.cpu().numpy(), x[r0:r1][::3])
        with pytest.raises(S.SpmvHipError, match="not a local row"):
            blk.set_send_list(torch.tensor([n]))
        blk.free()


def _dist(fmt, rows, cols, coo, dtype, p0, p1, ngpus):
    dev = (C.c_int * ngpus)(*([0] * ngpus))
    return L.check_handle(S.lib().spmv_hip_dist_create(fmt, rows, cols, len(coo), coo.ctypes.data_as(C.c_void_p), dtype, p0,
                                                       p1, ngpus, dev))


@pytest.mark.parametrize("ngpus", [1, 2, 3, 5, 8])
@pytest.mark.parametrize("name", ["stencil", "powerlaw", "rmat"])
def test_dist_product_of_n_blocks_matches_the_oracle(orc, name, ngpus):
    rows, cols, rp, cj, va = _matrices()[name]
    coo_sorted = G.csr_to_coo_numpy(rows, rp, cj, va)
    x = np.random.default_rng(3).random(cols)
    want = orc.mult_csr_base(rp.numpy(), cj.numpy(), va.numpy(), x)
    lib = S.lib()
    for fmt, p0, p1, dtype in ((S.SELL, 64, 1024, S.F64), (S.CSR_SCALAR, 0, 0, S.F64), (S.CSR_BAL, 0, 0, S.F64),
                               (S.SELLP, 0, 0, S.F32)):
        coo = coo_sorted[np.random.default_rng(5).permutation(len(coo_sorted))].copy()    # the creator sorts in place
        d = _dist(fmt, rows, cols, coo, dtype, p0, p1, ngpus)
        assert np.array_equal(coo, coo_sorted)
        n = C.c_int()
        bounds = np.zeros(ngpus + 1, np.int32)
        blocks = (L.RowBlockInfo * ngpus)()
        assert lib.spmv_hip_dist_get_info(d, C.byref(n), bounds.ctypes.data_as(C.c_void_p), blocks) == 0
        assert n.value == ngpus and bounds[0] == 0 and bounds[-1] == rows
        assert bounds.tolist() == row_blocks_by_nnz(rp, ngpus, 1024)
        assert sum(b.nnz for b in blocks) == len(coo) and all(b.rows == bounds[k + 1] - bounds[k] for k, b in enumerate(blocks))
        access = np.full(ngpus * ngpus, -1, np.int32)
        assert lib.spmv_hip_dist_get_peer_access(d, access.ctypes.data_as(C.c_void_p)) == 0
        assert (access == 2).all()                      # every block sits on device 0
        xin = x.astype(np.float32).astype(np.float64) if dtype == S.F32 else x
        for scale in (1.0, -2.5):                       # twice: the exchange buffers and streams are reused
            if scale != 1.0:                            # first on the default kernels, then on what the trial picks
                L.check(lib.spmv_hip_dist_analyze(d, 2))
            y = np.full(rows, np.nan)
            xs = np.ascontiguousarray(xin * scale)
            L.check(lib.spmv_hip_dist_mult(d, xs.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)))
            ref = orc.mult_csr_base(rp.numpy(), cj.numpy(), va.numpy(), xs)
            if dtype == S.F32:
                assert orc.rel_l2(y, ref) <= 1e-5
            elif fmt == S.CSR_SCALAR:
                assert y.tobytes() == ref.tobytes()
            else:
                assert orc.rel_l2(y, ref) <= 1e-13
        lib.spmv_hip_dist_free(d)
    assert want is not None


def test_dist_bench_is_the_drivers_power_iteration(orc):
    """spmv_hip_dist_bench against a replay of driver_spmv.c:292-305 on the oracle."""
    rows, cols, rp, cj, va = G.stencil27(12, 12, 24)
    coo = G.csr_to_coo_numpy(rows, rp, cj, va)
    d = _dist(S.CSR_SCALAR, rows, cols, coo, S.F64, 0, 0, 3)
    x0 = np.random.default_rng(8).random(cols)
    x0 /= np.linalg.norm(x0)
    x = x0.copy()
    it, sec = C.c_int(), C.c_double()
    L.check(S.lib().spmv_hip_dist_bench(d, 7, 0.0, x.ctypes.data_as(C.c_void_p), C.byref(it), C.byref(sec)))
    assert it.value >= 7 and sec.value > 0
    xr = x0.copy()
    for _ in range(it.value):
        y = orc.mult_csr_base(rp.numpy(), cj.numpy(), va.numpy(), xr)
        xr = y / np.sqrt(np.dot(y, y))
    assert np.max(np.abs(x - xr)) <= 1e-12 * np.max(np.abs(xr))
    S.lib().spmv_hip_dist_free(d)


def test_dist_rejects_bad_arguments():
    lib = S.lib()
    rows, cols, rp, cj, va = G.laplacian2d(10)
    coo = G.csr_to_coo_numpy(rows, rp, cj, va)
    one = (C.c_int * 1)(0)
    assert not lib.spmv_hip_dist_create(S.SELL, rows, cols + 1, len(coo), coo.ctypes.data_as(C.c_void_p), S.F64, 64, 1024, 1, one)
    assert b"square" in lib.spmv_hip_last_error()
    far = (C.c_int * 1)(63)
    assert not lib.spmv_hip_dist_create(S.SELL, rows, cols, len(coo), coo.ctypes.data_as(C.c_void_p), S.F64, 64, 1024, 1, far)
    assert b"does not exist" in lib.spmv_hip_last_error()
    assert not lib.spmv_hip_dist_create(S.SELL, rows, cols, len(coo), coo.ctypes.data_as(C.c_void_p), S.F64, 64, 1024, 0, None)


def _device_blocks(fmt, rows, cols, rp, cj, va, dtype, p0, p1, bounds):
    """One spmv_hip_rowblock per row range of `bounds`, each from its CSR shard in HBM (global columns)."""
    td = torch.float64 if dtype == S.F64 else torch.float32
    out = []
    for r0, r1 in zip(bounds, bounds[1:]):
        rp_b, cj_b, va_b = slice_row_block(rp, cj, va, r0, r1)
        rp_d, cj_d, va_d = rp_b.to(DEV), cj_b.to(DEV), va_b.to(td).to(DEV)
        if cj_d.numel() == 0:
            cj_d, va_d = torch.zeros(1, dtype=torch.int32, device=DEV), torch.zeros(1, dtype=td, device=DEV)
        out.append(L.check_handle(S.lib().spmv_hip_rowblock_create(fmt, r1 - r0, r0, cols, int(rp_b[-1]), rp_d.data_ptr(),
                                                                    cj_d.data_ptr(), va_d.data_ptr(), dtype, p0, p1, 0)))
        torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("fmt,p0,p1", [(S.SELL, 64, 1024), (S.CSR_SCALAR, 0, 0)])
def test_dist_from_blocks_built_shard_by_shard(orc, fmt, p0, p1):
    """spmv_hip_dist_create_from_blocks: the way a matrix with more non-zeros than an int counts reaches the
    partitioned product -- no global COO, every block from its own CSR shard in HBM.  8 uneven blocks."""
    lib = S.lib()
    rows, cols, rp, cj, va = G.stencil27(12, 11, 53)
    plane = 12 * 11
    bounds = [plane * (53 * r // 8) for r in range(9)]                  # slabs of 6 or 7 planes, the 66 / 67 shape of 530 planes
    blocks = _device_blocks(fmt, rows, cols, rp, cj, va, S.F64, p0, p1, bounds)
    arr = (C.c_void_p * 8)(*blocks)
    d = L.check_handle(lib.spmv_hip_dist_create_from_blocks(8, arr))
    L.check(lib.spmv_hip_dist_set_option(d, b"exact_order", 1))
    L.check(lib.spmv_hip_dist_analyze(d, 2))
    got_bounds = np.zeros(9, np.int32)
    info = (L.RowBlockInfo * 8)()
    assert lib.spmv_hip_dist_get_info(d, None, got_bounds.ctypes.data_as(C.c_void_p), info) == 0
    assert got_bounds.tolist() == bounds
    assert [b.halo for b in info] == [plane] + [2 * plane] * 6 + [plane]     # the neighbouring planes only
    x = np.random.default_rng(17).random(cols)
    want = orc.mult_csr_base(rp.numpy(), cj.numpy(), va.numpy(), x)
    y = np.full(rows, np.nan)
    L.check(lib.spmv_hip_dist_mult(d, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)))
    assert y.tobytes() == want.tobytes()                                # lane-per-row everywhere: mult_csr_base's bits
    lib.spmv_hip_dist_free(d)                                           # frees the blocks too


def test_dist_from_blocks_refuses_blocks_that_do_not_tile_the_matrix():
    lib = S.lib()
    rows, cols, rp, cj, va = G.laplacian2d(40)
    blocks = _device_blocks(S.CSR_SCALAR, rows, cols, rp, cj, va, S.F64, 0, 0, [0, 640, 1600])
    swapped = (C.c_void_p * 2)(blocks[1], blocks[0])
    assert not lib.spmv_hip_dist_create_from_blocks(2, swapped)
    assert b"does not continue" in lib.spmv_hip_last_error()
    first_only = (C.c_void_p * 1)(blocks[0])
    assert not lib.spmv_hip_dist_create_from_blocks(1, first_only)
    assert b"end at row 640 of 1600" in lib.spmv_hip_last_error()
    for b in blocks:                                                    # after a refusal the blocks are still the caller's
        assert lib.spmv_hip_rowblock_free(b) == 0


@pytest.mark.parametrize("threads,direct", [("1", "1"), ("0", "1"), ("1", "0")])
@pytest.mark.parametrize("fmt,p0", [(S.CSR_MERGE, 0), (S.CSR_VECTOR, 0), (S.SELL, 64)])
def test_exact_order_pins_the_bits_of_a_partitioned_product(orc, fmt, p0, threads, direct, monkeypatch):
    """"exact_order" set before the first kernel trial: whatever the trials measure, y has the bits of the
    handles' default kernels -- over repeated create / analyze cycles, with the launches issued by the helper
    threads (default) and by the calling thread (SPMV_DIST_THREADS=0), the halo stored straight into the
    consumers' vectors (default) and packed, then copied (SPMV_DIST_DIRECT_PUSH=0)."""
    monkeypatch.setenv("SPMV_DIST_THREADS", threads)
    monkeypatch.setenv("SPMV_DIST_DIRECT_PUSH", direct)
    lib = S.lib()
    rows, cols, rp, cj, va = G.power_law_rows(6000, seed=11)
    coo_sorted = G.csr_to_coo_numpy(rows, rp, cj, va)
    x = np.random.default_rng(2).random(cols)
    want = orc.mult_csr_base(rp.numpy(), cj.numpy(), va.numpy(), x)
    seen = set()
    for cycle in range(3):
        d = _dist(fmt, rows, cols, coo_sorted.copy(), S.F64, p0, 1024 if fmt == S.SELL else 0, 3)
        y0 = np.full(rows, np.nan)
        L.check(lib.spmv_hip_dist_mult(d, x.ctypes.data_as(C.c_void_p), y0.ctypes.data_as(C.c_void_p)))   # default kernels
        L.check(lib.spmv_hip_dist_set_option(d, b"exact_order", 1))
        L.check(lib.spmv_hip_dist_analyze(d, 1 + cycle))
        y = np.full(rows, np.nan)
        L.check(lib.spmv_hip_dist_mult(d, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)))
        assert y.tobytes() == y0.tobytes()
        assert orc.rel_l2(y, want) <= 1e-13
        seen.add(y.tobytes())
        lib.spmv_hip_dist_free(d)
    assert len(seen) == 1
