"""Row-block SpMV with the HIP engine: two ranks rehearsed on ONE GPU.

The GPU box has a single MI355X, so both ranks use cuda:0 and talk over gloo
(halo staged through the host); the local products, the interior/boundary split,
the row-mapped boundary kernel and the stream ordering are the real ones.  A
multi-rank RCCL run needs one GPU per rank.  Tolerance: rel-L2 <=
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


ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _run_bench(extra_env, *argv, timeout=400):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SPMV_BENCH_BACKEND")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_launches_its_own_ranks_gloo_rehearsal():
    """`python bench.py --gpus 2` by itself: two ranks on this box's one GPU over gloo, one valid line."""
    import json
    p = _run_bench({"SPMV_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--size", "96", "--steps", "5", "--warmup", "2")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 5 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["config"]["rows"] == 2 * 96 ** 3
    comm = rec["communication"]                      # on by default for N > 1
    assert comm["halo_entries_rank0"] == 96 * 96 and comm["exchange_alone_ms"] > 0 and "exposed_ms" in comm


def test_bench_uneven_slabs_of_a_fixed_grid():
    """--total-planes deals a fixed grid over the ranks (the shape of BASELINE config 4, 530^3 over 8 GPUs):
    41 planes over 2 ranks = slabs of 20 and 21 planes, halo exchange between unequal row blocks."""
    import json
    p = _run_bench({"SPMV_BENCH_BACKEND": "gloo"}, "--gpus", "2", "--size", "64", "--total-planes", "41", "--steps", "3",
                   "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert rec["config"]["rows"] == 64 * 64 * 41 and rec["scaling"] == "strong"
    assert rec["config"]["nnz"] == (3 * 64 - 2) ** 2 * (3 * 41 - 2)
    assert rec["communication"]["halo_entries_rank0"] == 64 * 64


def test_bench_refuses_more_rccl_ranks_than_gpus():
    have = torch.cuda.device_count()
    p = _run_bench({}, "--gpus", str(have + 1), "--steps", "2")
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert f"has {have}" in p.stderr


def _rccl_world1(port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("nccl", device_id=dev)
    try:
        a = torch.arange(1000, dtype=torch.float64, device=dev)
        full = torch.empty(1000, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(full, a)
        s = torch.sum(a * a).reshape(1)
        dist.all_reduce(s)
        recv = torch.zeros(64, dtype=torch.float64, device=dev)
        ops = [dist.P2POp(dist.isend, a[100:164].contiguous(), 0), dist.P2POp(dist.irecv, recv, 0)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        torch.cuda.synchronize()
        out_q.put((bool(torch.equal(full, a)), float(s), bool(torch.equal(recv, a[100:164]))))
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_of_the_row_block_product_run_on_this_gpu():
    """The three RCCL calls of spmv_amd.distributed (all_gather_into_tensor, all_reduce, one group of
    isend/irecv on device tensors) on a one-rank communicator: what one GPU can run of the RCCL path."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_world1, args=(_free_port(), q))
    p.start()
    gathered, sumsq, p2p = q.get(timeout=110)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert gathered and p2p and sumsq == float(sum(i * i for i in range(1000)))


def test_bench_eight_ranks_uneven_slabs_and_the_one_process_leg():
    """`python bench.py --gpus 8` end to end on this box's one GPU (gloo): 67 planes over 8 ranks = slabs of 8 or 9
    planes (the 66 / 67 shape of 530 planes over 8 GPUs), ranks with one neighbour and ranks with two; then the same
    slabs as 8 blocks of one process through spmv_hip_dist_*, whose product of x_i = i must be exactly 0.0 on
    every row away from the grid's faces."""
    import json
    p = _run_bench({"SPMV_BENCH_BACKEND": "gloo"}, "--gpus", "8", "--size", "48", "--total-planes", "67", "--steps", "3",
                   "--warmup", "1")
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert rec["n_gpus"] == 8 and rec["scaling"] == "strong" and rec["value"] > 0
    assert rec["config"]["rows"] == 48 * 48 * 67 and rec["config"]["nnz"] == (3 * 48 - 2) ** 2 * (3 * 67 - 2)
    assert rec["rccl"]["world"] == 8 and rec["rccl"]["ranks_seen"] == 8
    assert rec["communication"]["halo_entries_rank0"] == 48 * 48
    one = rec["c_abi_one_process"]
    assert "error" not in one, one
    assert one["blocks"] == 8 and one["nnz"] == rec["config"]["nnz"]
    assert one["halo_per_block"] == [48 * 48] + [2 * 48 * 48] * 6 + [48 * 48]
    assert one["halo_check"]["ok"] and one["halo_check"]["rows_wrong"] == 0
