#!/usr/bin/env python3
"""Headline benchmark: SpMV GFLOP/s + achieved HBM GB/s of the HIP engine.

    python bench.py --gpus N --steps K --warmup W

N = 1: the headline configuration -- SELL-C-sigma (C=64, sigma=1024) fp64 on the
synthetic 3-D 27-point stencil 252^3 (16 003 008 rows, 428 661 064 nnz), matrix
and vectors resident in HBM before the timed region.  A step is one y = A*x.
N > 1 (weak scaling): every rank owns a z-slab of 252 planes of a 252 x 252 x
(252*N) stencil; a step is the halo exchange of x plus the local product.

One JSON line on stdout (rank 0).  `roofline.achieved` uses the ALGORITHMIC
bytes nnz*(V+4) + rows*(4+V) + cols*V of one launch divided by the mean HIP-event
duration of the launches in the timed region; peak is 8 TB/s (MI355X HBM3E).
`cpu_baseline` times the compiled reference (oracle/_ref, kind "reference") or the
oracle port on the host cores for a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--format", default="sell", choices=["sell", "sellp", "ell", "csr_scalar", "csr_vector",
                                                           "csr_merge", "csr_bal"])
    ap.add_argument("--matrix", default="stencil27", choices=["stencil27", "laplacian2d", "rmat"])
    ap.add_argument("--size", dest="n", type=int, default=0, help="grid edge (stencil27: 252, laplacian2d: 1000) or R-MAT scale")
    ap.add_argument("--planes", type=int, default=0, help="N > 1: z-planes per GPU (default: the grid edge)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--sigma", type=int, default=1024)
    ap.add_argument("--p0", type=int, default=0, help="format parameter (lanes per row / tile size / slice height)")
    ap.add_argument("--exchange", default="halo", choices=["halo", "allgather"])
    ap.add_argument("--comm-breakdown", action="store_true",
                    help="N > 1: after the timed region also time the exchange alone and the local products alone")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=100, help="stencil edge of the CPU-baseline sample")
    return ap.parse_args()


def algorithmic_bytes(rows, cols, nnz, vbytes):
    return nnz * (vbytes + 4) + rows * (4 + vbytes) + cols * vbytes


def cpu_baseline(cpu_n: int):
    """Reference CPU kernels on the host cores, bounded sample: 27-point stencil cpu_n^3,
    the driver's own timing rule (>= 10 iterations and > 2 s, only mult timed; reference
    src/driver_spmv.c:292-308)."""
    import numpy as np
    import oracle
    from spmv_amd import generators as G

    rows, cols, rp, cj, va = G.stencil27(cpu_n)
    rp, cj, va = rp.numpy(), cj.numpy(), va.numpy()
    nnz = len(cj)
    x0 = oracle.aligned_empty(cols)
    x0[:] = 1.0 / np.sqrt(cols)
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)

    def power_loop(mult, min_seconds=2.0):
        x = oracle.aligned_empty(cols)          # the reference's AVX2 kernels need 64-byte aligned vectors
        x[:] = x0
        y = oracle.aligned_empty(rows + 64, fill=0.0)
        spent, it, t0 = 0.0, 0, time.perf_counter()
        while True:
            it += 1
            t1 = time.perf_counter(); mult(x, y); t2 = time.perf_counter()
            spent += t2 - t1
            nrm = float(np.sqrt(np.dot(y[:rows], y[:rows])))
            x[:] = y[:rows] / nrm
            if t2 - t0 > min_seconds and it >= 10:
                break
        return 2.0 * nnz * it / spent / 1e9

    results = {}
    if oracle.ref_available():
        kind = "reference"
        R = oracle.ref()
        coo = G.csr_to_coo_numpy(rows, *map(__import__("torch").from_numpy, (rp, cj, va)))
        csr = R.ref_create_csr(rows, cols, nnz, 1, coo)
        results[("csr_base", 1)] = power_loop(lambda x, y: R.ref_mult_csr_base(csr, x, y))
        sell = R.ref_create_sell(csr, 64, 1024)

        def sell_mult(x, y):
            y[:] = 0.0
            R.ref_mult_sell(sell, x, y)
        if rows % 64 == 0:
            results[("sellcs_mv", 1)] = power_loop(sell_mult)
        mrg = R.ref_create_csr_merge(threads, rows, cols, nnz, coo)
        results[("csr_merge", threads)] = power_loop(lambda x, y: R.ref_mult_csr_merge(mrg, x, y))
    else:
        kind = "port"
        L = oracle.lib()
        results[("csr_base", 1)] = power_loop(lambda x, y: L.orc_mult_csr_base(rows, rp, cj, va, x, y))
        results[("csr_omp", threads)] = power_loop(lambda x, y: L.orc_mult_csr_omp(rows, rp, cj, va, x, y))
    (best_name, best_cores), best = max(results.items(), key=lambda kv: kv[1])
    return {
        "value": round(best, 3), "unit": "GFLOP/s", "cores": best_cores, "kind": kind,
        "kernel": best_name,
        "all": {f"{k}@{c}": round(v, 3) for (k, c), v in results.items()},
        "sample": f"27-point stencil {cpu_n}^3 ({rows} rows, {nnz} nnz), fp64, driver timing rule "
                  f"(>=10 its, >2 s, mult only)",
    }


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    import spmv_amd as S
    from spmv_amd import generators as G
    from spmv_amd.matrix import Matrix

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    # SPMV_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    # (ranks share devices, halo staged through the host); the driver's runs use nccl = RCCL.
    backend = os.environ.get("SPMV_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    S.lib().spmv_hip_init(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    dtype = S.F64 if args.dtype == "f64" else S.F32
    tdtype = torch.float64 if dtype == S.F64 else torch.float32
    vbytes = 8 if dtype == S.F64 else 4
    fmt = {"sell": S.SELL, "sellp": S.SELLP, "ell": S.ELL, "csr_scalar": S.CSR_SCALAR, "csr_vector": S.CSR_VECTOR,
           "csr_merge": S.CSR_MERGE, "csr_bal": S.CSR_BAL}[args.format]
    p0 = args.p0 or (64 if fmt == S.SELL else 0)
    p1 = args.sigma if fmt == S.SELL else 0
    stream = torch.cuda.current_stream().cuda_stream

    # ---- workload -----------------------------------------------------------
    if args.matrix == "stencil27":
        n = args.n or 252
        if world == 1:
            rows, cols, rp, cj, va = G.stencil27(n, device=dev)
            workload = f"stencil27 {n}^3"
        else:
            plane = n * n
            nzp = args.planes or n          # 530^3 over 8 GPUs: --size 530 --planes 66
            r0, r1 = rank * nzp * plane, (rank + 1) * nzp * plane
            rows, cols, rp, cj, va = G.stencil27(n, n, nzp * world, device=dev, row_begin=r0, row_end=r1)
            workload = f"stencil27 {n}x{n}x{nzp * world}, z-slab of {nzp} planes per GPU"
    elif args.matrix == "laplacian2d":
        n = args.n or 1000
        rows, cols, rp, cj, va = G.laplacian2d(n, device=dev)
        workload = f"laplacian2d {n}^2"
    else:
        n = args.n or 22
        rows, cols, rp, cj, va = G.rmat(n, 16, device=dev)
        workload = f"rmat scale {n} edge factor 16"
    if world > 1 and args.matrix != "stencil27":
        sys.exit("multi-GPU bench is defined on the stencil27 workload")
    nnz = int(rp[-1])
    va = va.to(tdtype)

    if world == 1:
        m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), dtype, p0, p1,
                                   stream, keepalive=(rp, cj, va))
        m.analyze(3)           # kernel variant by trial, as create_sell does (reference sell.c:30)
        if fmt in (S.SELL, S.SELLP, S.ELL):
            m._keepalive = None
            del rp, cj, va
            torch.cuda.empty_cache()
        g = torch.Generator(device=dev).manual_seed(1234)
        x = torch.rand(cols, dtype=tdtype, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x)
        y = torch.empty(rows, dtype=tdtype, device=dev)

        def step():
            m.mult_dev(x.data_ptr(), y.data_ptr(), stream)
        info = m.info
        stored, variant = info.stored, info.variant
        global_rows, global_nnz, global_cols = rows, nnz, cols
    else:
        from spmv_amd.distributed import RowBlockSpmv
        op = RowBlockSpmv(fmt, rows, cols, rp, cj, va, dtype, p0, p1, row_begin=rank * rows, exchange=args.exchange,
                          device=dev)
        del rp, cj, va
        torch.cuda.empty_cache()
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        x = torch.rand(rows, dtype=tdtype, device=dev, generator=g)
        y = torch.empty(rows, dtype=tdtype, device=dev)
        if op.A_glob is None:                 # halo exchange: x lives in the operator's extended vector,
            op.x_local.copy_(x)               # so a step starts from it without a 128 MB copy
            x = op.x_local

        def step():
            op.mult(x, y)
        stored, variant = op.stored, op.variant
        global_rows, global_nnz, global_cols = rows * world, None, cols
        t = torch.tensor([nnz], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        global_nnz = int(t.item())

    # ---- timed region -------------------------------------------------------
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        starts[k].record()
        step()
        ends[k].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = sorted(s.elapsed_time(e) for s, e in zip(starts, ends))
    mean_kernel_ms = sum(kernel_ms) / len(kernel_ms)

    # outside the timed region: the exchange alone and the local products alone, to say how much of
    # the communication the overlap hides (max over ranks of per-step means)
    comm = None
    if world > 1 and args.comm_breakdown:
        def timed_ms(fn, reps=10):
            fn()
            torch.cuda.synchronize()
            dist.barrier()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            t = torch.tensor([a.elapsed_time(b) / reps], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        exchange_ms = timed_ms(op.exchange_only)
        compute_ms = timed_ms(lambda: op.compute_only(y))
        comm = {"exchange_alone_ms": round(exchange_ms, 5), "local_products_alone_ms": round(compute_ms, 5),
                "halo_entries_rank0": int(op.nhalo)}

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        gflops = 2.0 * global_nnz / (ms_per_step * 1e-3) / 1e9
        b_alg = algorithmic_bytes(rows, cols if world == 1 else rows, nnz, vbytes)   # one launch, one GPU
        achieved = b_alg / (mean_kernel_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        key = f"{args.format}/{args.matrix}/{n}/{args.dtype}/gpus{world}/variant{variant}"   # PMC figure of that kernel
        if os.path.exists(tfile):
            traffic = json.load(open(tfile)).get(key)
        out = {
            "metric": "SpMV GFLOP/s + achieved HBM GB/s vs peak, CSR & SELL-C-σ, 1/2/4/8 GPUs",
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload, "format": args.format,
                       "params": {"C": 64, "sigma": args.sigma} if fmt == S.SELL else {"p0": p0},
                       "rows": global_rows, "nnz": global_nnz, "stored_per_gpu": int(stored), "kernel_variant": variant,
                       "exchange": args.exchange if world > 1 else None},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": b_alg, "kernel_ms_mean": round(mean_kernel_ms, 5),
                         "kernel_ms_min": round(kernel_ms[0], 5), "kernel_ms_median": round(kernel_ms[len(kernel_ms) // 2], 5)},
        }
        if comm is not None:
            comm["exposed_ms"] = round(max(ms_per_step - comm["local_products_alone_ms"], 0.0), 5)
            out["communication"] = comm
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_n)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
