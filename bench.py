#!/usr/bin/env python3
"""Headline benchmark: SpMV GFLOP/s + achieved HBM GB/s of the HIP engine.

    python bench.py --gpus N --steps K --warmup W

N = 1: the headline configuration -- SELL-C-sigma (C=64, sigma=1024) fp64 on the
synthetic 3-D 27-point stencil 252^3 (16 003 008 rows, 428 661 064 nnz), matrix
and vectors resident in HBM before the timed region.  A step is one y = A*x.
N > 1 (weak scaling): every rank owns a z-slab of 252 planes of a 252 x 252 x
(252*N) stencil; a step is the halo exchange of x plus the local product.

Started without torch.distributed.run's environment and with N > 1, the script
launches its own N ranks (python -m torch.distributed.run on 127.0.0.1) before
anything touches a GPU, relays rank 0's line and exits with the ranks' status.

One JSON line on stdout (rank 0).  `roofline.achieved` uses the ALGORITHMIC
bytes nnz*(V+4) + rows*(4+V) + cols*V of one launch divided by the mean HIP-event
duration of the launches in the timed region; peak is 8 TB/s (MI355X HBM3E).
`roofline.traffic` is the HBM traffic of one launch from the PMC passes recorded
in profiles/hbm_traffic.json for exactly this kernel source (null otherwise),
`frac_moved` the fraction of peak those bytes amount to; `roofline.general` is the
same measurement for the fastest kernel variant that reads every column index.
`cpu_baseline` times the compiled reference (oracle/_ref, kind "reference") or the
oracle port on the host cores for a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0
METRIC = "SpMV GFLOP/s + achieved HBM GB/s vs peak, CSR & SELL-C-σ, 1/2/4/8 GPUs"
KERNEL_SOURCES = ("spmv_amd/csrc/kernels.hip", "spmv_amd/csrc/device_utils.h", "spmv_amd/csrc/internal.h")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--format", default="sell", choices=["sell", "sellp", "ell", "csr_scalar", "csr_vector",
                                                           "csr_merge", "csr_bal"])
    ap.add_argument("--matrix", default="stencil27", choices=["stencil27", "laplacian2d", "rmat"])
    ap.add_argument("--size", dest="n", type=int, default=0, help="grid edge (stencil27: 252, laplacian2d: 1000) or R-MAT scale")
    ap.add_argument("--planes", type=int, default=0, help="N > 1: z-planes per GPU (default: the grid edge)")
    ap.add_argument("--total-planes", type=int, default=0,
                    help="N > 1: z-planes of the whole grid, dealt over the GPUs as evenly as they go (strong-scaling shape; "
                         "BASELINE config 4 is --size 530 --total-planes 530)")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--sigma", type=int, default=1024)
    ap.add_argument("--p0", type=int, default=0, help="format parameter (lanes per row / tile size / slice height)")
    ap.add_argument("--variant", type=int, default=-1, help="N = 1: this kernel variant instead of the one the trial picks")
    ap.add_argument("--exchange", default="halo", choices=["halo", "allgather"])
    ap.add_argument("--no-comm-breakdown", action="store_true",
                    help="N > 1: skip timing the exchange alone and the local products alone after the timed region")
    ap.add_argument("--no-general", action="store_true", help="N = 1: skip the every-index-read variant's measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=100, help="stencil edge of the CPU-baseline sample")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="run only the host-CPU baseline and print its JSON (what the N = 1 run starts as a child)")
    return ap.parse_args(argv)


def algorithmic_bytes(rows, cols, nnz, vbytes):
    return nnz * (vbytes + 4) + rows * (4 + vbytes) + cols * vbytes


def kernel_source_hash() -> str:
    """sha256 over the kernel sources: a traffic figure belongs to exactly one build of the kernels."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def recorded_traffic(key: str):
    """HBM bytes per launch measured by PMC for `key` with the present kernel sources, else (None, why)."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if not os.path.exists(path):
        return None, "profiles/hbm_traffic.json is missing"
    entry = json.load(open(path)).get(key)
    if entry is None:
        return None, f"no PMC record for {key}"
    if not isinstance(entry, dict) or entry.get("kernel_sources_sha256") != kernel_source_hash():
        return None, f"the PMC record for {key} was taken with other kernel sources"
    return int(entry["bytes"]), entry.get("source", "")


# ------------------------------------------------------------------ CPU baseline --

def host_cpus():
    """Logical CPUs this process may run on, and how many physical cores they are."""
    cpus = sorted(os.sched_getaffinity(0))
    cores, model = set(), ""
    try:
        phys = core = None
        proc = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("processor"):
                proc = int(line.split(":")[1])
            elif line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = int(line.split(":")[1])
            elif line.startswith("core id"):
                core = int(line.split(":")[1])
            elif not line.strip():
                if proc in cpus and phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = proc = None
    except OSError:
        pass
    return cpus, (len(cores) or len(cpus)), model


def l3_domains(cpus):
    """How many last-level caches (on an EPYC host: core complexes, each with its own link to memory) the
    allowed CPUs spread over -- the bandwidth a memory-bound OpenMP kernel can draw scales with this
    number, not with the thread count."""
    groups = set()
    for c in cpus:
        try:
            groups.add(open(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list").read().strip())
        except OSError:
            return None
    return len(groups)


def cpu_baseline_child(cpu_n: int):
    """Reference CPU kernels on the host cores, bounded samples, the driver's own timing rule (>= 10
    iterations and > 2 s, only mult timed; reference src/driver_spmv.c:292-308).  Single-threaded
    kernels run pinned to one CPU like spmv.run:29-31 (OMP_NUM_THREADS=1, taskset -c 0); the OpenMP
    kernels run on the threads the parent chose, bound close to cores.  Prints one JSON object."""
    cpus, physical, model = host_cpus()         # before any import loads an OpenMP runtime: OMP_PROC_BIND
    import numpy as np                          # narrows the main thread's own mask to its place
    import oracle
    from spmv_amd import generators as G
    import torch

    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or len(cpus)
    have_ref = oracle.ref_available()
    R = oracle.ref() if have_ref else None      # loaded under the full affinity mask: libgomp lays out its places now
    L = oracle.lib()

    def power_loop(mult, rows, cols, nnz, min_seconds=2.0):
        x = oracle.aligned_empty(cols)          # the reference's AVX2 kernels need 64-byte aligned vectors
        x[:] = 1.0 / np.sqrt(cols)
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

    def pinned(fn):
        os.sched_setaffinity(0, {cpus[0]})
        try:
            return fn()
        finally:
            os.sched_setaffinity(0, set(cpus))

    def workload(gen):
        rows, cols, rp, cj, va = gen
        rp, cj, va = rp.numpy(), cj.numpy(), va.numpy()
        coo = G.csr_to_coo_numpy(rows, *map(torch.from_numpy, (rp, cj, va)))
        return rows, cols, rp, cj, va, coo

    results, scaling = {}, {}
    rows, cols, rp, cj, va, coo = workload(G.stencil27(cpu_n))
    nnz = len(cj)
    if have_ref:
        kind = "reference"
        csr = R.ref_create_csr(rows, cols, nnz, 1, coo)
        results[("csr_base", 1)] = pinned(lambda: power_loop(lambda x, y: R.ref_mult_csr_base(csr, x, y), rows, cols, nnz))
        if rows % 64 == 0:
            sell = R.ref_create_sell(csr, 64, 1024)

            def sell_mult(x, y):
                y[:] = 0.0                        # sellcs_mv_d_autovector accumulates into y (sellcs_mv_autovector.c:48)
                R.ref_mult_sell(sell, x, y)
            results[("sellcs_mv", 1)] = pinned(lambda: power_loop(sell_mult, rows, cols, nnz))
        t = 1
        while True:                               # csr_merge on 1, 2, 4, ... threads: says what the host's cores add
            mrg = R.ref_create_csr_merge(t, rows, cols, nnz, coo)
            scaling[t] = power_loop(lambda x, y: R.ref_mult_csr_merge(mrg, x, y), rows, cols, nnz,
                                    min_seconds=2.0 if t == threads else 0.5)
            if t == threads:
                break
            t = min(2 * t, threads)
        results[("csr_merge", threads)] = scaling[threads]
        bal = R.ref_create_csr_bal(threads, rows, cols, nnz, coo)
        results[("csr_bal", threads)] = power_loop(lambda x, y: R.ref_mult_csr_bal(bal, x, y), rows, cols, nnz)
    else:
        kind = "port"
        results[("csr_base", 1)] = pinned(lambda: power_loop(lambda x, y: L.orc_mult_csr_base(rows, rp, cj, va, x, y), rows, cols, nnz))
        results[("csr_omp", threads)] = power_loop(lambda x, y: L.orc_mult_csr_omp(rows, rp, cj, va, x, y), rows, cols, nnz)
    sample = (f"27-point stencil {cpu_n}^3 ({rows} rows, {nnz} nnz), fp64, driver timing rule "
              f"(>=10 its, >2 s, mult only)")

    # BASELINE configs[0]: plain CSR (csr.c) fp64 on the 2-D 5-point Laplacian 1000^2, one pinned core
    rows1, cols1, rp1, cj1, va1, coo1 = workload(G.laplacian2d(1000))
    if have_ref:
        csr1 = R.ref_create_csr(rows1, cols1, len(cj1), 1, coo1)
        c1 = pinned(lambda: power_loop(lambda x, y: R.ref_mult_csr_base(csr1, x, y), rows1, cols1, len(cj1)))
    else:
        c1 = pinned(lambda: power_loop(lambda x, y: L.orc_mult_csr_base(rows1, rp1, cj1, va1, x, y), rows1, cols1, len(cj1)))

    (best_name, best_cores), best = max(results.items(), key=lambda kv: kv[1])
    print(json.dumps({
        "value": round(best, 3), "unit": "GFLOP/s", "cores": best_cores, "kind": kind, "kernel": best_name,
        "all": {f"{k}@{c}": round(v, 3) for (k, c), v in results.items()},
        "sample": sample,
        "csr_merge_by_threads": {str(k): round(v, 3) for k, v in scaling.items()},
        "config1_laplacian2d_1000": {"kernel": "csr_base", "cores": 1, "value": round(c1, 3), "unit": "GFLOP/s",
                                     "rows": rows1, "nnz": len(cj1)},
        "host": {"cpu_model": model, "logical_cpus_allowed": len(cpus), "physical_cores_allowed": physical, "l3_domains_allowed": l3_domains(cpus),
                 "omp_threads": threads, "omp_proc_bind": os.environ.get("OMP_PROC_BIND", ""),
                 "omp_places": os.environ.get("OMP_PLACES", ""), "single_thread_runs_pinned_to_cpu": cpus[0]},
    }), flush=True)


def cpu_baseline(cpu_n: int):
    """Runs the CPU baseline in a child process of its own: the OpenMP runtime reads its binding from the
    environment when it is loaded, and the child never touches the GPU."""
    env = dict(os.environ)
    allowed = len(os.sched_getaffinity(0))
    threads = min(int(env.get("OMP_NUM_THREADS", "0")) or allowed, allowed)
    env.update({"OMP_NUM_THREADS": str(threads), "OMP_PROC_BIND": "close", "OMP_PLACES": "cores"})
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-n", str(cpu_n)],
                         env=env, capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        raise RuntimeError(f"cpu baseline failed ({out.returncode}): {out.stderr[-2000:]}")
    return json.loads(out.stdout.strip().splitlines()[-1])


# ---------------------------------------------------------------- rank launcher --

def launch_ranks(args, argv, script=None):
    """python bench.py --gpus N (N > 1) outside torch.distributed.run: start the N ranks, relay rank 0's
    JSON line, return the exit status.  Nothing here initialises a GPU (device_count only counts).
    `script` (default: this file) is what every rank runs."""
    backend = os.environ.get("SPMV_BENCH_BACKEND", "nccl")
    if backend == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} GPUs, this node has {have} "
                  f"(SPMV_BENCH_BACKEND=gloo rehearses the ranks on fewer devices)", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print(f"bench.py: the ranks exited with status {proc.returncode}", file=sys.stderr)
        return proc.returncode
    if line is None:
        print("bench.py: the ranks printed no result line", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


# ------------------------------------------------------------------------- main --

def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.cpu_baseline_only:
        cpu_baseline_child(args.cpu_n)
        return 0
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args, argv)
    world = int(env_world or "1")
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE is {world} but --gpus is {args.gpus}; launch one rank per GPU", file=sys.stderr)
        return 2

    import torch
    import torch.distributed as dist
    import spmv_amd as S
    from spmv_amd import generators as G
    from spmv_amd.matrix import Matrix

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SPMV_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with fewer GPUs than ranks
    # (ranks share devices, halo staged through the host); the driver's runs use nccl = RCCL.
    backend = os.environ.get("SPMV_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < world:
        print(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node has {ndev}", file=sys.stderr)
        return 2
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
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
    if world > 1 and args.matrix != "stencil27":
        print("bench.py: the multi-GPU bench is defined on the stencil27 workload", file=sys.stderr)
        return 2
    if args.matrix == "stencil27":
        n = args.n or 252
        if world == 1:
            rows, cols, rp, cj, va = G.stencil27(n, device=dev)
            workload = f"stencil27 {n}^3"
        else:
            plane = n * n
            if args.total_planes:           # 530^3 over 8 GPUs: --size 530 --total-planes 530 (slabs of 66 or 67 planes)
                nz_all = args.total_planes
                p0_, p1_ = rank * nz_all // world, (rank + 1) * nz_all // world
                workload = f"stencil27 {n}x{n}x{nz_all}, z-slabs of {nz_all // world}..{-(-nz_all // world)} planes per GPU"
            else:
                nzp = args.planes or n
                nz_all = nzp * world
                p0_, p1_ = rank * nzp, (rank + 1) * nzp
                workload = f"stencil27 {n}x{n}x{nz_all}, z-slab of {nzp} planes per GPU"
            r0, r1 = p0_ * plane, p1_ * plane
            rows, cols, rp, cj, va = G.stencil27(n, n, nz_all, device=dev, row_begin=r0, row_end=r1)
    elif args.matrix == "laplacian2d":
        n = args.n or 1000
        rows, cols, rp, cj, va = G.laplacian2d(n, device=dev)
        workload = f"laplacian2d {n}^2"
    else:
        n = args.n or 22
        rows, cols, rp, cj, va = G.rmat(n, 16, device=dev)
        workload = f"rmat scale {n} edge factor 16"
    nnz = int(rp[-1])
    va = va.to(tdtype)

    m = op = None
    if world == 1:
        m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), dtype, p0, p1,
                                   stream, keepalive=(rp, cj, va))
        if args.variant >= 0:
            m.set_option("variant", args.variant)
        else:
            m.analyze(3)       # kernel variant by trial, as create_sell does (reference sell.c:30)
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
        global_rows, global_nnz = rows, nnz
    else:
        from spmv_amd.distributed import RowBlockSpmv
        op = RowBlockSpmv(fmt, rows, cols, rp, cj, va, dtype, p0, p1, row_begin=r0, exchange=args.exchange,
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
        t = torch.tensor([nnz, rows], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        global_nnz, global_rows = int(t[0].item()), int(t[1].item())

    def timed_region(fn, steps, warmup):
        """`warmup` untimed steps, then exactly `steps` bracketed by barrier + synchronize on both sides.
        Returns (wall seconds, max over ranks; sorted per-step HIP-event durations in ms, this rank)."""
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        starts = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        ends = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        t0 = time.perf_counter()
        for k in range(steps):
            starts[k].record()
            fn()
            ends[k].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, sorted(s.elapsed_time(e) for s, e in zip(starts, ends))

    # ---- timed region -------------------------------------------------------
    elapsed, kernel_ms = timed_region(step, args.steps, args.warmup)
    mean_kernel_ms = sum(kernel_ms) / len(kernel_ms)

    # outside the timed region: the exchange alone and the local products alone, to say how much of
    # the communication the overlap hides (max over ranks of per-step means)
    comm = None
    if world > 1 and not args.no_comm_breakdown:
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
            tt = torch.tensor([a.elapsed_time(b) / reps], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt.item())
        exchange_ms = timed_ms(op.exchange_only)
        compute_ms = timed_ms(lambda: op.compute_only(y))
        comm = {"exchange_alone_ms": round(exchange_ms, 5), "local_products_alone_ms": round(compute_ms, 5),
                "halo_entries_rank0": int(op.nhalo), "backend": backend,
                "collectives": "batch_isend_irecv (one group of peer-to-peer sends/receives per product)"
                               if args.exchange == "halo" else "all_gather_into_tensor of x"}

    # outside the timed region, N = 1: the fastest variant of the same handle that reads every column
    # index (what a matrix without grid structure gets), measured the same way
    b_alg = algorithmic_bytes(rows, cols if world == 1 else rows, nnz, vbytes)   # one launch, one GPU
    general = None
    index_skipping = {S.SELL: range(16, 24), S.SELLP: range(16, 24), S.ELL: range(16, 24), S.CSR_SCALAR: range(16, 23),
                      S.CSR_VECTOR: (7, 8), S.CSR_MERGE: (2,), S.CSR_BAL: (2,)}[fmt]
    if world == 1 and not args.no_general and variant in index_skipping:
        every_index = {S.SELL: (0, 1, 2, 3, 4, 9, 10, 11, 12), S.SELLP: (0, 1, 2, 3, 4, 9, 10, 11, 12),
                       S.ELL: (0, 1, 2, 3, 4, 9, 10, 11, 12), S.CSR_SCALAR: (0, 2, 3, 4, 8, 10, 11, 12),
                       S.CSR_VECTOR: (5, 6), S.CSR_MERGE: (0, 1), S.CSR_BAL: (0, 1)}[fmt]
        trial = {}
        for v in every_index:
            m.set_option("variant", v)
            _, ms = timed_region(step, 5, 2)
            trial[v] = ms[len(ms) // 2]
        gv = min(trial, key=trial.get)
        m.set_option("variant", gv)
        _, gms = timed_region(step, args.steps, args.warmup)
        g_mean = sum(gms) / len(gms)
        m.set_option("variant", variant)

    def roofline_record(mean_ms, ms_sorted, v):
        achieved = b_alg / (mean_ms * 1e-3) / 1e9
        key = f"{args.format}/{args.matrix}/{n}/{args.dtype}/gpus{world}/variant{v}"
        traffic, note = recorded_traffic(key)
        rec = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
               "frac_moved": round(traffic / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
               "traffic_source": note if traffic else None,
               "algorithmic_bytes_per_launch": b_alg, "kernel_variant": v, "kernel_ms_mean": round(mean_ms, 5),
               "kernel_ms_min": round(ms_sorted[0], 5), "kernel_ms_median": round(ms_sorted[len(ms_sorted) // 2], 5)}
        if traffic is None:
            print(f"bench.py: roofline.traffic is null: {note}", file=sys.stderr)
        return rec

    if world == 1 and not args.no_general and variant in index_skipping:
        general = roofline_record(g_mean, gms, gv)

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        gflops = 2.0 * global_nnz / (ms_per_step * 1e-3) / 1e9
        out = {
            "metric": METRIC,
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "strong" if args.total_planes else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload, "format": args.format,
                       "params": {"C": 64, "sigma": args.sigma} if fmt == S.SELL else {"p0": p0},
                       "rows": global_rows, "nnz": global_nnz, "stored_per_gpu": int(stored), "kernel_variant": variant,
                       "exchange": args.exchange if world > 1 else None},
            "roofline": roofline_record(mean_kernel_ms, kernel_ms, variant),
        }
        if general is not None:
            out["roofline"]["general"] = general
        if comm is not None:
            comm["exposed_ms"] = round(max(ms_per_step - comm["local_products_alone_ms"], 0.0), 5)
            out["communication"] = comm
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_n)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
