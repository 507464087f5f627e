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
bytes nnz*(V+4) + rows*(4+V) + cols*V of one launch divided by the kernel's duration
`kernel_ms`; peak is 8 TB/s (MI355X HBM3E).  `kernel_ms` is the smaller of two bounds
from above: `kernel_ms_mean`, the mean of the HIP-event pairs around every launch of
the timed region (the headline kernel's: it matches the kernel trace), and
`kernel_ms_in_a_row`, the same number of launches queued between ONE event pair after
the timed region (a 12 us product of the small matrix: Python lets the queue run empty
between two launches, and an event pair around a launch on an empty queue reads 2 us
more than the kernel trace).
`roofline.traffic` is the HBM traffic of one launch from the PMC passes recorded
in profiles/hbm_traffic.json for exactly this kernel source (null otherwise),
`frac_moved` the fraction of peak those bytes amount to; `roofline.general` is the
same measurement for the fastest kernel variant that reads every column index.
`cpu_baseline` times the compiled reference (oracle/_ref, kind "reference") or the
oracle port on the host cores for a bounded sample of the same workload; it runs, in
a child process, before this process makes its first GPU call.
`other_configs` (N = 1, the default command only) carries the other single-GPU
configurations of the benchmark list, each measured like the headline one: the 2-D
5-point Laplacian 1000^2 on lane-per-row CSR (the format of csr.c) and on SELLP, and the R-MAT scale 26 matrix on balanced CSR as it is and with its
columns dealt over the XCDs (option hot_columns 2, with the time of its three kernels).
For N > 1 the line also carries `rccl` (what the communicator reports) and
`c_abi_one_process`: the same slabs driven by ONE process through spmv_hip_dist_*
(peer-to-peer pushes between the GPUs), run in a child after the ranks have finished.
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
                                                           "csr_merge", "csr_bal", "sell_dfc"])
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
    ap.add_argument("--block-map", type=int, default=-1,
                    help="N = 1, with --variant: option block_map (how workgroups are dealt over the matrix) instead of 0")
    ap.add_argument("--hot-columns", type=int, default=0, choices=[0, 1, 2],
                    help="N = 1, CSR-based formats: option hot_columns (1 columns relabeled by reference count, 2 dealt over the XCDs)")
    ap.add_argument("--hot-variant", type=int, default=-1, help="with --hot-columns 2: this tile kernel of the column-split copy")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1: skip the other single-GPU configurations (Laplacian 1000^2, R-MAT 26)")
    ap.add_argument("--rmat-scale", type=int, default=26, help="R-MAT scale of the other_configs entry")
    ap.add_argument("--no-one-process", action="store_true",
                    help="N > 1: skip the run of the same slabs through spmv_hip_dist_* from one process")
    ap.add_argument("--one-process-only", action="store_true",
                    help="run only the one-process spmv_hip_dist_* leg over --gpus devices and print its JSON")
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

    def power_loop(mult, rows, cols, nnz, min_seconds=2.0, vectors=None):
        if vectors is None:
            x = oracle.aligned_empty(cols)      # the reference's AVX2 kernels need 64-byte aligned vectors
            y = oracle.aligned_empty(rows + 64, fill=0.0)
        else:
            x, y = vectors                      # already placed (first touched) by the caller
        x[:] = 1.0 / np.sqrt(cols)
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
                                    min_seconds=2.0 if t == threads else 1.0)
            if t == threads:
                break
            t = min(2 * t, threads)
        results[("csr_merge", threads)] = scaling[threads]
        bal = R.ref_create_csr_bal(threads, rows, cols, nnz, coo)
        results[("csr_bal", threads)] = power_loop(lambda x, y: R.ref_mult_csr_bal(bal, x, y), rows, cols, nnz)
    else:
        kind = "port"
        results[("csr_base", 1)] = pinned(lambda: power_loop(lambda x, y: L.orc_mult_csr_base(rows, rp, cj, va, x, y), rows, cols, nnz))
    # rows dealt over the threads, every thread's rows first written by that thread (the reference's
    # create_csr_numa / mult_csr_numa, csr_numa.c:3-58, which its driver never calls; here as the restatement):
    # the creators above fill their arrays from one thread, so all pages sit where that thread ran
    rp2, cj2, va2 = np.empty(rows + 1, np.int32), np.empty(nnz, np.int32), np.empty(nnz, np.float64)
    x2, y2 = oracle.aligned_empty(cols), oracle.aligned_empty(rows + 64)
    L.orc_csr_copy_first_touch(rows, rp, cj, va, rp2, cj2, va2, x2[:rows], y2[:rows])
    results[("csr_omp_first_touch(port)", threads)] = power_loop(lambda x, y: L.orc_mult_csr_omp(rows, rp2, cj2, va2, x, y),
                                                                 rows, cols, nnz, vectors=(x2, y2))
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
        "value": round(best, 3), "unit": "GFLOP/s", "cores": best_cores,
        "kind": "port" if "(port)" in best_name else kind, "kernel": best_name,
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


# ------------------------------------------------- one process over all GPUs --

def one_process_child(args):
    """The slabs of the N > 1 run behind ONE process and the C ABI: a spmv_hip_rowblock per GPU built from
    its CSR shard in HBM, spmv_hip_dist_create_from_blocks, spmv_hip_dist_analyze, then the driver's
    timed loop (spmv_hip_dist_bench: exchange by peer-to-peer pushes + products, host clock per product).
    Before it, one product of x_i = i: on every row away from the grid's faces the stencil (centre 26,
    26 neighbours -1, offsets symmetric) must give exactly 0.0 in any summation order -- every partial sum
    is an integer below 2^53 -- so a halo entry that arrives late, or in the wrong slot, shows.
    Prints one JSON object."""
    import ctypes as C
    import numpy as np
    import torch
    import spmv_amd as S
    from spmv_amd import _lib as L
    from spmv_amd import generators as G

    P = args.gpus
    ndev = torch.cuda.device_count()
    backend = os.environ.get("SPMV_BENCH_BACKEND", "nccl")
    if ndev < 1 or (backend == "nccl" and ndev < P):
        print(json.dumps({"error": f"{P} blocks need {P} GPUs, this node has {ndev}"}))
        return 0
    devices = [p % ndev for p in range(P)]
    dtype = S.F64 if args.dtype == "f64" else S.F32
    tdtype = torch.float64 if dtype == S.F64 else torch.float32
    fmt = FORMATS[args.format]
    p0 = args.p0 or (64 if fmt == S.SELL else 0)
    p1 = args.sigma if fmt == S.SELL else 0
    n = args.n or 252
    plane = n * n
    if args.total_planes:
        nz_all = args.total_planes
        cuts = [p * nz_all // P for p in range(P + 1)]
    else:
        nzp = args.planes or n
        nz_all = nzp * P
        cuts = [p * nzp for p in range(P + 1)]
    lib = S.lib()
    t_build = time.perf_counter()
    blocks, nnz_total = [], 0
    for p in range(P):
        dev = torch.device("cuda", devices[p])
        torch.cuda.set_device(dev)
        L.check(lib.spmv_hip_init(devices[p]))
        r0, r1 = cuts[p] * plane, cuts[p + 1] * plane
        rows, cols, rp, cj, va = G.stencil27(n, n, nz_all, device=dev, row_begin=r0, row_end=r1)
        va = va.to(tdtype)
        nnz = int(rp[-1])
        nnz_total += nnz
        blocks.append(L.check_handle(lib.spmv_hip_rowblock_create(fmt, rows, r0, cols, nnz, rp.data_ptr(), cj.data_ptr(),
                                                                 va.data_ptr(), dtype, p0, p1, 0)))
        torch.cuda.synchronize()
        del rp, cj, va
        torch.cuda.empty_cache()
    arr = (C.c_void_p * P)(*blocks)
    d = L.check_handle(lib.spmv_hip_dist_create_from_blocks(P, arr))
    L.check(lib.spmv_hip_dist_analyze(d, 3))
    build_s = time.perf_counter() - t_build
    access = np.zeros(P * P, np.int32)
    L.check(lib.spmv_hip_dist_get_peer_access(d, access.ctypes.data_as(C.c_void_p)))
    pairs = [access[q * P + p] for q in range(P) for p in range(P) if p != q]
    info = (L.RowBlockInfo * P)()
    L.check(lib.spmv_hip_dist_get_info(d, None, None, info))
    total_rows = plane * nz_all
    out = {"blocks": P, "devices": devices, "format": args.format, "dtype": args.dtype, "rows": total_rows, "nnz": nnz_total,
           "halo_per_block": [b.halo for b in info], "interior_variant": [b.interior_variant for b in info],
           "peer_access": {"direct": int(sum(v == 1 for v in pairs)), "through_host": int(sum(v == 0 for v in pairs)),
                           "same_device": int(sum(v == 2 for v in pairs))},
           "helper_threads": os.environ.get("SPMV_DIST_THREADS", "1") != "0", "build_seconds": round(build_s, 2)}
    def halo_check(scale, shift):
        """One product of x_i = scale * i + shift: exactly 0.0 on every row away from the grid's faces."""
        x = np.arange(total_rows, dtype=np.float64) * scale + shift
        y = np.full(total_rows, np.nan)
        L.check(lib.spmv_hip_dist_mult(d, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)))
        g = y.reshape(nz_all, n, n)[1:-1, 1:-1, 1:-1]
        return int(g.size), int(np.count_nonzero(g)), bool(np.isfinite(y).all())

    if dtype == S.F64:
        checked, bad, finite = halo_check(1.0, 0.0)
        out["halo_check"] = {"product": "x_i = i: exactly 0.0 on every row away from the grid's faces", "rows_checked": checked,
                             "rows_wrong": bad, "ok": bad == 0 and finite}
    rng = np.random.default_rng(1234)
    x = rng.random(total_rows)
    x /= np.linalg.norm(x)
    it, sec = C.c_int(0), C.c_double(0.0)
    # the helper threads' first launches pay the runtime's per-thread set-up (the first ~50 products of a new handle
    # run several times slower): 100 untimed products before the warm-up the command line asks for
    L.check(lib.spmv_hip_dist_bench(d, 100, 0.0, x.ctypes.data_as(C.c_void_p), C.byref(it), C.byref(sec)))
    if args.warmup > 0:
        L.check(lib.spmv_hip_dist_bench(d, args.warmup, 0.0, x.ctypes.data_as(C.c_void_p), C.byref(it), C.byref(sec)))
    L.check(lib.spmv_hip_dist_bench(d, args.steps, 0.0, x.ctypes.data_as(C.c_void_p), C.byref(it), C.byref(sec)))
    ms = sec.value / it.value * 1e3
    vbytes = 8 if dtype == S.F64 else 4
    out.update({"steps": it.value, "ms_per_step": round(ms, 5), "value": round(2.0 * nnz_total / ms / 1e6, 2), "unit": "GFLOP/s",
                "timing": "host clock around each product, from the first launch to the last block's completion (spmv_hip_dist_bench)",
                "algorithmic_gbs": round(algorithmic_bytes(total_rows, total_rows, nnz_total, vbytes) / ms / 1e6, 1)})
    if dtype == S.F64:
        # once more after the timed loop, with another x: every cache of every GPU now holds halo lines of the
        # loop's last iterate, and a consumer that read one of those instead of the entry just delivered shows
        checked, bad, finite = halo_check(3.0, 7.0)
        out["halo_check"]["after_the_timed_loop"] = {"product": "x_i = 3 i + 7", "rows_wrong": bad, "ok": bad == 0 and finite}
        out["halo_check"]["ok"] = out["halo_check"]["ok"] and bad == 0 and finite
    lib.spmv_hip_dist_free(d)
    print(json.dumps(out), flush=True)
    return 0


def one_process_leg(argv, gpus):
    """Runs one_process_child in a child process of its own (it drives every GPU of the node) and returns its
    record; a failure there becomes {"error": ...} and never costs the rank's own result line."""
    keep = [a for a in argv if a not in ("--no-comm-breakdown",)]
    cmd = [sys.executable, os.path.abspath(__file__)] + keep + ["--one-process-only"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                              "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE")}
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    except subprocess.TimeoutExpired:
        return {"error": "the one-process leg did not finish within 420 s"}
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        return {"error": f"exit status {out.returncode}: {(out.stderr or out.stdout)[-600:]}"}
    return json.loads(lines[-1])


# ------------------------------------------------------------------------- main --

FORMATS = {}            # name -> library constant, filled on first use (the import must not precede the rank launcher)


def _formats():
    import spmv_amd as S
    if not FORMATS:
        FORMATS.update({"sell": S.SELL, "sellp": S.SELLP, "ell": S.ELL, "csr_scalar": S.CSR_SCALAR, "csr_vector": S.CSR_VECTOR,
                        "csr_merge": S.CSR_MERGE, "csr_bal": S.CSR_BAL, "sell_dfc": S.SELL_DFC})
    return FORMATS


def traffic_key(fmt_name, matrix, n, dtype_name, world, info):
    """Name of a configuration in profiles/hbm_traffic.json."""
    v = f"variant{info.variant}"
    if fmt_name == "csr_vector":
        v += f"lanes{info.lanes_per_row}"
    if info.hot_columns == 1:
        v += "+hot1"
    elif info.hot_columns == 2:
        v += f"+hot2tiles{info.hot_variant}"
    if info.block_map and info.hot_columns != 2:
        v += f"+map{info.block_map}"
    return f"{fmt_name}/{matrix}/{n}/{dtype_name}/gpus{world}/{v}"


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.cpu_baseline_only:
        cpu_baseline_child(args.cpu_n)
        return 0
    if args.one_process_only:
        _formats()
        return one_process_child(args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return launch_ranks(args, argv)
    world = int(env_world or "1")
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE is {world} but --gpus is {args.gpus}; launch one rank per GPU", file=sys.stderr)
        return 2

    # the host-CPU baseline first, in a child of its own: this process has not touched a GPU yet, and the
    # child's threads have the host to themselves
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.cpu_n)

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

    formats = _formats()
    stream = torch.cuda.current_stream().cuda_stream

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

    def launches_in_a_row_ms(fn, steps):
        """ms per launch of `steps` launches queued one after the other between ONE pair of HIP events (outside the
        timed region).  An event pair around every launch is the kernel's duration plus the start of a launch on a
        queue that Python lets run empty: on a product of ~12 us that is 2 us more than the kernel trace shows, on the
        headline product nothing.  Queued launches pay the gap between two kernels of a stream instead (~6 us after a
        0.75 ms kernel, nothing measurable after a short one).  Both bound the kernel's duration from above."""
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(steps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / steps

    def roofline_record(b_alg, mean_ms, ms_sorted, key, v, in_a_row_ms=None):
        """`mean_ms`: mean of the HIP-event pairs around every launch of the timed region; `in_a_row_ms`: the same
        launches queued between one pair.  The kernel's duration is taken as the smaller of the two."""
        events_ms = mean_ms
        if in_a_row_ms is not None and in_a_row_ms < mean_ms:
            mean_ms = in_a_row_ms
        achieved = b_alg / (mean_ms * 1e-3) / 1e9
        traffic, note = recorded_traffic(key)
        rec = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
               "frac_moved": round(traffic / (mean_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
               "traffic_source": note if traffic else None,
               "algorithmic_bytes_per_launch": b_alg, "kernel_variant": v, "kernel_ms": round(mean_ms, 5),
               "kernel_ms_mean": round(events_ms, 5),
               "kernel_ms_in_a_row": round(in_a_row_ms, 5) if in_a_row_ms is not None else None,
               "kernel_ms_min": round(ms_sorted[0], 5), "kernel_ms_median": round(ms_sorted[len(ms_sorted) // 2], 5)}
        if traffic is None:
            print(f"bench.py: roofline.traffic is null: {note}", file=sys.stderr)
        return rec

    def generate(matrix, n):
        if matrix == "stencil27":
            return G.stencil27(n, device=dev), f"stencil27 {n}^3"
        if matrix == "laplacian2d":
            return G.laplacian2d(n, device=dev), f"laplacian2d {n}^2"
        return G.rmat(n, 16, device=dev), f"rmat scale {n} edge factor 16"

    skip_sliced = tuple(range(16, 24)) + (28, 29, 30, 31)
    index_skipping = {S.SELL: skip_sliced, S.SELLP: skip_sliced, S.ELL: skip_sliced, S.CSR_SCALAR: range(16, 23),
                      S.CSR_VECTOR: (7, 8), S.CSR_MERGE: (2,), S.CSR_BAL: (2,), S.SELL_DFC: range(9, 13)}
    every_sliced = (0, 1, 2, 3, 4, 9, 10, 11, 12)
    every_index = {S.SELL: every_sliced, S.SELLP: every_sliced, S.ELL: every_sliced, S.CSR_SCALAR: (0, 2, 3, 4, 8, 10, 11, 12),
                   S.CSR_VECTOR: (5, 6), S.CSR_MERGE: (0, 1), S.CSR_BAL: (0, 1), S.SELL_DFC: tuple(range(0, 9))}

    def single_gpu_config(matrix, n, fmt_name, dtype_name, sigma=1024, p0_arg=0, variant=-1, hot=0, hot_variant=-1, steps=20,
                          warmup=3, want_general=False, csr=None, block_map=-1):
        """One configuration on this GPU, measured like the headline one.  fmt_name "auto": the format the
        library's trial picks (spmv_hip_create_auto_from_device_csr).  `csr`: a generated matrix to reuse."""
        dtype = S.F64 if dtype_name == "f64" else S.F32
        tdtype = torch.float64 if dtype == S.F64 else torch.float32
        vbytes = 8 if dtype == S.F64 else 4
        (rows, cols, rp, cj, va), workload = csr if csr is not None else generate(matrix, n)
        nnz = int(rp[-1])
        va = va.to(tdtype)
        if fmt_name == "auto":
            m = Matrix.auto_from_device_csr(rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), dtype, 3, stream,
                                            keepalive=(rp, cj, va))
            fmt = m.info.format
            fmt_name = S.FORMAT_NAMES[fmt]
        else:
            fmt = formats[fmt_name]
            p0 = p0_arg or (64 if fmt in (S.SELL, S.SELL_DFC) else 0)
            p1 = sigma if fmt in (S.SELL, S.SELL_DFC) else 0
            m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), dtype, p0, p1,
                                       stream, keepalive=(rp, cj, va))
            if hot:
                m.set_option("hot_columns", hot)
            if variant >= 0:
                m.set_option("variant", variant)
                if block_map >= 0:
                    m.set_option("block_map", block_map)
                if hot == 2 and hot_variant >= 0:
                    m.set_option("hot_variant", hot_variant)
            else:
                m.analyze(3)       # kernel variant by trial, as create_sell does (reference sell.c:30)
        if fmt in (S.SELL, S.SELLP, S.ELL, S.SELL_DFC):      # the sliced layouts are copies: the CSR arrays may go
            m._keepalive = None
        del rp, cj, va
        torch.cuda.empty_cache()
        g = torch.Generator(device=dev).manual_seed(1234)
        x = torch.rand(cols, dtype=tdtype, device=dev, generator=g)
        x /= torch.linalg.vector_norm(x)
        y = torch.empty(rows, dtype=tdtype, device=dev)

        def step():
            m.mult_dev(x.data_ptr(), y.data_ptr(), stream)
        elapsed, kernel_ms = timed_region(step, steps, warmup)
        in_a_row_ms = launches_in_a_row_ms(step, steps)
        info = m.info
        b_alg = algorithmic_bytes(rows, cols, nnz, vbytes)
        res = {"workload": workload, "format": fmt_name, "fmt": fmt, "dtype": dtype_name, "rows": rows, "cols": cols, "nnz": nnz,
               "stored": int(info.stored), "variant": info.variant, "lanes_per_row": info.lanes_per_row,
               "hot_columns": info.hot_columns, "hot_variant": info.hot_variant, "block_map": info.block_map, "elapsed": elapsed, "kernel_ms": kernel_ms, "in_a_row_ms": in_a_row_ms,
               "b_alg": b_alg, "key": traffic_key(fmt_name, matrix, n, dtype_name, 1, info), "general": None}
        # outside the timed region: the fastest variant of the same handle that reads every column index
        # (what a matrix without grid structure gets), measured the same way
        if want_general and not info.hot_columns and info.variant in index_skipping[fmt]:
            trial = {}
            candidates = every_index[fmt]
            if fmt in (S.SELL, S.SELLP, S.ELL) and dtype == S.F64:
                candidates = candidates + (24, 25, 26, 27)      # 16-byte loads in the exact summation order (fp64 kernels)
            for v in candidates:
                m.set_option("variant", v)
                _, ms = timed_region(step, 5, 2)
                trial[v] = ms[len(ms) // 2]
            gv = min(trial, key=trial.get)
            m.set_option("variant", gv)
            other = {}
            for bm in (0, 1, 2, 3, 4, 5, 8):        # and the fastest way of dealing the workgroups over the matrix
                m.set_option("block_map", bm)
                _, ms = timed_region(step, 5, 2)
                other[bm] = ms[len(ms) // 2]
            m.set_option("block_map", min(other, key=other.get))
            _, gms = timed_region(step, steps, warmup)
            gkey = traffic_key(fmt_name, matrix, n, dtype_name, 1, m.info)
            res["general"] = roofline_record(b_alg, sum(gms) / len(gms), gms, gkey, gv)
            res["general"]["block_map"] = m.info.block_map
            m.set_option("variant", info.variant)
            m.set_option("block_map", info.block_map)
        m.free()
        del x, y
        torch.cuda.empty_cache()
        return res

    def other_entry(name, res):
        """One line of `other_configs` from a single_gpu_config result."""
        ms = res["kernel_ms"]
        rec = roofline_record(res["b_alg"], sum(ms) / len(ms), ms, res["key"], res["variant"], res["in_a_row_ms"])
        mean = rec["kernel_ms"]          # the smaller of: event pair per launch (mean), launches queued in a row
        entry = {"config": name, "workload": res["workload"], "format": res["format"], "dtype": res["dtype"],
                 "rows": res["rows"], "nnz": res["nnz"], "kernel_variant": res["variant"], "block_map": res["block_map"], "hot_columns": res["hot_columns"],
                 "ms": round(mean, 5), "ms_event_pair_per_launch": rec["kernel_ms_mean"], "ms_in_a_row": rec["kernel_ms_in_a_row"],
                 "ms_median": rec["kernel_ms_median"], "gflops": round(2.0 * res["nnz"] / mean / 1e6, 1),
                 "frac": rec["frac"], "algorithmic_bytes_per_launch": res["b_alg"], "traffic": rec["traffic"],
                 "frac_moved": rec["frac_moved"], "traffic_key": res["key"]}
        if res["format"] == "csr_vector":
            entry["lanes_per_row"] = res["lanes_per_row"]
        if res["hot_columns"] == 2:
            entry["hot_variant"] = res["hot_variant"]
            table = {}
            try:
                table = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            except OSError:
                pass
            rec_k = table.get(res["key"], {})
            if rec_k.get("kernel_sources_sha256") == kernel_source_hash() and "kernels" in rec_k:
                entry["kernels"] = rec_k["kernels"]      # per kernel of the product: ms and HBM bytes (rocprofv3 passes)
        return entry

    if world > 1 and args.matrix != "stencil27":
        print("bench.py: the multi-GPU bench is defined on the stencil27 workload", file=sys.stderr)
        return 2
    dtype = S.F64 if args.dtype == "f64" else S.F32
    tdtype = torch.float64 if dtype == S.F64 else torch.float32
    vbytes = 8 if dtype == S.F64 else 4
    fmt = formats[args.format]
    n = args.n or {"stencil27": 252, "laplacian2d": 1000, "rmat": 22}[args.matrix]

    if world == 1:
        # ---- the headline configuration (or the one the flags name) -----------------
        res = single_gpu_config(args.matrix, n, args.format, args.dtype, args.sigma, args.p0, args.variant, args.hot_columns,
                                args.hot_variant, args.steps, args.warmup, want_general=not args.no_general,
                                block_map=args.block_map)
        kernel_ms = res["kernel_ms"]
        mean_kernel_ms = sum(kernel_ms) / len(kernel_ms)
        ms_per_step = res["elapsed"] * 1e3 / args.steps
        out = {
            "metric": METRIC,
            "value": round(2.0 * res["nnz"] / (ms_per_step * 1e-3) / 1e9, 2), "unit": "GFLOP/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": res["workload"], "format": args.format,
                       "params": {"C": 64, "sigma": args.sigma} if fmt == S.SELL else {"p0": args.p0},
                       "rows": res["rows"], "nnz": res["nnz"], "stored_per_gpu": res["stored"], "kernel_variant": res["variant"],
                       "block_map": res["block_map"], "hot_columns": res["hot_columns"], "exchange": None},
            "roofline": roofline_record(res["b_alg"], mean_kernel_ms, kernel_ms, res["key"], res["variant"], res["in_a_row_ms"]),
        }
        if res["general"] is not None:
            out["roofline"]["general"] = res["general"]
        # ---- the other single-GPU configurations of the benchmark list ---------------
        default_run = (args.matrix, n, args.format, args.dtype, args.variant, args.hot_columns) == ("stencil27", 252, "sell", "f64", -1, 0)
        if default_run and not args.no_other_configs:
            others = []
            # the 80 MB of this matrix stay in the caches and a product takes ~15 us, so a kernel trial here picks by
            # noise: both entries name their kernel (the trial's usual picks).  No kernel trial runs on this matrix, so that
            # every launch of the headline's kernels in this process belongs to the headline's matrix (the kernel-trace
            # summary of this command averages per kernel)
            lap = generate("laplacian2d", 1000)
            others.append(other_entry("C1: plain CSR, the format of csr.c (lane per row, the library's default kernel)",
                                      single_gpu_config("laplacian2d", 1000, "csr_scalar", "f64", variant=0, csr=lap)))
            others.append(other_entry("C1: SELLP, kernel that skips the indices of diagonal slice columns",
                                      single_gpu_config("laplacian2d", 1000, "sellp", "f64", variant=16, csr=lap)))
            del lap
            torch.cuda.empty_cache()
            rm = generate("rmat", args.rmat_scale)
            others.append(other_entry("C3: balanced CSR as it is", single_gpu_config("rmat", args.rmat_scale, "csr_bal", "f64", csr=rm)))
            others.append(other_entry("C3: balanced CSR, columns dealt over the XCDs (hot_columns 2)",
                                      single_gpu_config("rmat", args.rmat_scale, "csr_bal", "f64", hot=2, csr=rm)))
            del rm
            torch.cuda.empty_cache()
            out["other_configs"] = others
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
        return 0

    # ---- N > 1: one z-slab of the stencil per rank ----------------------------------
    p0 = args.p0 or (64 if fmt == S.SELL else 0)
    p1 = args.sigma if fmt == S.SELL else 0
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
    nnz = int(rp[-1])
    va = va.to(tdtype)
    from spmv_amd.distributed import RowBlockSpmv
    op = RowBlockSpmv(fmt, rows, cols, rp, cj, va, dtype, p0, p1, row_begin=r0, exchange=args.exchange, device=dev)
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

    elapsed, kernel_ms = timed_region(step, args.steps, args.warmup)
    mean_kernel_ms = sum(kernel_ms) / len(kernel_ms)

    # outside the timed region: the exchange alone and the local products alone, to say how much of
    # the communication the overlap hides (max over ranks of per-step means)
    comm = None
    if not args.no_comm_breakdown:
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
        try:
            exchange_ms = timed_ms(op.exchange_only)
            compute_ms = timed_ms(lambda: op.compute_only(y))
            comm = {"exchange_alone_ms": round(exchange_ms, 5), "local_products_alone_ms": round(compute_ms, 5),
                    "halo_entries_rank0": int(op.nhalo), "backend": backend,
                    "collectives": "batch_isend_irecv (one group of peer-to-peer sends/receives per product)"
                                   if args.exchange == "halo" else "all_gather_into_tensor of x"}
        except Exception as e:                # the breakdown is an extra: the timed result above stands without it
            comm = {"error": f"{type(e).__name__}: {e}"[:300]}
    # what the communicator itself reports: every rank contributes its device to one all-gather
    mine = torch.tensor([dev_index], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
    seen = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(seen, mine)
    try:
        rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception as e:                    # a version query must never cost the result line
        rccl_version = f"unavailable ({type(e).__name__})"
    rccl = {"world": dist.get_world_size(), "backend": dist.get_backend(), "ranks_seen": len(seen),
            "devices_of_ranks": [int(v.item()) for v in seen], "version": rccl_version}
    b_alg = algorithmic_bytes(rows, rows, nnz, vbytes)        # one launch group, this GPU's slab
    key = f"{args.format}/{args.matrix}/{n}/{args.dtype}/gpus{world}/variant{variant}"
    out = None
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
                       "exchange": args.exchange},
            "roofline": roofline_record(b_alg, mean_kernel_ms, kernel_ms, key, variant),
            "rccl": rccl,
        }
        if comm is not None:
            if "error" not in comm:
                comm["exposed_ms"] = round(max(ms_per_step - comm["local_products_alone_ms"], 0.0), 5)
            out["communication"] = comm
    # the ranks are done: release the GPUs, then (rank 0) the same slabs from one process through the C ABI
    del op, x, y
    torch.cuda.empty_cache()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        if not args.no_one_process:
            out["c_abi_one_process"] = one_process_leg(argv, world)
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
