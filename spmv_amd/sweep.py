"""Sweep over the stand-ins of the reference's matrix list: every format in
fp64 and fp32, one CSV line per (matrix, format, dtype).

Counterpart of `./spmv.run` over matrix_lists/matrix_list_160.txt (reference
spmv.run:20-33, driver_spmv.c:139-316): per matrix the product is checked
first (the reference compares with the COO product and prints OK/ERR at
rel-L2 1e-10, driver_spmv.c:274-289,313; here the yardstick is the
bit-exact lane-per-row CSR kernel in fp64), then timed with the driver's power
loop (>= min_iters products and >= min_seconds, x <- y/|y| between products,
only the products timed, :292-308), GFLOP/s from the true nnz (:307).

    python -m spmv_amd.sweep --out sweep.csv [--scale 0.1] [--only name,name]
    python -m torch.distributed.run --nproc-per-node 8 -m spmv_amd.sweep --out sweep.csv

With several ranks the list is dealt out by size (largest first, round robin):
matrices are independent, so there is no data-path collective; rank 0 collects
the lines.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

import spmv_amd as S
from spmv_amd import standins
from spmv_amd.matrix import Matrix

CSV_HEADER = "Format;File;Rows;Columns;NNZ;GFlops;Time(s);GB/s;PctHBMPeak;Dtype;Variant;Iters;RelErr;Check"
HBM_PEAK_GBS = 8000.0
# MODE string of the reference driver (driver_spmv.c:186-261) -> (format, p0, p1)
FORMATS = {
    "csr_base": (S.CSR_SCALAR, 0, 0),
    "csr_vec": (S.CSR_VECTOR, 0, 0),
    "csr_merge": (S.CSR_MERGE, 0, 0),
    "csr_bal": (S.CSR_BAL, 0, 0),
    "ell": (S.ELL, 0, 0),
    "sellp_vec": (S.SELLP, 0, 0),
    "sell": (S.SELL, 64, 1024),
    "sell_dfc": (S.SELL_DFC, 64, 1024),
    "auto": (-1, 0, 0),
}
TOLERANCE = {S.F64: 1e-10, S.F32: 1e-5}


def shard(names, rank: int, world: int):
    """Largest first, dealt round robin: every rank gets a similar mix of sizes."""
    order = sorted(names, key=lambda n: (-standins.BY_NAME[n][3], n))
    return order[rank::world]


def algorithmic_bytes(rows: int, cols: int, nnz: int, vbytes: int) -> int:
    return nnz * (vbytes + 4) + rows * (4 + vbytes) + cols * vbytes


def timed_power_loop(m: Matrix, x, y, dtype, min_iters: int, min_seconds: float, stream: int):
    """driver_spmv.c:292-305 with x, y in HBM: returns (iterations, summed product seconds)."""
    lib = S.lib()
    n = x.numel()
    norm = torch.zeros(1, dtype=torch.float64, device=x.device)
    iters, total_ms = 0, 0.0
    while iters < min_iters or total_ms < min_seconds * 1e3:
        batch = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in batch:
            a.record()
            m.mult_dev(x.data_ptr(), y.data_ptr(), stream)
            b.record()
            S.check(lib.spmv_hip_norm2_dev(n, y.data_ptr(), dtype, norm.data_ptr(), stream))
            S.check(lib.spmv_hip_scale_dev(n, y.data_ptr(), x.data_ptr(), dtype, norm.data_ptr(), stream))
        torch.cuda.synchronize()
        total_ms += sum(a.elapsed_time(b) for a, b in batch)
        iters += len(batch)
        if not torch.isfinite(norm).all() or float(norm) == 0.0:
            break
    return iters, total_ms / 1e3


def run_matrix(name: str, scale: float, formats, dtypes, min_iters: int, min_seconds: float, dev):
    """All (format, dtype) lines of one stand-in."""
    rows, cols, rp, cj, va = standins.standin(name, scale, device=dev)
    nnz = int(rp[-1])
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(1)
    x0 = torch.rand(cols, dtype=torch.float64, device=dev, generator=g)
    x0 /= torch.linalg.vector_norm(x0)
    ref = Matrix.from_device_csr(S.CSR_SCALAR, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), va.data_ptr(), S.F64, 0, 0, st)
    y_ref = torch.empty(rows, dtype=torch.float64, device=dev)
    ref.mult_dev(x0.data_ptr(), y_ref.data_ptr(), st)
    torch.cuda.synchronize()
    ref.free()
    ref_norm = float(torch.linalg.vector_norm(y_ref))
    lines = []
    for dtype in dtypes:
        td = torch.float64 if dtype == S.F64 else torch.float32
        vals = va if dtype == S.F64 else va.to(td)
        for mode in formats:
            fmt, p0, p1 = FORMATS[mode]
            try:
                if mode == "auto":                            # format by trial (spmv_hip_create_auto_from_device_csr)
                    m = Matrix.auto_from_device_csr(rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), vals.data_ptr(),
                                                    dtype, 2, st)
                    fmt = m.info.format
                else:
                    m = Matrix.from_device_csr(fmt, rows, cols, nnz, rp.data_ptr(), cj.data_ptr(), vals.data_ptr(),
                                               dtype, p0, p1, st)
                    m.analyze(2)
                x = x0.to(td, copy=True)                      # the power loop overwrites it
                y = torch.empty(rows, dtype=td, device=dev)
                m.mult_dev(x.data_ptr(), y.data_ptr(), st)
                torch.cuda.synchronize()
                err = float(torch.linalg.vector_norm(y.to(torch.float64) - y_ref)) / (ref_norm or 1.0)
                iters, seconds = timed_power_loop(m, x, y, dtype, min_iters, min_seconds, st)
                variant = m.info.lanes_per_row * 256 + m.info.variant if fmt == S.CSR_VECTOR else m.info.variant
                if mode == "auto":
                    variant = f"{S.FORMAT_NAMES[fmt]}:{variant}" + ("", "+hot", "+xcd")[m.info.hot_columns]
                    if m.info.block_map and m.info.hot_columns != 2:
                        variant += f"+map{m.info.block_map}"
                m.free()
            except S.SpmvHipError as e:
                lines.append(f"{mode};{name};{rows};{cols};{nnz};0;0;0;0;{'f64' if dtype == S.F64 else 'f32'};-1;0;nan;"
                             f"FAIL({str(e)[:60]})")
                continue
            vb = 8 if dtype == S.F64 else 4
            gflops = 2.0 * nnz * iters / seconds / 1e9
            gbs = algorithmic_bytes(rows, cols, nnz, vb) * iters / seconds / 1e9
            ok = "OK" if err <= TOLERANCE[dtype] else "ERR"
            if nnz > 50_000_000:                              # keep long runs visibly alive
                print(f"    {name} {mode} {'f64' if dtype == S.F64 else 'f32'}: {gflops:.1f} GFLOP/s {ok}", flush=True)
            lines.append(f"{mode};{name};{rows};{cols};{nnz};{gflops:.3f};{seconds:.6f};{gbs:.1f};"
                         f"{100.0 * gbs / HBM_PEAK_GBS:.2f};{'f64' if dtype == S.F64 else 'f32'};{variant};{iters};"
                         f"{err:.3e};{ok}")
    return lines


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--out", default="sweep.csv")
    ap.add_argument("--scale", type=float, default=1.0, help="multiply rows and nnz of every stand-in")
    ap.add_argument("--only", default="", help="comma list of matrix names (default: the whole list)")
    ap.add_argument("--formats", default="csr_vec,csr_merge,ell,sellp_vec,sell")
    ap.add_argument("--dtypes", default="f64,f32")
    ap.add_argument("--min-iters", type=int, default=10)
    ap.add_argument("--min-seconds", type=float, default=0.1)
    ap.add_argument("--max-nnz", type=int, default=0, help="skip stand-ins above this nnz (0: none)")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    if not torch.cuda.is_available():
        raise SystemExit("spmv_amd.sweep needs a GPU: the products run in libspmv_hip only")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    S.check(S.lib().spmv_hip_init(local))

    names = [n for n in args.only.split(",") if n] or standins.names()
    for n in names:
        if n not in standins.BY_NAME:
            raise SystemExit(f"unknown matrix {n!r}")
    if args.max_nnz:
        names = [n for n in names if standins.BY_NAME[n][3] * args.scale <= args.max_nnz]
    formats = [f for f in args.formats.split(",") if f]
    for f in formats:
        if f not in FORMATS:
            raise SystemExit(f"unknown format {f!r}; known: {', '.join(FORMATS)}")
    dtypes = [S.F64 if d == "f64" else S.F32 for d in args.dtypes.split(",") if d]

    mine = shard(names, rank, world)
    lines, t0 = [], time.time()
    for k, name in enumerate(mine):
        got = run_matrix(name, args.scale, formats, dtypes, args.min_iters, args.min_seconds, dev)
        lines += got
        print(f"[rank {rank}] {k + 1}/{len(mine)} {name}: {len(got)} lines, {time.time() - t0:.0f} s", flush=True)
        torch.cuda.empty_cache()
    if world > 1:
        import torch.distributed as dist
        parts = [None] * world
        dist.all_gather_object(parts, lines)
        lines = [ln for p in parts for ln in p]
        dist.destroy_process_group()
    if rank == 0:
        order = {n: i for i, n in enumerate(standins.names())}
        lines.sort(key=lambda ln: (order[ln.split(";")[1]], ln.split(";")[9], ln.split(";")[0]))
        with open(args.out, "w") as f:
            f.write(CSV_HEADER + "\n")
            f.write("\n".join(lines) + "\n")
        bad = [ln for ln in lines if not ln.endswith(";OK")]
        print(f"{len(lines)} lines -> {args.out}; {len(bad)} not OK")
        for ln in bad[:20]:
            print("  " + ln)
    return 0


if __name__ == "__main__":
    sys.exit(main())
