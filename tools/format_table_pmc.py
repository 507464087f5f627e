#!/usr/bin/env python3
"""Every format on one synthetic matrix with the bytes it actually moved: per (format, dtype) one
untraced tools/tune.py run (kernel trial, median time of the tuned product), then rocprofv3 passes of
the same configuration for FETCH_SIZE and WRITE_SIZE.  Writes a markdown table with the algorithmic
GB/s (what the roofline fraction is defined on) next to the moved bytes and the moved GB/s.

    python tools/format_table_pmc.py --matrix stencil27 --n 252 --dtypes f64,f32 --out profiles/r02_format_table_stencil27_252.md

This script never touches the GPU itself; every measurement is a child process.  Moved bytes =
FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024 summed over the kernels of one product (on gfx950 FETCH_SIZE
tallies the 128-byte fabric read requests of these kernels as 64 bytes; profiles/*_pmc_by_request_size.json).
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
TUNE = os.path.join(ROOT, "tools", "tune.py")
WARMUP = 3      # untimed products tune.py runs before the timed ones
BLOCK_MAPS = "0,1,2,3,4,5,8"   # what spmv_hip_analyze compares (option block_map)


def tune(extra, steps, profiler=None, timeout=400, every=False):
    cmd = [sys.executable, TUNE, "--steps", str(steps)] + extra
    if profiler:
        cmd = profiler + ["--"] + cmd
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    if p.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed:\n{p.stdout[-1500:]}\n{p.stderr[-1500:]}")
    results = [json.loads(line[7:]) for line in p.stdout.splitlines() if line.startswith("RESULT ")]
    if not results:
        raise RuntimeError("tune.py printed no RESULT line")
    return results if every else results[-1]


def counter_per_product(directory, counter, launches):
    """Sum over the product's kernels (those launched exactly `launches` times) of the counter's mean."""
    per = collections.defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter:
                    per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    total, names = 0.0, []
    for name, v in per.items():
        if len(v) == launches and "spmv::" in name:
            total += sum(v[WARMUP:]) / len(v[WARMUP:])
            names.append(name.split("(")[0].replace("void ", ""))
    return total, names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--matrix", default="stencil27")
    ap.add_argument("--n", type=int, default=252)
    ap.add_argument("--formats", default="csr_scalar,csr_vector,csr_merge,csr_bal,sell,sell_dfc,sellp,ell")
    ap.add_argument("--dtypes", default="f64")
    ap.add_argument("--variants", default="auto", help="'auto' = what the kernel trial picks; 'general' = the fastest variant that reads every index")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    every_index = {"sell": "0,1,2,3,4,9,10,11,12", "sellp": "0,1,2,3,4,9,10,11,12", "ell": "0,1,2,3,4,9,10,11,12",
                   "csr_scalar": "0,2,3,4,8,10,11,12", "csr_merge": "0,1", "csr_bal": "0,1", "sell_dfc": "0,1,2,3,4,5,6,7,8",
                   "csr_vector": "5,6"}
    lines = [f"{args.matrix} n={args.n}, variants: {args.variants}; times are the smaller of the median of {args.steps} products with a HIP-event pair each and of {args.steps} products queued between one pair, bytes from "
             f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same configuration", "",
             "| format | dtype | variant | ms / SpMV | GFLOP/s | GB/s (algorithmic bytes) | % of 8 TB/s (algorithmic) | moved GB / SpMV (PMC) | "
             "moved / algorithmic | moved GB/s | % of 8 TB/s (moved) | stored / nnz |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for dname in args.dtypes.split(","):
        for name in args.formats.split(","):
            fmt = name.split("+")[0]
            base = ["--format", fmt, "--matrix", args.matrix, "--n", str(args.n), "--dtype", dname]
            if name.endswith("+hot"):
                base += ["--hot", "1"]
            if name.endswith("+xcd"):
                base += ["--hot", "2"]
            print(f"  ... {name} {dname}: timing", flush=True)
            try:
                if args.variants == "general":
                    trial = base + ["--variants", every_index[fmt] + (",24,25,26,27" if fmt in ("sell", "sellp", "ell") and dname == "f64" else "")]
                    if fmt == "csr_vector":                       # the LDS-staged kernel at 2, 4, 8 lanes per row
                        trial += ["--p0", "2,4,8"]
                    best = min(tune(trial, 8, every=True), key=lambda r: r["median_ms"])
                    pinned = base + ["--variants", str(best["variant"])] + (["--p0", str(best["lanes_per_row"])] if fmt == "csr_vector" else [])
                    # then the way of dealing the workgroups over the matrix, as the library's trial does
                    best = min(tune(pinned + ["--block-maps", BLOCK_MAPS], 8, every=True), key=lambda r: r["median_ms"])
                    first = tune(pinned + ["--block-maps", str(best["block_map"]), "--in-a-row"], args.steps)
                else:
                    first = tune(base + ["--variants", "auto", "--in-a-row"], args.steps)
                fixed = base + ["--variants", str(first["variant"]), "--block-maps", str(first.get("block_map", 0))]
                if fmt == "csr_vector":
                    fixed += ["--p0", str(first["lanes_per_row"])]
                if name.endswith("+xcd"):                         # the tile kernel of the column-split copy, as the trial chose it
                    fixed = base + ["--hot-variants", str(first.get("hot_variant", 0))]
                moved = {}
                for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                    d = tempfile.mkdtemp(prefix="ftpmc_", dir="/tmp")
                    print(f"  ... {name} {dname}: {counter} pass", flush=True)
                    tune(fixed, 5, profiler=["rocprofv3", "--pmc", counter, "--kernel-include-regex", "spmv::", "-f", "csv", "-d", d, "-o", "p"])
                    moved[counter], kernels = counter_per_product(d, counter, WARMUP + 5)
                    shutil.rmtree(d, ignore_errors=True)
                bytes_moved = moved["FETCH_SIZE"] * 1024 * 2 + moved["WRITE_SIZE"] * 1024
                ms, b_alg, nnz = first["median_ms"], first["b_alg"], first["nnz"]
                if first.get("in_a_row_ms"):
                    ms = min(ms, first["in_a_row_ms"])
                variant = f"{first['lanes_per_row']} lanes, {first['variant']}" if fmt == "csr_vector" else str(first["variant"])
                if name.endswith("+xcd"):
                    variant += f", tiles {first.get('hot_variant', 0)}"
                elif first.get("block_map", 0):
                    variant += f", block_map {first['block_map']}"
                lines.append(f"| {name} | {dname} | {variant} | {ms:.3f} | {2 * nnz / ms / 1e6:.0f} | {b_alg / ms / 1e6:.0f} | "
                             f"{b_alg / ms / 1e6 / 80:.1f} | {bytes_moved / 1e9:.3f} | {bytes_moved / b_alg:.2f} | {bytes_moved / ms / 1e6:.0f} | "
                             f"{bytes_moved / ms / 1e6 / 80:.1f} | {first['stored'] / max(nnz, 1):.3f} |")
                lines.append(f"<!-- {name} {dname}: kernels {', '.join(sorted(set(kernels)))} -->")
            except (RuntimeError, subprocess.TimeoutExpired) as e:
                lines.append(f"| {name} | {dname} | - | failed: {str(e)[:80]} | | | | | | | | |")
            print(lines[-2] if lines[-1].startswith("<!--") else lines[-1], flush=True)
    text = "\n".join(lines) + "\n"
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main()
