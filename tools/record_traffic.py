#!/usr/bin/env python3
"""Store the HBM traffic of one kernel configuration, measured by rocprofv3 PMC passes, in
profiles/hbm_traffic.json together with the hash of the kernel sources it was measured on.

    python tools/record_traffic.py --key sell/stencil27/252/f64/gpus1/variant22 \
        --fetch DIR_OF_FETCH_PASS --write DIR_OF_WRITE_PASS --source profiles/r02_bench_n1_pmc.json

The kernel taken from each pass is the one with the most dispatches among those whose name contains
--match (the timed loop of bench.py launches one SpMV kernel far more often than anything else).
bytes = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024: on gfx950 FETCH_SIZE tallies the 128-byte fabric
read requests of a streaming kernel as 64 bytes (checked per kernel against the request-size counters,
profiles/*_pmc_by_request_size.json), WRITE_SIZE reads the bytes as they are; both are KiB.
bench.py reports the figure only while the kernel sources still hash to the recorded value.
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def counter_of_main_kernel(directory, counter, match, skip):
    per = collections.defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and match in row["Kernel_Name"]:
                    per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    if not per:
        raise SystemExit(f"no {counter} rows for kernels matching {match!r} under {directory}")
    name = max(per, key=lambda k: len(per[k]))
    v = per[name][skip:] or per[name]
    return name, sum(v) / len(v), len(v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key", required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--match", default="spmv::")
    ap.add_argument("--skip", type=int, default=3, help="ignore the first dispatches (trial and warm-up)")
    ap.add_argument("--source", default="", help="where the summary of these passes is kept")
    args = ap.parse_args()
    kf, fetch, nf = counter_of_main_kernel(args.fetch, "FETCH_SIZE", args.match, args.skip)
    kw, write, nw = counter_of_main_kernel(args.write, "WRITE_SIZE", args.match, args.skip)
    if kf != kw:
        raise SystemExit(f"the passes disagree on the main kernel: {kf} vs {kw}")
    entry = {"bytes": int(fetch * 1024 * 2 + write * 1024), "read_bytes": int(fetch * 1024 * 2), "write_bytes": int(write * 1024),
             "kernel": kf, "dispatches": [nf, nw], "kernel_sources_sha256": bench.kernel_source_hash(), "source": args.source}
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {}
    table = {k: v for k, v in table.items() if isinstance(v, dict)}       # figures without a source hash are dropped
    table[args.key] = entry
    with open(path, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
        f.write("\n")
    print(json.dumps({args.key: entry}, indent=1))


if __name__ == "__main__":
    main()
