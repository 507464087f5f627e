#!/usr/bin/env python3
"""Summarise rocprofv3 output directories: per kernel, the mean of every PMC
counter found (counter_collection.csv) and the mean duration (kernel_trace.csv).

    python tools/pmc_summary.py --match spmv:: --out profiles/x.json DIR [DIR ...]

HBM bytes per launch are derived as FETCH_SIZE*1024*2 + WRITE_SIZE*1024 when both
counters are present: on gfx950 FETCH_SIZE counts 128-byte fabric read requests
as 64 bytes, hence the factor 2; WRITE_SIZE is taken as reported (KiB).
"""
import argparse
import collections
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--match", default="spmv::", help="substring a kernel name must contain")
    ap.add_argument("--skip", type=int, default=0, help="ignore the first N dispatches of each kernel (warm-up)")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    durations = collections.defaultdict(list)
    for d in args.dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if args.match in row["Kernel_Name"]:
                        counters[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    if args.match in row["Kernel_Name"]:
                        durations[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    out = {}
    for k in sorted(set(counters) | set(durations)):
        e = {}
        for c, v in counters[k].items():
            v = v[args.skip:] or v
            e[c] = {"mean": sum(v) / len(v), "dispatches": len(v)}
        if k in durations:
            v = durations[k][args.skip:] or durations[k]
            e["duration_us"] = {"mean": sum(v) / len(v) / 1e3, "min": min(v) / 1e3, "dispatches": len(v)}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = int(e["FETCH_SIZE"]["mean"] * 1024 * 2 + e["WRITE_SIZE"]["mean"] * 1024)
        sized = ["TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]
        if all(c in e for c in sized):
            e["read_bytes_by_request_size"] = int(sum(e[c]["mean"] * b for c, b in zip(sized, (32, 64, 128))))
        out[k] = e
    text = json.dumps(out, indent=1)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
