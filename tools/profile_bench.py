#!/usr/bin/env python3
"""Profile the headline benchmark with rocprofv3 and record the HBM traffic of every configuration its line reports.

    python tools/profile_bench.py r03

1. a kernel-trace pass of a plain `python3 bench.py` (all defaults) -> profiles/<tag>_bench_n1_kernel_stats.csv;
2. per configuration of that line (the headline variant, its every-index-read variant, each other_configs entry):
   one pass each with --pmc FETCH_SIZE and --pmc WRITE_SIZE of `bench.py` restricted to that configuration with the
   kernel variant pinned, and for products made of several kernels (hot_columns 2) a kernel-trace pass for their
   durations; the bytes per product go to profiles/hbm_traffic.json under the configuration's key, with the hash
   of the kernel sources they were measured on (bench.py reports a figure only while the sources still hash to it);
3. the line a plain `python3 bench.py` prints with those records in place -> profiles/<tag>_bench_n1.json.

bytes = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024: on gfx950 FETCH_SIZE tallies the 128-byte fabric read requests
of a streaming kernel as 64 bytes; WRITE_SIZE reads the bytes as they are; both are KiB.
This process never touches a GPU; every pass is a child under its own time limit, and the first failure ends the run.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

STEPS, WARMUP = 10, 3
# what the kernel trial chooses from on the headline matrix: the index-skipping kernels (27 columns per batch with 8-byte
# loads, 14 with 16-byte loads) and the widest every-index kernels, each with and without non-temporal loads; every block_map
# the trial compares
SWEEP_SELL_VARIANTS = "18,22,28,29,30,31,4,12,24,26"
SWEEP_MAPS = "0,1,2,3,4,5,8"
SHORT = (("wave_tile", "tile"), ("tile_fixup", "fixup"), ("permute_x", "permute_x"), ("sum_partials", "sum_partials"))
# (csr_block_pattern_kernel belongs to the creators: every CSR-based handle scans its row blocks once)
PRODUCT_KERNELS = re.compile(r"spmv::(sliced|csr_(?!block_pattern)|wave_tile|tile_fixup|permute_x|sum_partials|sell_dfc)")


def run(cmd, log, limit):
    with open(log, "w") as f:
        p = subprocess.run(cmd, stdout=f, stderr=subprocess.STDOUT, timeout=limit, cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp"))
    if p.returncode != 0:
        raise SystemExit(f"{' '.join(cmd[:6])} ... failed with status {p.returncode}: see {log}")


def counter_per_product(directory, counter, products):
    """{kernel name: counter summed over its dispatches / products} for the kernels of the product."""
    total = collections.defaultdict(float)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and PRODUCT_KERNELS.search(row["Kernel_Name"]):
                    total[row["Kernel_Name"]] += float(row["Counter_Value"])
    if not total:
        raise SystemExit(f"no {counter} rows under {directory}")
    return {k: v / products for k, v in total.items()}


def counter_by_setting(directory, counter, manifest):
    """Per setting of a tools/traffic_sweep.py run: the counter summed over the product kernels of one product.
    The product kernels' dispatches, in dispatch order, split evenly over the settings of the manifest."""
    rows = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and PRODUCT_KERNELS.search(row["Kernel_Name"]):
                    order = row.get("Dispatch_Id") or row.get("Start_Timestamp") or row.get("Correlation_Id")
                    rows.append((int(order), float(row["Counter_Value"])))
    rows.sort()
    n = len(manifest["order"])
    if not rows or len(rows) % (n * manifest["launches"]):
        raise SystemExit(f"{len(rows)} product-kernel dispatches under {directory} do not split over {n} settings x "
                         f"{manifest['launches']} products")
    per = len(rows) // n
    return [sum(v for _, v in rows[i * per:(i + 1) * per]) / manifest["launches"] for i in range(n)]


def duration_per_product(directory, products):
    total = collections.defaultdict(float)
    for path in glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if PRODUCT_KERNELS.search(row["Kernel_Name"]):
                    total[row["Kernel_Name"]] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
    return {k: v / products for k, v in total.items()}


def short_name(kernel):
    for needle, name in SHORT:
        if needle in kernel:
            return name
    return kernel.split("spmv::")[-1].split("<")[0].split("(")[0]


def configurations(line):
    """(key, bench.py flags) of everything the line reports traffic for."""
    out = []
    main_flags = ["--no-general", "--no-other-configs", "--no-cpu-baseline"]
    r = line["roofline"]
    def with_map(v, k):
        return f"variant{v}" + (f"+map{k}" if k else "")
    k = line["config"].get("block_map", 0)
    out.append((f"sell/stencil27/252/f64/gpus1/{with_map(r['kernel_variant'], k)}",
                ["--variant", str(r["kernel_variant"]), "--block-map", str(k)] + main_flags, False))
    if "general" in r:
        g, k = r["general"]["kernel_variant"], r["general"].get("block_map", 0)
        out.append((f"sell/stencil27/252/f64/gpus1/{with_map(g, k)}", ["--variant", str(g), "--block-map", str(k)] + main_flags, False))
    for o in line.get("other_configs", []):
        matrix, size = ("laplacian2d", "1000") if o["workload"].startswith("laplacian2d") else ("rmat", o["workload"].split()[2])
        flags = ["--matrix", matrix, "--size", size, "--format", o["format"], "--variant", str(o["kernel_variant"]),
                 "--block-map", str(o.get("block_map", 0))] + main_flags
        if o["format"] == "csr_vector":
            flags += ["--p0", str(o["lanes_per_row"])]
        if o["hot_columns"]:
            flags += ["--hot-columns", str(o["hot_columns"])]
        if o["hot_columns"] == 2:
            flags += ["--hot-variant", str(o["hot_variant"])]
        out.append((o["traffic_key"], flags, o["hot_columns"] == 2))
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    out = os.path.join(ROOT, "gpurun_out", f"profbench_{tag}")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    py, script = "python3", os.path.join(ROOT, "bench.py")
    # 1. the default command under the kernel trace
    run(["rocprofv3", "--kernel-trace", "--stats", "-f", "csv", "-d", os.path.join(out, "trace"), "-o", "t", "--", py, script],
        os.path.join(out, "bench_trace.log"), 600)
    line = json.loads([ln for ln in open(os.path.join(out, "bench_trace.log")) if ln.startswith("{") and '"metric"' in ln][-1])
    for path in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(path, os.path.join(ROOT, "profiles", f"{tag}_bench_n1_kernel_stats.csv"))
    shutil.rmtree(os.path.join(out, "trace"), ignore_errors=True)
    # 2. the counter passes
    table_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    table = json.load(open(table_path)) if os.path.exists(table_path) else {}
    # records taken on other kernel sources can no longer be reported: drop them
    table = {k: v for k, v in table.items() if isinstance(v, dict) and v.get("kernel_sources_sha256") == bench.kernel_source_hash()}
    products = WARMUP + 2 * STEPS          # bench.py: warm-up, the timed region, then the same launches queued in a row
    for n, (key, flags, split) in enumerate(configurations(line)):
        cmd = [py, script, "--steps", str(STEPS), "--warmup", str(WARMUP)] + flags
        per = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, f"c{n}_{counter}")
            run(["rocprofv3", "--pmc", counter, "--kernel-include-regex", "spmv::", "-f", "csv", "-d", d, "-o", "p", "--"] + cmd,
                os.path.join(out, f"c{n}_{counter}.log"), 400)
            per[counter] = counter_per_product(d, counter, products)
            shutil.rmtree(d, ignore_errors=True)
        read_b = {k: v * 1024 * 2 for k, v in per["FETCH_SIZE"].items()}
        write_b = {k: v * 1024 for k, v in per["WRITE_SIZE"].items()}
        entry = {"bytes": int(sum(read_b.values()) + sum(write_b.values())), "read_bytes": int(sum(read_b.values())),
                 "write_bytes": int(sum(write_b.values())), "kernel": "; ".join(sorted(read_b)),
                 "products": products, "kernel_sources_sha256": bench.kernel_source_hash(),
                 "source": f"tools/profile_bench.py {tag} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py {' '.join(flags)})"}
        if split:
            d = os.path.join(out, f"c{n}_trace")
            run(["rocprofv3", "--kernel-trace", "-f", "csv", "-d", d, "-o", "t", "--"] + cmd, os.path.join(out, f"c{n}_trace.log"), 400)
            ms = duration_per_product(d, products)
            shutil.rmtree(d, ignore_errors=True)
            entry["kernels"] = {short_name(k): {"ms": round(ms.get(k, 0.0), 4), "bytes": int(read_b[k] + write_b.get(k, 0.0))}
                                for k in sorted(read_b)}
        table[key] = entry
        with open(table_path, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
            f.write("\n")
        print(f"{key}: {entry['bytes'] / 1e9:.3f} GB per product", flush=True)
    # 2b. the kernel trial of a later run may settle on a neighbouring setting (the non-temporal twin of a variant,
    # another block_map): the traffic of every setting it chooses from, measured in one process per counter
    sweeps = [("stencil27", 252, "sell", SWEEP_SELL_VARIANTS)]
    rmat = [o for o in line.get("other_configs", []) if o["workload"].startswith("rmat") and not o["hot_columns"]]
    if rmat:
        sweeps.append(("rmat", int(rmat[0]["workload"].split()[2]), rmat[0]["format"], "0,1"))
    for matrix, size, fmt, variants in sweeps:
        per, manifest = {}, None
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d, mf = os.path.join(out, f"sweep_{matrix}_{counter}"), os.path.join(out, f"sweep_{matrix}_{counter}.json")
            run(["rocprofv3", "--pmc", counter, "--kernel-include-regex", "spmv::", "-f", "csv", "-d", d, "-o", "p", "--", py,
                 os.path.join(ROOT, "tools", "traffic_sweep.py"), "--matrix", matrix, "--size", str(size), "--format", fmt,
                 "--variants", variants, "--maps", SWEEP_MAPS, "--manifest", mf], os.path.join(out, f"sweep_{matrix}_{counter}.log"), 600)
            manifest = json.load(open(mf))
            per[counter] = counter_by_setting(d, counter, manifest)
            shutil.rmtree(d, ignore_errors=True)
        for i, (v, k) in enumerate(manifest["order"]):
            read_b, write_b = per["FETCH_SIZE"][i] * 1024 * 2, per["WRITE_SIZE"][i] * 1024
            key = f"{fmt}/{matrix}/{size}/f64/gpus1/variant{v}" + (f"+map{k}" if k else "")
            if key in table and table[key].get("kernel_sources_sha256") == bench.kernel_source_hash() and "kernels" in table[key]:
                continue
            table[key] = {"bytes": int(read_b + write_b), "read_bytes": int(read_b), "write_bytes": int(write_b),
                          "products": manifest["launches"], "kernel_sources_sha256": bench.kernel_source_hash(),
                          "source": f"tools/profile_bench.py {tag} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                    f"tools/traffic_sweep.py: {manifest['launches']} products per setting, split by dispatch order)"}
        with open(table_path, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
            f.write("\n")
        print(f"sweep {matrix} {fmt}: {len(manifest['order'])} settings recorded", flush=True)
    # 3. the line a plain run prints now
    with open(os.path.join(out, "bench_final.json"), "w") as f:
        p = subprocess.run([py, script], stdout=f, stderr=open(os.path.join(out, "bench_final.err"), "w"), timeout=600)
    if p.returncode != 0:
        raise SystemExit("the final bench run failed")
    final = [ln for ln in open(os.path.join(out, "bench_final.json")) if ln.startswith("{")][-1]
    open(os.path.join(ROOT, "profiles", f"{tag}_bench_n1.json"), "w").write(final)
    # profiles/ does not travel back from a GPU box, gpurun_out/ does
    for name in ("hbm_traffic.json", f"{tag}_bench_n1.json", f"{tag}_bench_n1_kernel_stats.csv"):
        shutil.copy(os.path.join(ROOT, "profiles", name), out)
    print(f"profile_bench {tag} done")


if __name__ == "__main__":
    main()
