#!/bin/bash
# Profile the headline benchmark with rocprofv3 and record the HBM traffic of every configuration its line
# reports: see tools/profile_bench.py.
#   tools/profile_bench.sh r03    -> profiles/r03_bench_n1.json, r03_bench_n1_kernel_stats.csv, hbm_traffic.json
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
python3 "$root/tools/profile_bench.py" "${1:-rXX}"
