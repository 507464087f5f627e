#!/bin/bash
# The reference's sweep (spmv.run over matrix_lists/matrix_list_160.txt, reference spmv.run:5-37) through
# the C driver: the 159 stand-ins written as .bin files, then ./spmv_hip.run once per MODE and dtype.
#   tools/run_c5_driver.sh [scale] [min_seconds]     -> gpurun_out/c5_driver/<mode>.csv and all.csv
# scale multiplies rows and nnz of every stand-in (1.0 = the SuiteSparse sizes, ~78 GB of .bin files).
set -e
scale=${1:-0.1}
secs=${2:-0.05}
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
out=$root/gpurun_out/c5_driver
rm -rf "$out"; mkdir -p "$out"
python3 tools/make_matrix_list.py standins_160 --scale "$scale" --device cuda:0 > "$out/generate.log" 2>&1
first=1
for mode in csr_base csr_vec csr_merge csr_bal ell sellp_vec sell:1024 sell_dfc:1024 auto; do
    for dt in f64 f32; do
        tag=$(echo "${mode}_$dt" | tr ':' '-')
        rm -rf "$out/tmp"; mkdir -p "$out/tmp"
        LIST=matrix_lists/standins_160.txt MODE="$mode:$dt" OUT="$out/tmp" SPMV_MIN_SECONDS=$secs timeout -k 10 900 ./spmv_hip.run > "$out/$tag.log" 2>&1
        mv "$out/tmp/1.csv" "$out/$tag.csv"
        if [ $first = 1 ]; then head -1 "$out/$tag.csv" > "$out/all.csv"; first=0; fi
        tail -n +2 "$out/$tag.csv" >> "$out/all.csv"
        echo "$tag: $(grep -c ' OK' "$out/$tag.log") OK, $(grep -c ' ERR$' "$out/$tag.log") ERR"
    done
done
rm -rf "$out/tmp" "$root/matrices/standins_160"
