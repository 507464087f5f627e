#!/bin/bash
# Profile the headline benchmark with rocprofv3: a kernel-trace pass of a plain `python3 bench.py`, all
# defaults (durations), then one PMC pass each for FETCH_SIZE and WRITE_SIZE of the
# kernel variant that run picked and of the every-index-read variant it reported, recorded in
# profiles/hbm_traffic.json against the hash of the kernel sources.
#   tools/profile_bench.sh r02        -> profiles/r02_bench_n1.json, r02_bench_n1_kernel_stats.csv, hbm_traffic.json
# Each pass runs under its own time limit; the script stops at the first failure.
set -e
tag=${1:-rXX}
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/profbench_$tag
rm -rf "$out"; mkdir -p "$out" "$root/profiles"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -f csv -d "$out/trace" -o t -- python3 "$root/bench.py" > "$out/bench.json" 2> "$out/bench.err"
tail -1 "$out/bench.json" > "$root/profiles/${tag}_bench_n1.json"
find "$out/trace" -name "*kernel_stats.csv" -exec cp {} "$root/profiles/${tag}_bench_n1_kernel_stats.csv" \;
read -r variant general < <(python3 - "$root/profiles/${tag}_bench_n1.json" <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))["roofline"]
print(r["kernel_variant"], r.get("general", {}).get("kernel_variant", -1))
PY
)
for v in $variant $general; do
    [ "$v" -lt 0 ] && continue
    for c in FETCH_SIZE WRITE_SIZE; do
        timeout -s KILL 300 rocprofv3 --pmc $c -f csv -d "$out/v${v}_$c" -o p -- python3 "$root/bench.py" --variant "$v" --no-general --no-cpu-baseline --steps 10 --warmup 3 > "$out/v${v}_$c.log" 2>&1
    done
    python3 "$root/tools/record_traffic.py" --key "sell/stencil27/252/f64/gpus1/variant$v" --fetch "$out/v${v}_FETCH_SIZE" --write "$out/v${v}_WRITE_SIZE" --source "tools/profile_bench.sh $tag (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --variant $v)" > "$out/v${v}_traffic.json"
done
rm -rf "$out/trace" "$out"/v*_FETCH_SIZE "$out"/v*_WRITE_SIZE
# the line a plain `python3 bench.py` prints, now with the traffic record in place
timeout -k 10 400 python3 "$root/bench.py" > "$out/bench_final.json" 2> "$out/bench_final.err"
tail -1 "$out/bench_final.json" > "$root/profiles/${tag}_bench_n1.json"
# profiles/ does not travel back from a GPU box, gpurun_out/ does
cp "$root/profiles/hbm_traffic.json" "$root/profiles/${tag}_bench_n1.json" "$root/profiles/${tag}_bench_n1_kernel_stats.csv" "$out/"
echo "profile_bench $tag done: variant $variant, general $general"
