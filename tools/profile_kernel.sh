#!/bin/bash
# Profile one kernel configuration of tools/tune.py with rocprofv3: a kernel-trace
# pass (durations) and one PMC pass each for FETCH_SIZE and WRITE_SIZE (HBM traffic).
#   tools/profile_kernel.sh TAG --format csr_merge --variants 0 [more tune.py options]
# Output directories: gpurun_out/prof_TAG_{trace,fetch,write}; summary JSON: gpurun_out/prof_TAG.json
# Each pass runs under its own time limit; the script stops at the first failure.
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
# PROF_KERNEL_REGEX limits the counter passes to the kernels it matches (the others run unprofiled)
only=()
if [ -n "$PROF_KERNEL_REGEX" ]; then only=(--kernel-include-regex "$PROF_KERNEL_REGEX"); fi
timeout -k 10 240 rocprofv3 --kernel-trace --stats -f csv -d "$out/prof_${tag}_trace" -o t -- python3 "$root/tools/tune.py" "$@" --steps 10 > "$out/prof_${tag}_trace.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE "${only[@]}" -f csv -d "$out/prof_${tag}_fetch" -o f -- python3 "$root/tools/tune.py" "$@" --steps 5 > "$out/prof_${tag}_fetch.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE "${only[@]}" -f csv -d "$out/prof_${tag}_write" -o w -- python3 "$root/tools/tune.py" "$@" --steps 5 > "$out/prof_${tag}_write.log" 2>&1
# read requests of the L2 to the fabric by size: an exact byte count for any access width
timeout -s KILL 240 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum "${only[@]}" -f csv -d "$out/prof_${tag}_rdreq" -o r -- python3 "$root/tools/tune.py" "$@" --steps 5 > "$out/prof_${tag}_rdreq.log" 2>&1
python3 "$root/tools/pmc_summary.py" --skip 3 --out "$out/prof_${tag}.json" "$out/prof_${tag}_trace" "$out/prof_${tag}_fetch" "$out/prof_${tag}_write" "$out/prof_${tag}_rdreq" > /dev/null
rm -rf "$out/prof_${tag}_rdreq"
# keep the per-kernel statistics, drop the raw traces (they exceed what travels back)
find "$out/prof_${tag}_trace" -name "*kernel_stats.csv" -exec cp {} "$out/prof_${tag}_kernel_stats.csv" \;
rm -rf "$out/prof_${tag}_trace" "$out/prof_${tag}_fetch" "$out/prof_${tag}_write"
echo "profile $tag done"
