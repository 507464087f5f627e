#!/bin/bash
# Which unit the multi-vector product keeps busy: rocprofv3 counter passes around tools/multivec_table.py.
#   tools/profile_multivec.sh TAG --formats sellp --vectors 8
# Output: gpurun_out/mv_TAG_<pass>/ (csv per pass) and gpurun_out/mv_TAG.json (mean per kernel and counter).
# Each pass holds at most 8 SQ, 4 TCP, 2 TA, 2 TD counters and runs under its own time limit; the script stops
# at the first failure.
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args=("$@")
pass() {
    name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-include-regex "_mv_" -f csv -d "$out/mv_${tag}_$name" -o p -- \
        python3 "$root/tools/multivec_table.py" "${args[@]}" --steps 4 > "$out/mv_${tag}_$name.log" 2>&1
}
pass sq SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum
python3 "$root/tools/pmc_summary.py" --skip 2 --out "$out/mv_${tag}.json" "$out/mv_${tag}_sq" "$out/mv_${tag}_ta" "$out/mv_${tag}_tlb" > /dev/null
echo "multivec counters $tag done"
