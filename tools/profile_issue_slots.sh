#!/bin/bash
# Which unit a kernel keeps busy: four rocprofv3 counter passes (instruction mix, busy / wait cycles per
# unit, LDS conflicts, L1 -> L2 requests) of one tools/tune.py configuration.
#   tools/profile_issue_slots.sh TAG --format csr_scalar --variants 11 [more tune.py options]
# Summary: gpurun_out/issue_TAG.json (per kernel: the mean of every counter over the timed launches).
# SQ_WAVE_CYCLES, SQ_WAIT_* and SQ_ACTIVE_INST_* count quad-cycles summed over all waves; each pass holds at
# most 8 SQ, 4 TCP and 4 TCC counters.  Each pass runs under its own time limit; the script stops at the
# first failure.
set -e
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
pass() {
    name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --kernel-include-regex "spmv::" -f csv -d "$out/issue_${tag}_$name" -o p -- \
        python3 "$root/tools/tune.py" "${args[@]}" --steps 5 > "$out/issue_${tag}_$name.log" 2>&1
}
args=("$@")
pass mix SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES
pass busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_THREAD_CYCLES_VALU
pass l1 TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
python3 "$root/tools/pmc_summary.py" --skip 3 --out "$out/issue_${tag}.json" "$out/issue_${tag}_mix" "$out/issue_${tag}_busy" \
    "$out/issue_${tag}_lds" "$out/issue_${tag}_l1" > /dev/null
rm -rf "$out/issue_${tag}_mix" "$out/issue_${tag}_busy" "$out/issue_${tag}_lds" "$out/issue_${tag}_l1"
echo "issue slots $tag done"
