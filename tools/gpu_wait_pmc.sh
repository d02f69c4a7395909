#!/bin/bash
# Wait-state counters of the decode kernels: how many wave-cycles are parked (s_waitcnt / barrier), how many
# stall at issue, how many issue an instruction.  Counters only (no --stats / traces), 8 SQ counters per pass.
#   bash tools/gpu_wait_pmc.sh <tag> [kinds]        -> gpurun_out/<tag>/pmc_wait_{a,b}/ ; python tools/summarize_wait_pmc.py <tag>
TAG=${1:-r05a}; KINDS=${2:-randn,blank}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > "$OUT/sq_counters.txt"
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS"
Bc="SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"
timeout 300 rocprofv3 --pmc $A --kernel-trace --output-format csv -d "$OUT/pmc_wait_a" -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_kinds.py" --kinds $KINDS > "$OUT/pmc_wait_a.log" 2>&1; echo "pmc wait A rc=$?"
timeout 300 rocprofv3 --pmc $Bc --kernel-trace --output-format csv -d "$OUT/pmc_wait_b" -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_kinds.py" --kinds $KINDS > "$OUT/pmc_wait_b.log" 2>&1; rc=$?; echo "pmc wait B rc=$rc"
if [ $rc -ne 0 ]; then
  tail -5 "$OUT/pmc_wait_b.log"
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d "$OUT/pmc_wait_b" -o p -- python "$GRAFT_REPO_ROOT/tools/pmc_kinds.py" --kinds $KINDS > "$OUT/pmc_wait_b.log" 2>&1; echo "pmc wait B (short list) rc=$?"
fi
grep -h "launch" "$OUT/pmc_wait_a.log" | tail -8
