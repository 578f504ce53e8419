#!/bin/bash
# rocprofv3 kernel stats + two PMC passes over tools/bench_match.py (the on-device matcher):  tools/pmc_match.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_match}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- python3 tools/bench_match.py --reps 20 > "$OUT/stats.log" 2> "$OUT/stats.err" || echo "stats pass failed rc=$?"
run() {
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o "$name" -- python3 tools/bench_match.py --reps 3 > "$OUT/$name.log" 2> "$OUT/$name.err" || echo "pass $name failed rc=$?"
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
ls "$OUT"
