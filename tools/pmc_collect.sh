#!/bin/bash
# Collect rocprofv3 PMC counters for one forward of the bench workload, one counter group per pass
# (gfx950 slot limits: 8 SQ, 4 TCC with FETCH_SIZE=3 / WRITE_SIZE=2).  --pmc runs use --kernel-trace only.
#   usage: tools/pmc_collect.sh <outdir> [layer_profile args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
mkdir -p "$OUT"
export TMPDIR=/tmp
# one stream lane: every conv launch then covers the whole 64-frame batch, the same launch shape bench.py's
# HIP-event roofline figures are taken on (profiling mode runs single-lane); per-launch bytes are comparable
export KP2D_LANES=1
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o "$name" -- \
      python3 tools/layer_profile.py --reps 1 "${EXTRA[@]}" > "$OUT/$name.log" 2>&1 || echo "pass $name failed rc=$?"
}
EXTRA=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS
run tcc_rd FETCH_SIZE GRBM_GUI_ACTIVE
run tcc_wr WRITE_SIZE TCC_HIT TCC_MISS
ls "$OUT"
