#!/bin/bash
# Regenerate everything under profiles/ that is quoted in DESIGN.md, in one GPU-box session:
#   tools/refresh_profiles.sh [outdir]      (default gpurun_out/refresh; copy the results into profiles/ as r<round>_* afterwards)
# Steps are joined so that a failing GPU step stops the script (no GPU step runs after a timeout).
set -eu -o pipefail
OUT=${1:-gpurun_out/refresh}
mkdir -p "$OUT"
export TMPDIR=/tmp
step() { echo "[refresh] $*"; }

step "bench line"
timeout -k 10 300 python3 bench.py 2>/dev/null | tail -1 > "$OUT/bench.json"
cat "$OUT/bench.json" | cut -c1-200

step "rocprofv3 kernel stats, default two lanes"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats2" -o r -- python3 bench.py --no-cpu-baseline --no-precision-modes > "$OUT/stats2.log" 2>&1
step "rocprofv3 kernel stats, single lane"
KP2D_LANES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats1" -o r -- python3 bench.py --no-cpu-baseline --no-precision-modes > "$OUT/stats1.log" 2>&1
find "$OUT/stats1" "$OUT/stats2" -name "*kernel_trace.csv" -delete      # only the --stats summaries are kept

step "per-layer table"
timeout -k 10 200 python3 tools/layer_profile.py > "$OUT/layers.txt" 2>/dev/null
head -3 "$OUT/layers.txt"

step "PMC passes"
tools/pmc_collect.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1

step "sweep"
: > "$OUT/sweep.jsonl"
sweep() { timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> "$OUT/sweep.jsonl"; tail -1 "$OUT/sweep.jsonl" | cut -c60-100; }
sweep --height 120 --width 160
sweep
sweep --batch 32
sweep --height 480 --width 640 --batch 32
sweep --config S_A --v3 --n-classes 19 --height 480 --width 640 --batch 32
sweep --config S_A --v3
sweep --config N
sweep --batch 1 --steps 200
sweep --precision fp32
step "LightGlue"
timeout -k 10 200 python3 tools/bench_lightglue.py 2>/dev/null | tail -1 > "$OUT/lightglue.jsonl"
timeout -k 10 200 python3 tools/bench_lightglue.py --pairs 1 2>/dev/null | tail -1 >> "$OUT/lightglue.jsonl"
cut -c1-200 "$OUT/lightglue.jsonl"
step "PCIe-inclusive front-end"
timeout -k 10 200 python3 tools/bench_frontend.py 2>/dev/null | tail -1 > "$OUT/frontend.jsonl"
timeout -k 10 200 python3 tools/bench_frontend.py --pinned 2>/dev/null | tail -1 >> "$OUT/frontend.jsonl"
timeout -k 10 200 python3 tools/bench_frontend.py --batch 1 --steps 200 2>/dev/null | tail -1 >> "$OUT/frontend.jsonl"
cut -c1-200 "$OUT/frontend.jsonl"
step done
