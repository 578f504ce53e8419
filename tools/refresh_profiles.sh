#!/bin/bash
# Regenerate everything under profiles/ that is quoted in DESIGN.md, in one GPU-box session:
#   tools/refresh_profiles.sh [outdir] [part]      (default gpurun_out/refresh; copy the results into profiles/ as r<round>_* afterwards)
# part: all (default) | timed (bench lines, sweep, LightGlue, front-end) | prof (rocprofv3 stats, per-layer tables) | pmc — a gpurun call
# is limited to 20 minutes, the whole script takes longer
# Steps are joined so that a failing GPU step stops the script (no GPU step runs after a timeout).
set -eu -o pipefail
OUT=${1:-gpurun_out/refresh}
PART=${2:-all}
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
mkdir -p "$OUT"
export TMPDIR=/tmp
step() { echo "[refresh] $*"; }
# stderr of every GPU step goes to $OUT/<step>.err (never /dev/null): a fault or abort in a profiling step leaves its message there
ERRN=0
errf() { ERRN=$((ERRN + 1)); echo "$OUT/step_${ERRN}.err"; }

if want timed; then
step "bench line"
timeout -k 10 300 python3 bench.py 2>"$(errf)" | tail -1 > "$OUT/bench.json"
timeout -k 10 300 python3 bench.py --steps 1000 --no-cpu-baseline --no-precision-modes 2>"$(errf)" | tail -1 > "$OUT/bench_sustained_1000steps.json"
cat "$OUT/bench.json" | cut -c1-200

# (timed runs first, profiler passes last: a box that has just run PMC passes measured 3-8 % slower, and short
# latency-bound runs such as the one-pair matcher up to 1.6x slower)
step "sweep"
: > "$OUT/sweep.jsonl"
sweep() { timeout -k 10 300 python3 bench.py --no-cpu-baseline "$@" 2>"$(errf)" | tail -1 >> "$OUT/sweep.jsonl"; tail -1 "$OUT/sweep.jsonl" | cut -c60-100; }
sweep --height 120 --width 160
sweep --height 120 --width 160 --in-flight 4      # (short steps: four in flight, profiles/r5_hw_queues.txt)
sweep
sweep --batch 32
sweep --batch 32 --in-flight 4
sweep --height 480 --width 640 --batch 32
sweep --config S_A --v3 --n-classes 19 --height 480 --width 640 --batch 32
sweep --config S_A --v3
sweep --config N
sweep --batch 1 --steps 300 --in-flight 1      # (a latency figure: one frame at a time)
sweep --precision fp32
step "LightGlue"
timeout -k 10 300 python3 tools/bench_lightglue.py --sweep 8,16,32 --steps 60 --warmup 10 2>"$(errf)" | grep '^{' > "$OUT/lightglue.jsonl"
timeout -k 10 200 python3 tools/bench_lightglue.py --pairs 1 --steps 200 --warmup 20 2>"$(errf)" | tail -1 >> "$OUT/lightglue.jsonl"
cut -c1-200 "$OUT/lightglue.jsonl"
step "PCIe-inclusive front-end"
timeout -k 10 200 python3 tools/bench_frontend.py 2>"$(errf)" | tail -1 > "$OUT/frontend.jsonl"
timeout -k 10 200 python3 tools/bench_frontend.py --pinned 2>"$(errf)" | tail -1 >> "$OUT/frontend.jsonl"
timeout -k 10 200 python3 tools/bench_frontend.py --pinned --in-flight 2 2>"$(errf)" | tail -1 >> "$OUT/frontend.jsonl"
timeout -k 10 200 python3 tools/bench_frontend.py --pinned --in-flight 3 2>"$(errf)" | tail -1 >> "$OUT/frontend.jsonl"
timeout -k 10 200 python3 tools/bench_frontend.py --batch 1 --steps 2000 2>"$(errf)" | tail -1 >> "$OUT/frontend.jsonl"
timeout -k 10 200 python3 tools/bench_frontend.py --batch 1 --steps 2000 --match 2>"$(errf)" | tail -1 >> "$OUT/frontend.jsonl"
timeout -k 10 200 python3 tools/bench_frontend.py --batch 1 --steps 2000 --match --top-k-matches 200 2>"$(errf)" | tail -1 >> "$OUT/frontend.jsonl"
timeout -k 10 200 python3 tools/bench_frontend.py --batch 1 --steps 2000 --match --semantic 2>"$(errf)" | tail -1 >> "$OUT/frontend.jsonl"
timeout -k 10 200 python3 tools/bench_frontend.py --batch 1 --steps 1000 --lightglue --top-k-matches 200 2>"$(errf)" | tail -1 >> "$OUT/frontend.jsonl"
cut -c1-200 "$OUT/frontend.jsonl"
fi
if want prof; then
step "rocprofv3 kernel stats, the default bench command (two steps in flight: kernels of the two streams overlap, durations are wall time under sharing)"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats2" -o r -- python3 bench.py --no-cpu-baseline --no-precision-modes > "$OUT/stats2.log" 2>&1
step "rocprofv3 kernel stats, one step at a time on a single lane (nothing overlaps: the launch shape of bench.py's HIP-event figures)"
KP2D_LANES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats1" -o r -- python3 bench.py --no-cpu-baseline --no-precision-modes --in-flight 1 > "$OUT/stats1.log" 2>&1
find "$OUT/stats1" "$OUT/stats2" -name "*kernel_trace.csv" -delete      # only the --stats summaries are kept

step "per-layer tables"
timeout -k 10 200 python3 tools/layer_profile.py > "$OUT/layers.txt" 2>"$(errf)"
head -3 "$OUT/layers.txt"
timeout -k 10 200 python3 tools/layer_profile.py --batch 32 > "$OUT/layers_32frames.txt" 2>"$(errf)"
timeout -k 10 200 python3 tools/layer_profile.py --config S_A --v3 --n-classes 19 --height 480 --width 640 --batch 32 > "$OUT/layers_cfg4.txt" 2>"$(errf)"
timeout -k 10 200 python3 tools/layer_profile.py --batch 1 --reps 20 > "$OUT/layers_1frame.txt" 2>"$(errf)"

fi
if want pmc; then
step "PMC passes"
tools/pmc_collect.sh "$OUT/pmc" > "$OUT/pmc.log" 2>&1
python3 tools/pmc_summarize.py "$OUT/pmc" > "$OUT/pmc_summary.txt" 2>"$(errf)"
python3 tools/pmc_traffic.py "$OUT/pmc" "$OUT/traffic.json" > /dev/null 2>"$(errf)"
tools/pmc_collect.sh "$OUT/pmc_cfg4" --config S_A --v3 --n-classes 19 --height 480 --width 640 --batch 32 > "$OUT/pmc_cfg4.log" 2>&1
python3 tools/pmc_summarize.py "$OUT/pmc_cfg4" > "$OUT/pmc_cfg4_summary.txt" 2>"$(errf)"
find "$OUT/pmc" "$OUT/pmc_cfg4" -name "*.csv" -delete      # only the summaries are kept
fi

step done
