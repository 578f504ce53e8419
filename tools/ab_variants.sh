#!/bin/bash
# A/B timing of library builds on ONE GPU box (boxes of the pool differ by +-3 %, so variants are only comparable
# inside one session):  tools/ab_variants.sh OUT.jsonl [bench.py flags ...] -- LIB_A LIB_B ...
# Each library (a path, or "default") runs the bench line alternately, three rounds; the kernel under test is
# selected with KP2D_LIB (nano-vs-slam_amd/_lib.py).  Joined with && so a failing GPU step stops the script.
set -eu -o pipefail
OUT=$1; shift
FLAGS=()
while [ "$1" != "--" ]; do FLAGS+=("$1"); shift; done
shift
mkdir -p "$(dirname "$OUT")"; : > "$OUT"
for round in 1 2 3; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset KP2D_LIB; else export KP2D_LIB="$PWD/$lib"; fi
    line=$(timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-precision-modes --steps 40 "${FLAGS[@]}" 2>>"${OUT%.jsonl}.err" | tail -1)
    echo "{\"lib\": \"$lib\", \"round\": $round, \"line\": $line}" >> "$OUT"
    python3 - "$lib" "$line" <<'PY'
import json, sys
d = json.loads(sys.argv[2]); r = d.get("roofline", {})
print(f"{sys.argv[1]:60s} {d['value']:9.1f} {d['unit']}  {d['ms_per_step']:.3f} ms/step  conv {r.get('achieved')} TFLOP/s", flush=True)
PY
  done
done
